"""Frame-by-frame replay: a synthetic sensor stream (include/vilo_synth.h, vilo_synth_stream_*) through the host-side
sliding-window manager (cerberus_amd/host/vilo_sliding_window.h, vilo_sw_*), i.e. what Estimator::processMeasurements /
processImage (src/estimator/estimator.cpp:400-846) do around optimization(). One robot or a fleet in lockstep (one batched
solve per image)."""
import ctypes as C
import os

import numpy as np

from . import _ctypes as T
from . import api, synth

MAX_SAMPLES, MAX_FEATURES = 256, 512


class SwOptions(C.Structure):
    _fields_ = [("use_leg", C.c_int32), ("optimize_leg_bias", C.c_int32), ("estimate_extrinsic", C.c_int32), ("estimate_td", C.c_int32),
                ("max_num_iterations", C.c_int32), ("fixed_iterations", C.c_int32), ("dump_dir", C.c_char_p),
                ("streaming_preintegration", C.c_int32), ("resident", C.c_int32)]


class StreamParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("imu_rate_hz", C.c_double), ("frame_rate_hz", C.c_double), ("pixel_noise", C.c_double),
                ("t0", C.c_double), ("cloud_per_10m", C.c_int32), ("max_features", C.c_int32), ("drop_prob", C.c_double),
                ("stereo_prob", C.c_double)]


_host = None


def host_lib():
    global _host
    if _host is None:
        api.lib()   # libvilo_gpu.so first: libvilo_host.so links against it
        path = os.path.join(T.LIB_DIR, "libvilo_host.so")
        if not os.path.exists(path):
            raise RuntimeError("libvilo_host.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _host = C.CDLL(path)
        _host.vilo_sw_create.restype = C.c_void_p
        _host.vilo_sw_create.argtypes = [C.c_void_p, C.POINTER(T.Config), C.POINTER(SwOptions)]
        _host.vilo_sw_destroy.argtypes = [C.c_void_p]
        _host.vilo_sw_attach_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _host.vilo_sw_attach_prior_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _host.vilo_sw_set_extrinsics.argtypes = [C.c_void_p, T.c_double_p, T.c_double_p, C.c_double]
        _host.vilo_sw_init_first_pose.argtypes = [C.c_void_p, T.c_double_p, T.c_double_p, T.c_double_p]
        _host.vilo_sw_process_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _host.vilo_sw_push_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _host.vilo_sw_process_image.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _host.vilo_sw_process_images.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _host.vilo_sw_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        _host.vilo_sw_last_summary.argtypes = [C.c_void_p, C.POINTER(T.SolveSummary)]
    return _host


def _dp(a):
    return a.ctypes.data_as(T.c_double_p)


class Stream:
    """vilo_synth_stream: samples + tracked features + ground truth, image by image."""

    def __init__(self, cfg, seed=20260925, t0=0.0, **kw):
        L = synth.synth_lib()
        L.vilo_synth_stream_create.restype = C.c_void_p
        L.vilo_synth_stream_create.argtypes = [C.POINTER(T.Config), C.POINTER(StreamParams)]
        L.vilo_synth_stream_next.argtypes = [C.c_void_p] + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                            C.c_void_p, C.c_void_p]
        L.vilo_synth_stream_extrinsics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vilo_synth_stream_destroy.argtypes = [C.c_void_p]
        self.L, self.p = L, StreamParams()
        L.vilo_synth_stream_default_params(C.byref(self.p))
        self.p.seed, self.p.t0 = seed, t0
        for k, v in kw.items():
            setattr(self.p, k, v)
        self.h = C.c_void_p(L.vilo_synth_stream_create(C.byref(cfg), C.byref(self.p)))
        self.max_feat = max(MAX_FEATURES, int(self.p.max_features) + 64)
        self.samples = np.zeros((MAX_SAMPLES, T.SAMPLE_DOUBLES))
        self.ids = np.zeros(self.max_feat, np.int32)
        self.obs = np.zeros((self.max_feat, 11))
        self.stereo = np.zeros(self.max_feat, np.uint8)
        self.truth = np.zeros(20)

    def extrinsics(self):
        tic, ric, td = np.zeros((2, 3)), np.zeros((2, 9)), C.c_double()
        self.L.vilo_synth_stream_extrinsics(self.h, tic.ctypes.data, ric.ctypes.data, C.byref(td))
        return tic, ric, td.value

    def next(self):
        ns, nf, hd = C.c_int(), C.c_int(), C.c_double()
        rc = self.L.vilo_synth_stream_next(self.h, self.samples.ctypes.data, MAX_SAMPLES, C.byref(ns), self.ids.ctypes.data, self.obs.ctypes.data,
                                           self.stereo.ctypes.data, self.max_feat, C.byref(nf), C.byref(hd), self.truth.ctypes.data)
        if rc != 0:
            raise RuntimeError("vilo_synth_stream_next: buffer too small")
        return dict(header=hd.value, samples=self.samples[:ns.value].copy(), ids=self.ids[:nf.value].copy(), obs=self.obs[:nf.value].copy(),
                    stereo=self.stereo[:nf.value].copy(), truth=self.truth.copy())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vilo_synth_stream_destroy(self.h)
            self.h = None


class SlidingWindow:
    """vilo::SlidingWindow behind its C entry points."""

    def __init__(self, ctx, cfg, use_leg=1, optimize_leg_bias=1, estimate_extrinsic=0, estimate_td=0, max_num_iterations=0, fixed_iterations=0,
                 dump_dir=None, streaming_preintegration=1, resident=1):
        self.H, self.ctx = host_lib(), ctx
        self._dump = dump_dir.encode() if dump_dir else None
        o = SwOptions(use_leg, optimize_leg_bias, estimate_extrinsic, estimate_td, max_num_iterations, fixed_iterations, self._dump,
                      streaming_preintegration, resident)
        self.h = C.c_void_p(self.H.vilo_sw_create(ctx.h if ctx is not None else None, C.byref(cfg), C.byref(o)))

    def attach_streams(self, pool, base_id):
        """pool: api.PreintStreams shared by a fleet; this robot uses objects base_id .. base_id + 10"""
        self._pool = pool
        self.H.vilo_sw_attach_streams(self.h, pool.h, base_id)

    def attach_prior_pool(self, pool, base_slot):
        """pool: api.PriorPool shared by a fleet; this robot alternates between slots base_slot and base_slot + 1"""
        self._ppool = pool
        self.H.vilo_sw_attach_prior_pool(self.h, pool.h, base_slot)

    def set_extrinsics(self, tic, ric, td):
        self.H.vilo_sw_set_extrinsics(self.h, _dp(np.ascontiguousarray(tic)), _dp(np.ascontiguousarray(ric)), td)

    def init_first_pose(self, p, R, v=None):
        p, R = np.ascontiguousarray(p, np.float64), np.ascontiguousarray(R, np.float64)
        vv = np.ascontiguousarray(v, np.float64) if v is not None else None
        self.H.vilo_sw_init_first_pose(self.h, _dp(p), _dp(R), _dp(vv) if vv is not None else None)

    def process_samples(self, samples):
        s = np.ascontiguousarray(samples)
        self.H.vilo_sw_process_samples(self.h, s.ctypes.data, len(s))

    def process_image(self, header, ids, obs, stereo):
        ids, obs, stereo = np.ascontiguousarray(ids, np.int32), np.ascontiguousarray(obs), np.ascontiguousarray(stereo, np.uint8)
        self.ctx._check(self.H.vilo_sw_process_image(self.h, header, len(ids), ids.ctypes.data, obs.ctypes.data, stereo.ctypes.data))

    def state(self):
        F = T.F
        flags = np.zeros(6, np.int32)
        out = dict(Ps=np.zeros((F, 3)), Rs=np.zeros((F, 3, 3)), Vs=np.zeros((F, 3)), Bas=np.zeros((F, 3)), Bgs=np.zeros((F, 3)), Rho=np.zeros((F, 4)),
                   tic=np.zeros((2, 3)), ric=np.zeros((2, 3, 3)))
        td = C.c_double()
        self.H.vilo_sw_get_state(self.h, flags.ctypes.data, *[out[k].ctypes.data for k in ("Ps", "Rs", "Vs", "Bas", "Bgs", "Rho", "tic", "ric")], C.byref(td))
        out.update(frame_count=int(flags[0]), solver_flag=int(flags[1]), marginalization_flag=int(flags[2]), n_optimizations=int(flags[3]),
                   feature_count=int(flags[4]), prior_n=int(flags[5]), td=td.value)
        return out

    def summary(self):
        s = T.SolveSummary()
        self.H.vilo_sw_last_summary(self.h, C.byref(s))
        return s

    def __del__(self):
        if getattr(self, "h", None):
            self.H.vilo_sw_destroy(self.h)
            self.h = None


class MeasurementProcessor:
    """vilo::MeasurementProcessor (cerberus_amd/host/vilo_sensor_buffer.h): timestamped IMU / leg messages and feature frames
    in, processIMULeg / processImage calls out, as Estimator::processMeasurements does (estimator.cpp:400-521)."""

    def __init__(self, sliding_window):
        self.H, self.sw = host_lib(), sliding_window
        self.H.vilo_mp_create.restype = C.c_void_p
        self.H.vilo_mp_create.argtypes = [C.c_void_p]
        self.H.vilo_mp_destroy.argtypes = [C.c_void_p]
        self.H.vilo_mp_input_imu.argtypes = [C.c_void_p, C.c_double, T.c_double_p, T.c_double_p]
        self.H.vilo_mp_input_leg.argtypes = [C.c_void_p, C.c_double, T.c_double_p, T.c_double_p, T.c_double_p]
        self.H.vilo_mp_input_feature.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.H.vilo_mp_input_sample.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        self.H.vilo_mp_busy_ms.argtypes = [C.c_void_p]
        self.H.vilo_mp_busy_ms.restype = C.c_double
        self.H.vilo_mp_queue_size.argtypes = [C.c_void_p]
        self.H.vilo_mp_process.argtypes = [C.c_void_p]
        self.H.vilo_mp_last_interval.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.h = C.c_void_p(self.H.vilo_mp_create(sliding_window.h))

    def input_sample(self, t, sample):
        """sample: one row of the Stream's sample array (dt ignored: the processor derives it from the stamps)"""
        s = np.ascontiguousarray(sample, np.float64)
        assert s.size == T.SAMPLE_DOUBLES
        self.H.vilo_mp_input_sample(self.h, t, s.ctypes.data)   # = inputIMU(acc, gyr) + inputLeg(phi, dphi, c)

    def busy_ms(self):
        """wall time spent inside this processor's C entry points so far (message intake, preintegration, solve, marginalisation, slide)"""
        return float(self.H.vilo_mp_busy_ms(self.h))

    def input_feature(self, t, ids, obs, stereo):
        ids, obs, stereo = np.ascontiguousarray(ids, np.int32), np.ascontiguousarray(obs), np.ascontiguousarray(stereo, np.uint8)
        rc = self.H.vilo_mp_input_feature(self.h, t, len(ids), ids.ctypes.data, obs.ctypes.data, stereo.ctypes.data)
        if rc < 0:
            raise RuntimeError("vilo_mp_input_feature failed: %d" % rc)
        return rc

    def process(self):
        rc = self.H.vilo_mp_process(self.h)
        if rc < 0:
            raise RuntimeError("vilo_mp_process failed: %d" % rc)
        return rc

    def queue_size(self):
        return self.H.vilo_mp_queue_size(self.h)

    def last_interval(self):
        out = np.zeros((MAX_SAMPLES, T.SAMPLE_DOUBLES))
        n = self.H.vilo_mp_last_interval(self.h, out.ctypes.data, MAX_SAMPLES)
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.H.vilo_mp_destroy(self.h)
            self.h = None


def push_samples(ctx, windows):
    """Between images: the samples the windows have buffered since the last call go to their device-resident preintegration objects
    (SlidingWindow::pushSamples: the reference's push_back per message). Changes when the integration happens, not its result."""
    H = host_lib()
    hs = (C.c_void_p * len(windows))(*[w.h for w in windows])
    ctx._check(H.vilo_sw_push_samples(ctx.h, hs, len(windows)))


def process_images(ctx, windows, frames):
    """Fleet step: frames[w] is Stream.next() of robot w; one batched solve for every robot that is due."""
    H = host_lib()
    hs = (C.c_void_p * len(windows))(*[w.h for w in windows])
    headers = np.array([f["header"] for f in frames])
    off = np.zeros(len(frames) + 1, np.int32)
    off[1:] = np.cumsum([len(f["ids"]) for f in frames])
    ids = np.ascontiguousarray(np.concatenate([f["ids"] for f in frames]), np.int32)
    obs = np.ascontiguousarray(np.concatenate([f["obs"] for f in frames]))
    st = np.ascontiguousarray(np.concatenate([f["stereo"] for f in frames]), np.uint8)
    ctx._check(H.vilo_sw_process_images(ctx.h, hs, len(windows), headers.ctypes.data, off.ctypes.data, ids.ctypes.data, obs.ctypes.data, st.ctypes.data))


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def start_robot(ctx, cfg, stream, **kw):
    """A SlidingWindow initialised at the stream's first true pose and velocity (the reference starts at rest from the averaged
    accelerometer direction, estimator.cpp:524-544; the synthetic trajectory is already moving at t0)."""
    sw = SlidingWindow(ctx, cfg, **kw)
    sw.set_extrinsics(*stream.extrinsics())
    return sw


def feed(sw, frame, first):
    if first:
        t = frame["truth"]
        sw.init_first_pose(t[0:3], quat_to_R(t[3:7]).ravel(), t[7:10])
    sw.process_samples(frame["samples"])
