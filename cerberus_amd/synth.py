"""Synthetic sliding windows (numpy containers around include/vilo_synth.h).

The containers hold exactly what Estimator::optimization() (src/estimator/estimator.cpp:1054-1458)
reads: landmark/observation tables, per-interval preintegration records, the marginalisation prior
and the para_* state arrays in vector2double layout (estimator.cpp:848-901).
"""
import ctypes as C
import os

import numpy as np

from . import _ctypes as T

_synth = None


def synth_lib():
    global _synth
    if _synth is None:
        path = os.path.join(T.LIB_DIR, "libvilo_synth.so")
        if not os.path.exists(path):
            raise RuntimeError("libvilo_synth.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _synth = C.CDLL(path)
        _synth.vilo_synth_default_params.argtypes = [C.POINTER(T.SynthParams), C.c_int]
        _synth.vilo_synth_sizes.argtypes = [C.POINTER(T.SynthParams), T.c_int32_p, T.c_int32_p]
        _synth.vilo_synth_window.argtypes = [C.POINTER(T.Config), C.POINTER(T.SynthParams), C.POINTER(T.SynthOut)]
        _synth.vilo_synth_window.restype = C.c_int
    return _synth


def default_config():
    """config/a1_config/hardware_a1_vilo_config.yaml values (same numbers as vilo_default_config)."""
    c = T.Config()
    c.acc_n, c.acc_n_z, c.acc_w, c.gyr_n, c.gyr_w, c.g_norm = 0.9, 2.5, 0.0004, 0.05, 0.0002, 9.805
    c.phi_n = c.dphi_n = 0.00001
    c.rho_c_n, c.rho_nc_n = 0.00000001, 0.00000000001
    c.v_n_min_xy, c.v_n_min_z, c.v_n_min, c.v_n_max = 0.001, 0.005, 0.005, 900.0
    c.v_n_force_thres_ratio, c.v_n_term1_steep = 0.8, 10.0
    c.v_n_term2_var_rescale, c.v_n_term3_distance_rescale = 1.0e-6, 1.0e-3
    c.contact_sensor_type = 0
    ox = [0.1805, 0.1805, -0.1805, -0.1805]
    oy = [0.047, -0.047, 0.047, -0.047]
    d = [0.0838, -0.0838, 0.0838, -0.0838]
    for j in range(4):
        c.rho_fix[4 * j + 0], c.rho_fix[4 * j + 1], c.rho_fix[4 * j + 2], c.rho_fix[4 * j + 3] = ox[j], oy[j], d[j], 0.21
    for i in range(3):
        c.p_br[i] = 0.0
    for i in range(9):
        c.R_br[i] = 1.0 if i % 4 == 0 else 0.0
    c.focal_length, c.huber_delta = 460.0, 1.0
    return c


class PriorData:
    """MarginalizationInfo product (marginalization_factor.h:57-82) in numpy buffers."""

    def __init__(self):
        self.struct = T.Prior()
        self.x0 = np.zeros(7 * T.MAX_PRIOR_BLOCKS)
        self.J0 = np.zeros(T.MAX_PRIOR_DIM * T.MAX_PRIOR_DIM)
        self.r0 = np.zeros(T.MAX_PRIOR_DIM)
        self.rebind()

    def rebind(self):
        self.struct.x0, self.struct.J0, self.struct.r0 = T.dptr(self.x0), T.dptr(self.J0), T.dptr(self.r0)

    @property
    def n(self):
        return self.struct.n

    def J0_matrix(self):
        n = self.struct.n
        return self.J0[: n * n].reshape(n, n)

    def blocks(self):
        return [(self.struct.block_id[k], self.struct.block_size[k], self.struct.block_idx[k]) for k in range(self.struct.n_blocks)]

    def copy(self):
        p = PriorData()
        C.memmove(C.byref(p.struct), C.byref(self.struct), C.sizeof(T.Prior))
        p.x0[:], p.J0[:], p.r0[:] = self.x0, self.J0, self.r0
        p.rebind()
        return p


class Window:
    F = T.F

    def __init__(self, L, n_obs, n_samples):
        F = self.F
        self.L, self.n_obs, self.n_samples = L, n_obs, n_samples
        self.lm_start_frame = np.zeros(L, np.int32)
        self.lm_obs_offset = np.zeros(L + 1, np.int32)
        self.obs = np.zeros((n_obs, 11))
        self.obs_is_stereo = np.zeros(n_obs, np.uint8)
        self.samples = np.zeros((max(n_samples, 1), T.SAMPLE_DOUBLES))
        self.sample_offsets = np.zeros(F, np.int32)
        self.lin = np.zeros((F - 1, 10))
        self.pose = np.zeros((F, 7)); self.speed_bias = np.zeros((F, 9)); self.leg_bias = np.zeros((F, 4))
        self.ex_pose = np.zeros((2, 7)); self.td = np.zeros(1); self.inv_depth = np.zeros(L)
        self.truth_pose = np.zeros((F, 7)); self.truth_speed_bias = np.zeros((F, 9)); self.truth_leg_bias = np.zeros((F, 4))
        self.truth_inv_depth = np.zeros(L)
        self.prior = PriorData()
        self.preint = np.zeros((F - 1, T.PREINT_DOUBLES))       # filled by preintegration (GPU or oracle)
        self.preint_imu = np.zeros((F - 1, T.PREINT_IMU_DOUBLES))
        self.use_leg = 1
        # reference defaults for a moving robot with the shipped yaml: ESTIMATE_EXTRINSIC=1 (ex free once
        # |Vs[0]| > 0.2), ESTIMATE_TD=0 (td constant), OPTIMIZE_LEG_BIAS=1 (estimator.cpp:1074-1105)
        self.leg_bias_const, self.ex_const, self.td_const = 0, 0, 1

    def release_inputs(self):
        """Drop the host arrays a device batch has copied (observations, samples, preintegration records, prior): what stays is what
        `Batch.download` writes into (the state arrays) and the scalars. For harnesses that keep tens of thousands of windows resident."""
        self.obs = self.obs_is_stereo = self.samples = self.preint = self.preint_imu = None
        self.prior = None
        self.truth_pose = self.truth_speed_bias = self.truth_leg_bias = self.truth_inv_depth = None

    def twin(self):
        """Another window over the SAME input arrays (observations, samples, records, prior) with state arrays of its own: harnesses fill
        large batches with twins of a few generated windows without holding 0.5 MB of inputs per position."""
        import copy
        t = copy.copy(self)
        t.pose, t.speed_bias, t.leg_bias = self.pose.copy(), self.speed_bias.copy(), self.leg_bias.copy()
        t.ex_pose, t.td, t.inv_depth = self.ex_pose.copy(), self.td.copy(), self.inv_depth.copy()
        return t

    def state_arrays(self):
        return [self.pose, self.speed_bias, self.leg_bias, self.ex_pose, self.td, self.inv_depth]

    def clone_state(self):
        return [a.copy() for a in self.state_arrays()]

    def set_state(self, arrs):
        for dst, src in zip(self.state_arrays(), arrs):
            dst[...] = src

    def desc(self, types=T):
        """(WindowDesc, WindowState) ctypes views of this window for the struct family `types`."""
        d = types.WindowDesc()
        d.n_frames, d.n_landmarks, d.n_obs, d.use_leg = self.F, self.L, self.n_obs, self.use_leg
        d.lm_start_frame, d.lm_obs_offset = T.iptr(self.lm_start_frame), T.iptr(self.lm_obs_offset)
        d.obs, d.obs_is_stereo = T.dptr(self.obs), T.u8ptr(self.obs_is_stereo)
        d.preint = C.cast(self.preint.ctypes.data, C.POINTER(types.Preint))
        d.preint_imu = C.cast(self.preint_imu.ctypes.data, C.POINTER(types.PreintImu))
        d.prior = C.cast(C.pointer(self.prior.struct), C.POINTER(types.Prior)) if self.prior.struct.valid else None
        d.leg_bias_const, d.ex_const, d.td_const = self.leg_bias_const, self.ex_const, self.td_const
        s = types.WindowState()
        s.pose, s.speed_bias, s.leg_bias = T.dptr(self.pose), T.dptr(self.speed_bias), T.dptr(self.leg_bias)
        s.ex_pose, s.td, s.inv_depth = T.dptr(self.ex_pose), T.dptr(self.td), T.dptr(self.inv_depth)
        return d, s


def default_params(config=2, seed=20260925, n_landmarks=None, with_prior=True):
    p = T.SynthParams()
    synth_lib().vilo_synth_default_params(C.byref(p), config)
    p.seed = seed
    if n_landmarks is not None:
        p.n_landmarks = n_landmarks
    p.with_prior = 1 if with_prior else 0
    return p


def make_window(cfg=None, params=None, **kw):
    """Generate one synthetic window (preintegration records NOT filled: run vilo_preintegrate)."""
    lib = synth_lib()
    cfg = cfg or default_config()
    params = params or default_params(**kw)
    n_obs, n_s = C.c_int32(), C.c_int32()
    lib.vilo_synth_sizes(C.byref(params), C.byref(n_obs), C.byref(n_s))
    w = Window(params.n_landmarks, n_obs.value, n_s.value)
    o = T.SynthOut()
    o.lm_start_frame, o.lm_obs_offset = T.iptr(w.lm_start_frame), T.iptr(w.lm_obs_offset)
    o.obs, o.obs_is_stereo = T.dptr(w.obs), T.u8ptr(w.obs_is_stereo)
    o.samples = C.cast(w.samples.ctypes.data, C.POINTER(T.Sample))
    o.sample_offsets, o.lin = T.iptr(w.sample_offsets), T.dptr(w.lin)
    o.pose, o.speed_bias, o.leg_bias = T.dptr(w.pose), T.dptr(w.speed_bias), T.dptr(w.leg_bias)
    o.ex_pose, o.td, o.inv_depth = T.dptr(w.ex_pose), T.dptr(w.td), T.dptr(w.inv_depth)
    o.truth_pose, o.truth_speed_bias = T.dptr(w.truth_pose), T.dptr(w.truth_speed_bias)
    o.truth_leg_bias, o.truth_inv_depth = T.dptr(w.truth_leg_bias), T.dptr(w.truth_inv_depth)
    o.prior = C.pointer(w.prior.struct)
    rc = lib.vilo_synth_window(C.byref(cfg), C.byref(params), C.byref(o))
    if rc != 0:
        raise RuntimeError("vilo_synth_window failed: %d" % rc)
    return w
