"""Thin ctypes binding of the C-ABI in include/vilo_gpu.h (libvilo_gpu.so).

There is no CPU fallback: constructing a Context without the HIP library or without a GPU raises.
The methods mirror the reference's operator interface for the hot path:
  eval_* ........ ceres::CostFunction::Evaluate of the five factor classes (src/factor/*)
  preintegrate .. IMULegIntegrationBase ctor + push_back (imu_leg_integration_base.cpp:7-136)
  solve_windows . Estimator::optimization() solve half (estimator.cpp:1054-1245)
  marginalize ... Estimator::optimization() marginalisation half (estimator.cpp:1247-1455)
"""
import ctypes as C
import os

import numpy as np

from . import _ctypes as T

_lib = None


class ViloError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("VILO_GPU_LIB") or os.path.join(T.LIB_DIR, "libvilo_gpu.so")
        if not os.path.exists(path):
            raise ViloError("libvilo_gpu.so is missing (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(path)
        L.vilo_last_error.restype = C.c_char_p
        L.vilo_last_solve_ms.restype = C.c_double
        L.vilo_solve_wave_lds_bytes.restype = C.c_size_t
        _lib = L
    return _lib


def default_solve_opts(fixed_iterations=False, max_num_iterations=12):
    o = T.SolveOpts()
    lib().vilo_default_solve_opts(C.byref(o))
    o.fixed_iterations = 1 if fixed_iterations else 0
    o.max_num_iterations = max_num_iterations
    return o


def _p(a):
    return None if a is None else a.ctypes.data_as(T.c_double_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Batch:
    """Device-resident batch of windows (vilo_batch)."""

    def __init__(self, ctx, windows):
        self.ctx, self.windows = ctx, windows
        n = len(windows)
        self._descs = (T.WindowDesc * n)()
        self._states = (T.WindowState * n)()
        for i, w in enumerate(windows):
            d, s = w.desc(T)
            self._descs[i], self._states[i] = d, s
        self.handle = C.c_void_p()
        ctx._check(lib().vilo_batch_create(ctx.h, n, self._descs, self._states, C.byref(self.handle)))

    def reset(self):
        self.ctx._check(lib().vilo_batch_reset(self.ctx.h, self.handle))

    def prepare(self):
        """sqrt_info of every preintegration record again (what the reference does per IMULegFactor::Evaluate), asynchronous."""
        self.ctx._check(lib().vilo_batch_prepare(self.ctx.h, self.handle))

    def set_samples(self, on=True):
        """BASELINE configs[2]: keep the windows' samples in HBM and integrate every interval again (repropagate) at the biases of
        every point the solver linearises; on=False goes back to records integrated once."""
        if not on:
            self.ctx._check(lib().vilo_batch_set_samples(self.ctx.h, self.handle, None, None))
            return
        parts, offs, base = [], [0], 0
        for w in self.windows:
            n = int(w.sample_offsets[-1])
            parts.append(w.samples[:n])
            offs.extend((w.sample_offsets[1:] + base).tolist())
            offs.extend([base + n] * (10 - (len(w.sample_offsets) - 1)))
            base += n
        samples = np.ascontiguousarray(np.concatenate(parts), dtype=np.float64)
        offsets = np.ascontiguousarray(offs, dtype=np.int32)
        self.ctx._check(lib().vilo_batch_set_samples(self.ctx.h, self.handle, C.cast(samples.ctypes.data, C.POINTER(T.Sample)), T.iptr(offsets)))

    def marginalize(self, modes, priors_out):
        """vilo_batch_marginalize at the batch's device state (call download() first: the windows' arrays are the host copy of it).
        priors_out: one synth.PriorData per window."""
        n = len(self.windows)
        m = (C.c_int * n)(*modes)
        outs = (T.Prior * n)()
        for i, p in enumerate(priors_out):
            p.rebind()
            outs[i] = p.struct
        self.ctx._check(lib().vilo_batch_marginalize(self.ctx.h, self.handle, n, self._descs, self._states, m, outs))
        for i, p in enumerate(priors_out):
            C.memmove(C.byref(p.struct), C.byref(outs[i]), C.sizeof(T.Prior))
            p.rebind()

    def solve(self, opts):
        self.ctx._check(lib().vilo_batch_solve(self.ctx.h, self.handle, C.byref(opts)))
        return lib().vilo_last_solve_ms(self.ctx.h)

    def download(self):
        """Write the solver output into the windows' state arrays; returns the list of summaries."""
        n = len(self.windows)
        summ = (T.SolveSummary * n)()
        self.ctx._check(lib().vilo_batch_download(self.ctx.h, self.handle, self._states, summ))
        return list(summ)

    def fetch(self, what, win=0, max_n=1 << 22):
        out = np.zeros(max_n)
        n = lib().vilo_debug_fetch(self.ctx.h, self.handle, what, win, _p(out), max_n)
        if n < 0:
            raise ViloError("vilo_debug_fetch(%d) -> %d" % (what, n))
        return out[:n].copy()

    def close(self):
        if self.handle:
            lib().vilo_batch_destroy(self.ctx.h, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreintStreams:
    """vilo_preint_streams: device-resident IMULegIntegrationBase objects updated by push_back as samples arrive."""

    def __init__(self, ctx, n, imu_only=False):
        self.ctx, self.n, self.imu_only = ctx, n, imu_only
        self.h = C.c_void_p()
        ctx._check((lib().vilo_preint_streams_create_imu if imu_only else lib().vilo_preint_streams_create)(ctx.h, n, C.byref(self.h)))

    def reset(self, ids, first, lin):
        ids, first, lin = np.ascontiguousarray(ids, np.int32), _c(first), _c(lin)
        self.ctx._check(lib().vilo_preint_streams_reset(self.ctx.h, self.h, len(ids), T.iptr(ids), C.cast(first.ctypes.data, C.POINTER(T.Sample)), _p(lin)))

    def push(self, ids, samples, offsets):
        ids, samples, offsets = np.ascontiguousarray(ids, np.int32), _c(samples), np.ascontiguousarray(offsets, np.int32)
        self.ctx._check(lib().vilo_preint_streams_push(self.ctx.h, self.h, len(ids), T.iptr(ids), C.cast(samples.ctypes.data, C.POINTER(T.Sample)),
                                                       T.iptr(offsets)))

    def read(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        if self.imu_only:
            out = np.zeros((len(ids), T.PREINT_IMU_DOUBLES))
            self.ctx._check(lib().vilo_preint_streams_read_imu(self.ctx.h, self.h, len(ids), T.iptr(ids), C.cast(out.ctypes.data, C.POINTER(T.PreintImu))))
            return out
        out = np.zeros((len(ids), T.PREINT_DOUBLES))
        self.ctx._check(lib().vilo_preint_streams_read(self.ctx.h, self.h, len(ids), T.iptr(ids), C.cast(out.ctypes.data, C.POINTER(T.Preint))))
        return out

    def close(self):
        if self.h:
            lib().vilo_preint_streams_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PriorPool:
    """vilo_prior_pool: last_marginalization_info objects whose J0 / r0 stay on the device between frames."""

    def __init__(self, ctx, n_slots):
        self.ctx, self.n = ctx, n_slots
        self.h = C.c_void_p()
        ctx._check(lib().vilo_prior_pool_create(ctx.h, n_slots, C.byref(self.h)))

    def upload(self, slot, prior):
        self.ctx._check(lib().vilo_prior_pool_upload(self.ctx.h, self.h, slot, C.byref(prior.struct) if prior is not None else None))

    def download(self, slot, prior_out):
        prior_out.rebind()
        self.ctx._check(lib().vilo_prior_pool_download(self.ctx.h, self.h, slot, C.byref(prior_out.struct)))
        return prior_out

    def dim(self, slot):
        return lib().vilo_prior_pool_dim(self.h, slot)

    def close(self):
        if self.h:
            lib().vilo_prior_pool_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    def __init__(self, cfg, device=0):
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = lib().vilo_create(C.byref(self.h), C.byref(cfg), device)
        if rc != 0:
            raise ViloError("vilo_create failed (%d): no usable HIP device %d; this library has no CPU path" % (rc, device))

    def close(self):
        if self.h:
            lib().vilo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ViloError("vilo error %d: %s" % (rc, lib().vilo_last_error(self.h).decode()))

    SOLVER_FORMS = {"auto": -1, "wave": 0, "split": 3, "mw8": 4}

    def set_solver_form(self, form):
        """vilo_set_solver_form: 'auto' (by batch size), 'wave', 'split' (bitwise equal to 'wave'), 'mw8'."""
        self._check(lib().vilo_set_solver_form(self.h, self.SOLVER_FORMS[form]))

    def set_prior_form(self, form):
        """vilo_set_prior_form: 'eigen' (J0 = sqrt(S) V^T like the reference, default) or 'factor' (pivoted Cholesky factor where no
        eigenvalue would be dropped: same J0^T J0 / J0^T r0, a quarter of a single window's marginalisation time)."""
        self._check(lib().vilo_set_prior_form(self.h, {"eigen": 0, "factor": 1}[form]))

    def set_compact_rows(self, on):
        """vilo_set_compact_rows: batches created afterwards may / may not use the compact 16-column visual rows."""
        self._check(lib().vilo_set_compact_rows(self.h, 1 if on else 0))

    # ---- ceres::CostFunction-shaped batched evaluation ----
    def eval_proj(self, kind, obs, params, want_jac=True):
        """kind 0/1/2 = TwoFrameOneCam / TwoFrameTwoCam / OneFrameTwoCam; params: list of (n, size) arrays."""
        obs = _c(obs)
        n = obs.shape[0]
        params = [_c(p) for p in params]
        sizes = [[7, 7, 7, 1, 1], [7, 7, 7, 7, 1, 1], [7, 7, 1, 1]][kind]
        r = np.zeros((n, 2))
        Js = [np.zeros((n, 2, s)) for s in sizes] if want_jac else [None] * len(sizes)
        fn = [lib().vilo_eval_proj2f1c, lib().vilo_eval_proj2f2c, lib().vilo_eval_proj1f2c][kind]
        self._check(fn(self.h, n, _p(obs), *[_p(p) for p in params], _p(r), *[_p(j) for j in Js]))
        return r, Js

    def eval_imu_leg(self, preint, params, want_jac=True):
        preint = _c(preint)
        n = preint.shape[0]
        params = [_c(p) for p in params]
        sizes = [7, 9, 4, 7, 9, 4]
        r = np.zeros((n, 31))
        Js = [np.zeros((n, 31, s)) for s in sizes] if want_jac else [None] * 6
        self._check(lib().vilo_eval_imu_leg(self.h, n, C.cast(preint.ctypes.data, C.POINTER(T.Preint)), *[_p(p) for p in params],
                                            _p(r), *[_p(j) for j in Js]))
        return r, Js

    def eval_imu(self, preint, params, want_jac=True):
        preint = _c(preint)
        n = preint.shape[0]
        params = [_c(p) for p in params]
        sizes = [7, 9, 7, 9]
        r = np.zeros((n, 15))
        Js = [np.zeros((n, 15, s)) for s in sizes] if want_jac else [None] * 4
        self._check(lib().vilo_eval_imu(self.h, n, C.cast(preint.ctypes.data, C.POINTER(T.PreintImu)), *[_p(p) for p in params],
                                        _p(r), *[_p(j) for j in Js]))
        return r, Js

    def eval_prior(self, prior, params_concat, want_jac=True):
        """prior: synth.PriorData; params_concat: (n_eval, sum of global block sizes)."""
        pc = _c(params_concat)
        n_eval = pc.shape[0]
        n = prior.struct.n
        sg = sum(prior.struct.block_size[k] for k in range(prior.struct.n_blocks))
        r = np.zeros((n_eval, n))
        J = np.zeros((n_eval, n, sg)) if want_jac else None
        self._check(lib().vilo_eval_prior(self.h, n_eval, C.byref(prior.struct), _p(pc), _p(r), _p(J)))
        return r, J

    def pose_plus(self, x, d):
        x, d = _c(x), _c(d)
        out = np.zeros_like(x)
        self._check(lib().vilo_pose_plus(self.h, x.shape[0], _p(x), _p(d), _p(out)))
        return out

    # ---- preintegration ----
    def preintegrate(self, samples, offsets, lin):
        samples, lin = _c(samples), _c(lin)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = offsets.shape[0] - 1
        out = np.zeros((n, T.PREINT_DOUBLES))
        self._check(lib().vilo_preintegrate(self.h, n, C.cast(samples.ctypes.data, C.POINTER(T.Sample)), T.iptr(offsets), _p(lin),
                                            C.cast(out.ctypes.data, C.POINTER(T.Preint))))
        return out

    def preintegrate_imu(self, samples, offsets, lin6):
        samples, lin6 = _c(samples), _c(lin6)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = offsets.shape[0] - 1
        out = np.zeros((n, T.PREINT_IMU_DOUBLES))
        self._check(lib().vilo_preintegrate_imu(self.h, n, C.cast(samples.ctypes.data, C.POINTER(T.Sample)), T.iptr(offsets),
                                                _p(lin6), C.cast(out.ctypes.data, C.POINTER(T.PreintImu))))
        return out

    def preintegrate_window(self, w):
        """Fill w.preint (and w.preint_imu) from w.samples on the GPU."""
        w.preint[...] = self.preintegrate(w.samples, w.sample_offsets, w.lin)
        w.preint_imu[...] = self.preintegrate_imu(w.samples, w.sample_offsets, np.ascontiguousarray(w.lin[:, :6]))

    def preintegrate_windows(self, windows):
        """One launch for all intervals of all windows."""
        samples = np.concatenate([w.samples[: w.sample_offsets[-1]] for w in windows])
        offs, base = [0], 0
        for w in windows:
            offs.extend((w.sample_offsets[1:] + base).tolist())
            base += int(w.sample_offsets[-1])
        lin = np.concatenate([w.lin for w in windows])
        out = self.preintegrate(samples, np.array(offs, np.int32), lin)
        k = 0
        for w in windows:
            w.preint[...] = out[k:k + w.F - 1]
            k += w.F - 1

    # ---- solve ----
    def solve_windows(self, windows, opts=None):
        """Estimator::optimization() solve half on a list of windows; states updated in place."""
        opts = opts or default_solve_opts()
        n = len(windows)
        descs, states = (T.WindowDesc * n)(), (T.WindowState * n)()
        for i, w in enumerate(windows):
            descs[i], states[i] = w.desc(T)
        return self.solve_window_descs(descs, states, opts)

    def solve_window_descs(self, descs, states, opts):
        """vilo_solve_windows on descriptor arrays the caller built (and keeps alive); VILO_ERR_NUMERIC is a per-window outcome
        (termination 2 in that window's summary), everything else raises."""
        n = len(descs)
        summ = (T.SolveSummary * n)()
        rc = lib().vilo_solve_windows(self.h, n, descs, states, C.byref(opts), summ)
        if rc != -4:   # VILO_ERR_NUMERIC
            self._check(rc)
        return list(summ)

    def set_host_pipeline(self, lanes, sub_windows=1024):
        """vilo_set_host_pipeline: how vilo_solve_windows cuts a call with many host windows into sub-batches (lanes < 2: never)."""
        self._check(lib().vilo_set_host_pipeline(self.h, lanes, sub_windows))

    def gauge_fix(self, before_arrays, w):
        keep = [np.ascontiguousarray(a) for a in before_arrays]
        sb = T.WindowState()
        sb.pose, sb.speed_bias, sb.leg_bias, sb.ex_pose, sb.td, sb.inv_depth = [T.dptr(k) for k in keep]
        _, sa = w.desc(T)
        self._check(lib().vilo_gauge_fix(self.h, 1, C.byref(sb), C.byref(sa), w.F))

    def marginalize(self, w, mode, prior_out):
        d, s = w.desc(T)
        self._check(lib().vilo_marginalize(self.h, 1, C.byref(d), C.byref(s), mode, C.byref(prior_out.struct)))
