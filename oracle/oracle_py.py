"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes binding of oracle/liboracle.so (see oracle/vilo_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Struct layouts are restated here (they mirror vilo_oracle.h, not the product's headers).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_PRIOR_BLOCKS = 40
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class Config(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "acc_n", "acc_n_z", "acc_w", "gyr_n", "gyr_w", "g_norm", "phi_n", "dphi_n", "rho_c_n", "rho_nc_n", "v_n_min_xy",
        "v_n_min_z", "v_n_min", "v_n_max", "v_n_force_thres_ratio", "v_n_term1_steep", "v_n_term2_var_rescale",
        "v_n_term3_distance_rescale")] + [("contact_sensor_type", C.c_int32), ("pad0", C.c_int32), ("rho_fix", C.c_double * 16),
                                           ("p_br", C.c_double * 3), ("R_br", C.c_double * 9), ("focal_length", C.c_double),
                                           ("huber_delta", C.c_double)]


class Sample(C.Structure):
    _fields_ = [("dt", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3), ("phi", C.c_double * 12),
                ("dphi", C.c_double * 12), ("c", C.c_double * 4)]


class Preint(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("delta_eps", C.c_double * 12), ("lin_ba", C.c_double * 3), ("lin_bg", C.c_double * 3), ("lin_rho", C.c_double * 4),
                ("jacobian", C.c_double * 961), ("covariance", C.c_double * 961)]


class PreintImu(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("lin_ba", C.c_double * 3), ("lin_bg", C.c_double * 3), ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


PREINT_DOUBLES = C.sizeof(Preint) // 8
PREINT_IMU_DOUBLES = C.sizeof(PreintImu) // 8


class Prior(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_blocks", C.c_int32), ("block_id", C.c_int32 * MAX_PRIOR_BLOCKS),
                ("block_size", C.c_int32 * MAX_PRIOR_BLOCKS), ("block_idx", C.c_int32 * MAX_PRIOR_BLOCKS),
                ("x0", dp), ("J0", dp), ("r0", dp), ("valid", C.c_int32), ("pad", C.c_int32)]


class WindowDesc(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_landmarks", C.c_int32), ("n_obs", C.c_int32), ("use_leg", C.c_int32),
                ("lm_start_frame", ip), ("lm_obs_offset", ip), ("obs", dp), ("obs_is_stereo", u8p),
                ("preint", C.POINTER(Preint)), ("preint_imu", C.POINTER(PreintImu)), ("prior", C.POINTER(Prior)),
                ("leg_bias_const", C.c_int32), ("ex_const", C.c_int32), ("td_const", C.c_int32), ("pad", C.c_int32)]


class WindowState(C.Structure):
    _fields_ = [("pose", dp), ("speed_bias", dp), ("leg_bias", dp), ("ex_pose", dp), ("td", dp), ("inv_depth", dp)]


class SolveOpts(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("fixed_iterations", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("jacobi_scaling", C.c_int32),
                ("recompute_sqrt_info", C.c_int32)]


class Summary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("num_successful", C.c_int32), ("termination", C.c_int32), ("pad", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("cost_trace", C.c_double * 64),
                ("radius_trace", C.c_double * 64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_window_cost.restype = C.c_double
        L.orc_window_dim.restype = C.c_int
        L.orc_sqrt_info.restype = C.c_int
        L.orc_solve_window.restype = C.c_int
        L.orc_marginalize.restype = C.c_int
        L.orc_fk.argtypes = [dp, C.c_double, dp, dp]
        L.orc_jac.argtypes = [dp, C.c_double, dp, dp]
        L.orc_dfk_drho.argtypes = [dp, C.c_double, dp, dp]
        L.orc_dJ_dq.argtypes = [dp, C.c_double, dp, dp]
        L.orc_dJ_drho.argtypes = [dp, C.c_double, dp, dp]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def default_config():
    c = Config()
    lib().orc_default_config(C.byref(c))
    return c


def default_opts(fixed_iterations=False, max_num_iterations=12):
    o = SolveOpts()
    lib().orc_default_opts(C.byref(o))
    o.fixed_iterations = 1 if fixed_iterations else 0
    o.max_num_iterations = max_num_iterations
    return o


def config_from(other):
    """Copy a product-side config struct (same layout) into an oracle Config."""
    c = Config()
    assert C.sizeof(other) == C.sizeof(c)
    C.memmove(C.byref(c), C.byref(other), C.sizeof(c))
    return c


def kin(q, lc, rho_fix):
    q, qp = _d(q)
    rf, rfp = _d(rho_fix)
    f = np.zeros(3); J = np.zeros(9); dfr = np.zeros(3); dJq = np.zeros(27); dJr = np.zeros(9)
    L = lib()
    L.orc_fk(qp, lc, rfp, f.ctypes.data_as(dp))
    L.orc_jac(qp, lc, rfp, J.ctypes.data_as(dp))
    L.orc_dfk_drho(qp, lc, rfp, dfr.ctypes.data_as(dp))
    L.orc_dJ_dq(qp, lc, rfp, dJq.ctypes.data_as(dp))
    L.orc_dJ_drho(qp, lc, rfp, dJr.ctypes.data_as(dp))
    # column-major (Eigen) -> numpy
    return dict(f=f, J=J.reshape(3, 3).T.copy(), df_drho=dfr, dJ_dq=dJq.reshape(3, 9).T.copy(), dJ_drho=dJr)


def preintegrate_imu_leg(cfg, samples, lin):
    """samples: (n+1, 35) incl. the constructor sample first; lin: 10 = ba bg rho. Returns (PREINT_DOUBLES,) array."""
    s, _ = _d(samples)
    lin, _ = _d(lin)
    out = np.zeros(PREINT_DOUBLES)
    sp = C.cast(s.ctypes.data, C.POINTER(Sample))
    lib().orc_preintegrate_imu_leg(C.byref(cfg), sp, C.cast(s[1:].ctypes.data, C.POINTER(Sample)), C.c_int(s.shape[0] - 1),
                                   lin[0:3].ctypes.data_as(dp), lin[3:6].ctypes.data_as(dp), lin[6:10].ctypes.data_as(dp),
                                   C.cast(out.ctypes.data, C.POINTER(Preint)))
    return out


def repropagate_imu_leg(cfg, samples, lin0, lins):
    """One IMULegIntegrationBase object: constructor + push_back of `samples` at lin0, then one repropagate(ba, bg, rho) per row of lins
    (imu_leg_integration_base.cpp:62-86: the contact-force filter of contact_sensor_type 2 is NOT reset between the passes).
    Returns (1 + len(lins), PREINT_DOUBLES): the object's public state after the original integration and after every repropagate."""
    s, _ = _d(samples)
    lins = np.atleast_2d(np.ascontiguousarray(lins, dtype=np.float64))
    out = np.zeros((1 + lins.shape[0], PREINT_DOUBLES))
    ff = np.zeros(36)
    sp = C.cast(s.ctypes.data, C.POINTER(Sample))
    for i, lin in enumerate([np.ascontiguousarray(lin0, dtype=np.float64)] + list(lins)):
        lin = np.ascontiguousarray(lin)
        lib().orc_preintegrate_imu_leg_ff(C.byref(cfg), sp, C.cast(s[1:].ctypes.data, C.POINTER(Sample)), C.c_int(s.shape[0] - 1),
                                          lin[0:3].ctypes.data_as(dp), lin[3:6].ctypes.data_as(dp), lin[6:10].ctypes.data_as(dp),
                                          ff.ctypes.data_as(dp), C.cast(out[i].ctypes.data, C.POINTER(Preint)))
    return out


def preintegrate_imu(cfg, samples, lin):
    s, _ = _d(samples)
    lin, _ = _d(lin)
    out = np.zeros(PREINT_IMU_DOUBLES)
    sp = C.cast(s.ctypes.data, C.POINTER(Sample))
    lib().orc_preintegrate_imu(C.byref(cfg), sp, C.cast(s[1:].ctypes.data, C.POINTER(Sample)), C.c_int(s.shape[0] - 1),
                               lin[0:3].ctypes.data_as(dp), lin[3:6].ctypes.data_as(dp),
                               C.cast(out.ctypes.data, C.POINTER(PreintImu)))
    return out


def step_FV(cfg, s0, s1, delta_q_xyzw, lin):
    s0, _ = _d(s0); s1, _ = _d(s1); dq, _ = _d(delta_q_xyzw); lin, _ = _d(lin)
    F = np.zeros((31, 31)); V = np.zeros((31, 46))
    lib().orc_imu_leg_step_FV(C.byref(cfg), C.cast(s0.ctypes.data, C.POINTER(Sample)), C.cast(s1.ctypes.data, C.POINTER(Sample)),
                              dq.ctypes.data_as(dp), lin[0:3].ctypes.data_as(dp), lin[3:6].ctypes.data_as(dp),
                              lin[6:10].ctypes.data_as(dp), F.ctypes.data_as(dp), V.ctypes.data_as(dp))
    return F, V


def sqrt_info(cov, mode=0):
    cov, cp = _d(cov)
    n = cov.shape[0]
    U = np.zeros((n, n))
    rc = lib().orc_sqrt_info(cp, C.c_int(n), C.c_int(mode), U.ctypes.data_as(dp))
    if rc != 0:
        raise FloatingPointError("orc_sqrt_info rc=%d" % rc)
    return U


def _eval(fn, ctx_args, params, nres, sizes, want_jac=True):
    """Generic CostFunction::Evaluate call. params: list of 1-D arrays. Returns r, [J_k (nres x size_k)]."""
    keep = [np.ascontiguousarray(p, dtype=np.float64) for p in params]
    pp = (dp * len(keep))(*[k.ctypes.data_as(dp) for k in keep])
    r = np.zeros(nres)
    if want_jac:
        Js = [np.zeros((nres, s)) for s in sizes]
        jp = (dp * len(Js))(*[j.ctypes.data_as(dp) for j in Js])
        fn(*ctx_args, pp, r.ctypes.data_as(dp), jp)
        return r, Js
    fn(*ctx_args, pp, r.ctypes.data_as(dp), None)
    return r, None


def eval_imu_leg(cfg, preint_arr, params, want_jac=True):
    pre = C.cast(np.ascontiguousarray(preint_arr).ctypes.data, C.POINTER(Preint))
    return _eval(lib().orc_eval_imu_leg, (C.byref(cfg), pre), params, 31, [7, 9, 4, 7, 9, 4], want_jac)


def eval_imu(cfg, preint_arr, params, want_jac=True):
    pre = C.cast(np.ascontiguousarray(preint_arr).ctypes.data, C.POINTER(PreintImu))
    return _eval(lib().orc_eval_imu, (C.byref(cfg), pre), params, 15, [7, 9, 7, 9], want_jac)


def eval_proj(kind, cfg, obs12, params, want_jac=True):
    o, op = _d(obs12)
    fn = [lib().orc_eval_proj2f1c, lib().orc_eval_proj2f2c, lib().orc_eval_proj1f2c][kind]
    sizes = [[7, 7, 7, 1, 1], [7, 7, 7, 7, 1, 1], [7, 7, 1, 1]][kind]
    return _eval(fn, (C.byref(cfg), op), params, 2, sizes, want_jac)


def eval_prior(prior_struct, params, want_jac=True):
    n = prior_struct.n
    sizes = [prior_struct.block_size[k] for k in range(prior_struct.n_blocks)]
    pr = C.cast(C.pointer(prior_struct), C.POINTER(Prior))
    return _eval(lib().orc_eval_prior, (pr,), params, n, sizes, want_jac)


def pose_plus(x, d):
    x, xp = _d(x); d, dptr_ = _d(d)
    out = np.zeros(7)
    lib().orc_pose_plus(xp, dptr_, out.ctypes.data_as(dp))
    return out


def huber(delta, s):
    rho = np.zeros(3)
    lib().orc_huber(C.c_double(delta), C.c_double(s), rho.ctypes.data_as(dp))
    return rho


# ---- window-level helpers: take a cerberus_amd.synth.Window-like container (numpy arrays only) ----
import sys as _sys

_THIS = _sys.modules[__name__]


def fill_preint(cfg, w):
    """Run the oracle preintegration on w.samples for every interval and fill w.preint / w.preint_imu."""
    for k in range(w.F - 1):
        a, b = w.sample_offsets[k], w.sample_offsets[k + 1]
        w.preint[k] = preintegrate_imu_leg(cfg, w.samples[a:b], w.lin[k])
        w.preint_imu[k] = preintegrate_imu(cfg, w.samples[a:b], w.lin[k][:6])


def window_cost(cfg, w):
    d, s = w.desc(_THIS)
    return lib().orc_window_cost(C.byref(cfg), C.byref(d), C.byref(s))


def window_normal_eq(cfg, w):
    d, s = w.desc(_THIS)
    n = lib().orc_window_dim(C.byref(d))
    H = np.zeros((n, n)); g = np.zeros(n); cost = C.c_double()
    lib().orc_window_normal_eq(C.byref(cfg), C.byref(d), C.byref(s), H.ctypes.data_as(dp), g.ctypes.data_as(dp), C.byref(cost))
    return H, g, cost.value


def solve_window(cfg, w, opts=None, check=True):
    """In place on w's state arrays. Returns Summary (check=False: also when the solve ended in FAILURE)."""
    opts = opts or default_opts()
    d, s = w.desc(_THIS)
    sm = Summary()
    rc = lib().orc_solve_window(C.byref(cfg), C.byref(d), C.byref(s), C.byref(opts), C.byref(sm))
    if rc != 0 and check:
        raise FloatingPointError("orc_solve_window failed rc=%d" % rc)
    return sm


class repropagation:
    """Context manager: while active, every IMU-leg factor evaluation of window w (solve_window, marginalize, window_cost, ...) first
    integrates its interval again from w.samples at the biases of the evaluation point (BASELINE configs[2])."""

    def __init__(self, w):
        self._s = np.ascontiguousarray(w.samples, dtype=np.float64)
        self._o = np.ascontiguousarray(w.sample_offsets, dtype=np.int32)

    def __enter__(self):
        lib().orc_set_repropagation(C.cast(self._s.ctypes.data, C.POINTER(Sample)), self._o.ctypes.data_as(C.POINTER(C.c_int32)))
        return self

    def __exit__(self, *a):
        lib().orc_set_repropagation(None, None)
        return False


def branch_counts():
    """Branches the last solve_window took: [gn, cauchy, interpolated, rejected, longest rejected run, invalid, mu escalations, accepted]."""
    out = (C.c_int * 8)()
    lib().orc_last_branch_counts(out)
    return list(out)


def gauge_fix(before_arrays, w):
    """before_arrays: state arrays before the solve (w.clone_state()); fixes w's states in place."""
    sb = WindowState()
    keep = [np.ascontiguousarray(a) for a in before_arrays]
    sb.pose, sb.speed_bias, sb.leg_bias, sb.ex_pose, sb.td, sb.inv_depth = [k.ctypes.data_as(dp) for k in keep]
    _, sa = w.desc(_THIS)
    lib().orc_gauge_fix(C.byref(sb), C.byref(sa), C.c_int(w.F))


def R2ypr(R):
    out = np.zeros(3)
    lib().orc_R2ypr(np.ascontiguousarray(R, dtype=np.float64).ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def ypr2R(ypr):
    out = np.zeros((3, 3))
    lib().orc_ypr2R(np.ascontiguousarray(ypr, dtype=np.float64).ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def marginalize(cfg, w, mode, prior_out, want_A=False):
    """prior_out: cerberus_amd.synth.PriorData-like (struct + x0/J0/r0 buffers). Returns (rc, m, A, b)."""
    d, s = w.desc(_THIS)
    pr = C.cast(C.pointer(prior_out.struct), C.POINTER(Prior))
    m = C.c_int(0)
    if want_A:
        nmax = 19 + w.L + 128
        A = np.zeros(nmax * nmax); b = np.zeros(nmax)
        rc = lib().orc_marginalize(C.byref(cfg), C.byref(d), C.byref(s), C.c_int(mode), pr, A.ctypes.data_as(dp),
                                   b.ctypes.data_as(dp), C.byref(m))
        tot = m.value + prior_out.struct.n
        return rc, m.value, A[: tot * tot].reshape(tot, tot).copy(), b[:tot].copy()
    rc = lib().orc_marginalize(C.byref(cfg), C.byref(d), C.byref(s), C.c_int(mode), pr, None, None, C.byref(m))
    return rc, m.value, None, None
