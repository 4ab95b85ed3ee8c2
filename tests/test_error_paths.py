"""Error behaviour of the C-ABI (include/vilo_gpu.h vilo_status): bad arguments are refused before any device work, unsupported
problem shapes are reported as such with a message in vilo_last_error, and a window whose evaluation is not finite fails the way
ceres::Solve reports a failed evaluation (termination FAILURE, state untouched) without disturbing the other windows of the batch."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OK, NO_DEVICE, BAD_ARG, HIP, NUMERIC, UNSUPPORTED = 0, -1, -2, -3, -4, -5


@pytest.fixture(scope="module")
def ctx(cfg):
    from cerberus_amd import api
    c = api.Context(cfg, 0)
    yield c
    c.close()


def _win(cfg, ocfg, **kw):
    from cerberus_amd import synth
    from oracle import oracle_py as O
    w = synth.make_window(cfg, **kw)
    O.fill_preint(ocfg, w)
    return w


def _solve_rc(ctx, windows, n=None):
    from cerberus_amd import api, _ctypes as T
    W = len(windows)
    descs = (T.WindowDesc * W)(); states = (T.WindowState * W)(); summ = (T.SolveSummary * W)()
    for i, w in enumerate(windows):
        descs[i], states[i] = w.desc(T)
    opts = api.default_solve_opts(True, 3)
    return api.lib().vilo_solve_windows(ctx.h, W if n is None else n, descs, states, C.byref(opts), summ), summ


def test_status_codes_match_the_header():
    import os
    import re
    from conftest import ROOT
    txt = open(os.path.join(ROOT, "include", "vilo_gpu.h")).read()
    vals = dict(re.findall(r"(VILO_[A-Z_]+)\s*=\s*(-?\d+)", txt))
    assert int(vals["VILO_OK"]) == OK and int(vals["VILO_ERR_BAD_ARG"]) == BAD_ARG and int(vals["VILO_ERR_UNSUPPORTED"]) == UNSUPPORTED
    assert int(vals["VILO_ERR_NUMERIC"]) == NUMERIC and int(vals["VILO_ERR_HIP"]) == HIP and int(vals["VILO_ERR_NO_DEVICE"]) == NO_DEVICE


def test_bad_arguments_are_refused(ctx, cfg, ocfg):
    from cerberus_amd import api, _ctypes as T
    L = api.lib()
    w = _win(cfg, ocfg, n_landmarks=12, seed=1)
    assert _solve_rc(ctx, [w], n=0)[0] == BAD_ARG
    assert L.vilo_solve_windows(ctx.h, 1, None, None, None, None) == BAD_ARG
    assert L.vilo_marginalize(ctx.h, 1, None, None, 0, None) == BAD_ARG
    d, s = w.desc(T)
    p = T.Prior()
    assert L.vilo_marginalize(ctx.h, 1, C.byref(d), C.byref(s), 2, C.byref(p)) == BAD_ARG          # no such marginalization_flag
    assert L.vilo_optimize_windows(ctx.h, 1, C.byref(d), C.byref(s), None, None, None, None) == BAD_ARG
    # an interval without samples
    smp = np.zeros((3, T.SAMPLE_DOUBLES)); off = np.array([0, 2, 2], np.int32); lin = np.zeros((2, 10)); out = np.zeros((2, T.PREINT_DOUBLES))
    assert L.vilo_preintegrate(ctx.h, 2, C.cast(smp.ctypes.data, C.POINTER(T.Sample)), T.iptr(off), lin.ctypes.data_as(T.c_double_p),
                               C.cast(out.ctypes.data, C.POINTER(T.Preint))) == BAD_ARG
    pool = api.PreintStreams(ctx, 4)
    ids = np.array([1, 1], np.int32)
    assert L.vilo_preint_streams_reset(ctx.h, pool.h, 2, T.iptr(ids), C.cast(smp.ctypes.data, C.POINTER(T.Sample)), lin.ctypes.data_as(T.c_double_p)) == BAD_ARG
    ids = np.array([4], np.int32)
    assert L.vilo_preint_streams_read(ctx.h, pool.h, 1, T.iptr(ids), C.cast(out.ctypes.data, C.POINTER(T.Preint))) == BAD_ARG
    pool.close()


def test_malformed_windows_are_refused_with_a_message(ctx, cfg, ocfg):
    from cerberus_amd import api
    w = _win(cfg, ocfg, n_landmarks=12, seed=2)
    w.lm_start_frame[3] = 11                                   # start frame outside the 11-frame window
    assert _solve_rc(ctx, [w])[0] == BAD_ARG
    assert b"landmark" in api.lib().vilo_last_error(ctx.h)
    w = _win(cfg, ocfg, n_landmarks=12, seed=2)
    w.lm_start_frame[3] = 9                                    # its 11 - i%7 observations no longer fit
    assert _solve_rc(ctx, [w])[0] == BAD_ARG
    a, b = _win(cfg, ocfg, n_landmarks=12, seed=3), _win(cfg, ocfg, n_landmarks=12, seed=4)
    b.use_leg = 0
    rc, _ = _solve_rc(ctx, [a, b])
    assert rc == UNSUPPORTED and b"use_leg" in api.lib().vilo_last_error(ctx.h)
    w = _win(cfg, ocfg, n_landmarks=12, seed=5)
    w.prior.struct.block_id[0] = 5 * 16 + 3                    # a feature block in the prior: not a camera-side block
    rc, _ = _solve_rc(ctx, [w])
    assert rc == UNSUPPORTED and b"prior" in api.lib().vilo_last_error(ctx.h)


def test_a_non_finite_window_fails_alone(ctx, cfg, ocfg):
    """ceres::Solve on a problem whose initial evaluation is not finite returns FAILURE and leaves the parameters alone; the
    other windows of the batch are solved as if it was not there."""
    from cerberus_amd import api
    good, bad, ref = (_win(cfg, ocfg, n_landmarks=30, seed=6), _win(cfg, ocfg, n_landmarks=30, seed=7), _win(cfg, ocfg, n_landmarks=30, seed=6))
    bad.state_arrays()[0][4, 1] = np.nan
    before = bad.clone_state()
    s_ref = ctx.solve_windows([ref], api.default_solve_opts(True, 3))[0]
    rc, summ = _solve_rc(ctx, [good, bad])
    assert rc in (OK, NUMERIC)
    assert summ[1].termination == 2 and summ[1].num_successful == 0
    for a, b in zip(bad.state_arrays(), before):
        np.testing.assert_array_equal(a, b)
    assert summ[0].termination != 2 and summ[0].final_cost == s_ref.final_cost
    for a, b in zip(good.state_arrays(), ref.state_arrays()):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_a_bad_covariance_only_spoils_the_marginalisations_that_use_its_factor(ctx, cfg, ocfg):
    """A preintegration covariance that is not positive definite has no sqrt_info. MARGIN_OLD uses the IMU factor of interval 0 alone
    (estimator.cpp:1271-1297) and MARGIN_SECOND_NEW none (:1389-1410): a bad covariance in interval 7 must leave both priors exactly as
    they are without it (the reference would produce them), a bad one in interval 0 fails MARGIN_OLD (NUMERIC, no prior) and nothing else."""
    from cerberus_amd import api, _ctypes as T
    from cerberus_amd.synth import PriorData
    L = api.lib()

    def marg(w, mode):
        p = PriorData()
        d, s = w.desc(T)
        rc = L.vilo_marginalize(ctx.h, 1, C.byref(d), C.byref(s), mode, C.byref(p.struct))
        p.rebind()
        return rc, p

    cov0 = 33 + 961   # vilo_preint: 33 scalars, jacobian, covariance
    for mode in (0, 1):
        good = _win(cfg, ocfg, n_landmarks=30, seed=8)
        rc, pg = marg(good, mode)
        assert rc == OK and pg.struct.valid == 1
        w7 = _win(cfg, ocfg, n_landmarks=30, seed=8)
        w7.preint[7, cov0 + 5 * 31 + 5] = -1.0                       # interval 7: a negative diagonal entry
        rc, p7 = marg(w7, mode)
        assert rc == OK and p7.struct.valid == 1 and p7.n == pg.n
        np.testing.assert_array_equal(p7.J0_matrix(), pg.J0_matrix())
        np.testing.assert_array_equal(p7.r0[:p7.n], pg.r0[:pg.n])
        w0 = _win(cfg, ocfg, n_landmarks=30, seed=8)
        w0.preint[0, cov0 + 5 * 31 + 5] = -1.0                       # interval 0: the factor MARGIN_OLD marginalises
        rc, p0 = marg(w0, mode)
        if mode == 0:
            assert rc == NUMERIC and p0.struct.valid == 0
        else:
            assert rc == OK and p0.struct.valid == 1
            np.testing.assert_array_equal(p0.J0_matrix(), pg.J0_matrix())
    # the solve of such a window still fails alone, whichever interval it is
    w7 = _win(cfg, ocfg, n_landmarks=30, seed=8)
    w7.preint[7, cov0 + 5 * 31 + 5] = -1.0
    rc, summ = _solve_rc(ctx, [w7])
    assert summ[0].termination == 2


def test_mutated_window_tables_never_reach_the_packing_loops(ctx, cfg, ocfg):
    """A window that comes out of a file is untrusted (include/vilo_window_io.h bounds its counts by the file's size, nothing more): the tables
    vilo_batch_create indexes through — lm_obs_offset into the observations, lm_start_frame into the frames, the prior's block table into
    the prior and, on the device, into the assembly's LDS image — are checked before anything is packed. 400 seeded
    mutations of one table entry each: every call returns — OK with a finite result where the mutation happened to be harmless,
    BAD_ARG / UNSUPPORTED with a message otherwise — and a good window solved afterwards on the same context gives what it gave before."""
    from cerberus_amd import api
    ref = _win(cfg, ocfg, n_landmarks=30, seed=77)
    want = api.Context.solve_windows(ctx, [ref], api.default_solve_opts(True, 3))[0].final_cost
    rng = np.random.default_rng(11)
    outcomes = {OK: 0, BAD_ARG: 0, UNSUPPORTED: 0, NUMERIC: 0}
    for k in range(400):
        w = _win(cfg, ocfg, n_landmarks=30, seed=77)
        what = k % 5
        big = int(rng.choice([-1, -7, 1 << 20, 1 << 30, 12, 97, 255, 11, 86, 87]))
        if what == 0:
            w.lm_obs_offset[rng.integers(0, w.L + 1)] = big
        elif what == 1:
            w.lm_start_frame[rng.integers(0, w.L)] = big
        elif what == 2:
            w.prior.struct.block_idx[rng.integers(0, w.prior.struct.n_blocks)] = big
        elif what == 3:
            w.prior.struct.block_size[rng.integers(0, w.prior.struct.n_blocks)] = big
        else:
            w.prior.struct.block_id[rng.integers(0, w.prior.struct.n_blocks)] = big
        rc, summ = _solve_rc(ctx, [w])
        assert rc in outcomes, (k, what, big, rc)
        outcomes[rc] += 1
        if rc in (BAD_ARG, UNSUPPORTED):
            assert len(api.lib().vilo_last_error(ctx.h)) > 0
        if rc == OK:
            assert np.isfinite(summ[0].final_cost)
    assert outcomes[BAD_ARG] + outcomes[UNSUPPORTED] > 300, outcomes     # (most single-entry mutations break a table)
    again = _win(cfg, ocfg, n_landmarks=30, seed=77)
    assert ctx.solve_windows([again], api.default_solve_opts(True, 3))[0].final_cost == want
    print("mutated tables:", outcomes)
