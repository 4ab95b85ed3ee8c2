"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the
oracle on the same seeded inputs. Tolerances (FP64): factor residual/Jacobian <= 1e-11 relative to the
block's scale, whitened IMU quantities <= 1e-11 per entry / 1e-13 per row (measured 2e-12 / 6e-15), normal-equation
pieces <= 1e-9, states after an equal number of trust-region iterations <= 1e-8 (the measured differences, printed in
the assertion messages, are 1e-10 .. 1e-9: the elimination order of the speed / leg-bias chain and the summation order inside MFMA tiles differ
from the oracle's scalar loops)."""
import numpy as np
import pytest

from conftest import rand_pose
from oracle import oracle_py as O
from test_oracle_factors import _proj_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(cfg):
    from cerberus_amd import api
    c = api.Context(cfg, 0)
    yield c
    c.close()


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_eval_proj(ctx, ocfg, kind):
    rng = np.random.default_rng(100 + kind)
    n = 97
    setups = [_proj_setup(rng, kind) for _ in range(n)]
    obs = np.stack([s[0] for s in setups])
    nb = len(setups[0][1])
    params = [np.stack([s[1][k] for s in setups]) for k in range(nb)]
    r, Js = ctx.eval_proj(kind, obs, params)
    for i in range(n):
        r_o, J_o = O.eval_proj(kind, ocfg, obs[i], [p[i] for p in params])
        np.testing.assert_allclose(r[i], r_o, rtol=1e-12, atol=1e-11)
        for k in range(nb):
            np.testing.assert_allclose(Js[k][i], J_o[k], rtol=1e-11, atol=1e-11 * max(1.0, np.abs(J_o[k]).max()), err_msg="kind %d block %d" % (kind, k))
    r2, _ = ctx.eval_proj(kind, obs, params, want_jac=False)
    np.testing.assert_array_equal(r, r2)


def test_eval_imu_leg_and_imu(ctx, ocfg, small_window):
    w = small_window
    P = [w.pose[:-1], w.speed_bias[:-1], w.leg_bias[:-1], w.pose[1:], w.speed_bias[1:], w.leg_bias[1:]]
    r, Js = ctx.eval_imu_leg(w.preint, P)
    for k in range(10):
        r_o, J_o = O.eval_imu_leg(ocfg, w.preint[k], [p[k] for p in P])
        assert (np.abs(r[k] - r_o) / np.maximum(np.abs(r_o), 1e-12 * np.abs(r_o).max())).max() < 1e-11, ("imu_leg residual", k)
        Jg, Jo = np.hstack([Js[b][k] for b in range(6)]), np.hstack(J_o)
        assert (np.linalg.norm(Jg - Jo, axis=1) / np.linalg.norm(Jo, axis=1)).max() < 1e-13, ("imu_leg J", k)
    P4 = [w.pose[:-1], w.speed_bias[:-1], w.pose[1:], w.speed_bias[1:]]
    r, Js = ctx.eval_imu(w.preint_imu, P4)
    for k in range(10):
        r_o, J_o = O.eval_imu(ocfg, w.preint_imu[k], [p[k] for p in P4])
        assert (np.abs(r[k] - r_o) / np.maximum(np.abs(r_o), 1e-12 * np.abs(r_o).max())).max() < 1e-11
        Jg, Jo = np.hstack([Js[b][k] for b in range(4)]), np.hstack(J_o)
        assert (np.linalg.norm(Jg - Jo, axis=1) / np.linalg.norm(Jo, axis=1)).max() < 1e-13, ("imu J", k)


def test_sqrt_info_does_not_depend_on_the_records_wave_partner(ctx, small_window):
    """k_prepare_preint factors two records per wave (lanes 0 .. 30 and 32 .. 62). A record's sqrt_info — seen through the whitened
    residual and Jacobians — must be the same bit for bit alone in a wave (n = 1, the last record of an odd count), as the first or the
    second of a pair, and next to a partner whose covariance is not positive definite (which alone must fail the call)."""
    from cerberus_amd import api
    w = small_window
    P = [w.pose[:-1], w.speed_bias[:-1], w.leg_bias[:-1], w.pose[1:], w.speed_bias[1:], w.leg_bias[1:]]
    r_all, J_all = ctx.eval_imu_leg(w.preint, P)

    def sub(idx):
        return ctx.eval_imu_leg(w.preint[idx], [p[idx] for p in P])

    for idx in ([0], [3], [9], [0, 1, 2], [1, 2, 3], [9, 4, 7, 2, 5], [2, 2]):
        r, Js = sub(idx)
        for q, k in enumerate(idx):
            np.testing.assert_array_equal(r[q], r_all[k], err_msg="records %s, position %d" % (idx, q))
            for b in range(6):
                np.testing.assert_array_equal(Js[b][q], J_all[b][k])
    bad = w.preint.copy()
    bad[1, 33 + 961 + 5 * 31 + 5] = -1.0   # vilo_preint: 33 scalars, jacobian, covariance; a negative diagonal entry
    with pytest.raises(api.ViloError, match="positive definite"):
        ctx.eval_imu_leg(bad[:3], [p[:3] for p in P])
    with pytest.raises(api.ViloError, match="positive definite"):
        ctx.eval_imu_leg(bad[1:2], [p[1:2] for p in P])
    r, _ = ctx.eval_imu_leg(bad[2:5], [p[2:5] for p in P])
    np.testing.assert_array_equal(r, r_all[2:5])


def test_eval_prior_and_pose_plus(ctx, small_window):
    w = small_window
    rng = np.random.default_rng(3)
    blocks = w.prior.blocks()
    evals = []
    for _ in range(3):
        parts, off = [], 0
        for (bid, size, idx) in blocks:
            x0 = w.prior.x0[off:off + size]; off += size
            parts.append(O.pose_plus(x0, 0.01 * rng.normal(size=6)) if size == 7 else x0 + 0.01 * rng.normal(size=size))
        evals.append(parts)
    pc = np.stack([np.concatenate(p) for p in evals])
    r, J = ctx.eval_prior(w.prior, pc)
    for e, parts in enumerate(evals):
        r_o, J_o = O.eval_prior(w.prior.struct, parts)
        np.testing.assert_allclose(r[e], r_o, rtol=1e-12, atol=1e-12 * np.abs(r_o).max())
        np.testing.assert_allclose(J[e], np.concatenate(J_o, axis=1), rtol=0, atol=0)
    x = np.stack([rand_pose(rng) for _ in range(33)]); d = 0.1 * rng.normal(size=(33, 6))
    out = ctx.pose_plus(x, d)
    for i in range(33):
        np.testing.assert_allclose(out[i], O.pose_plus(x[i], d[i]), atol=2e-16)


def test_preintegrate(ctx, ocfg, small_window):
    w = small_window
    out = ctx.preintegrate(w.samples, w.sample_offsets, w.lin)
    for k in range(10):
        a, b = out[k], w.preint[k]
        np.testing.assert_allclose(a[:33], b[:33], rtol=1e-12, atol=1e-14, err_msg="state k=%d" % k)
        assert _rel(a[33:33 + 961], b[33:33 + 961]) < 1e-11, ("jacobian", k)
        assert _rel(a[33 + 961:], b[33 + 961:]) < 1e-10, ("covariance", k, _rel(a[33 + 961:], b[33 + 961:]))
        # entry-wise on the significant entries of the covariance
        ca, cb = a[33 + 961:], b[33 + 961:]
        big = np.abs(cb) > 1e-6 * np.abs(cb).max()
        assert np.abs(ca[big] / cb[big] - 1).max() < 1e-9
    out_i = ctx.preintegrate_imu(w.samples, w.sample_offsets, np.ascontiguousarray(w.lin[:, :6]))
    for k in range(10):
        np.testing.assert_allclose(out_i[k][:17], w.preint_imu[k][:17], rtol=1e-12, atol=1e-14)
        assert _rel(out_i[k][17:], w.preint_imu[k][17:]) < 1e-10


@pytest.fixture(scope="module")
def ctx_force(cfg):
    """contact_sensor_type 2 (config/go1_config/hardware_go1_vilo_config.yaml:30): foot forces instead of contact flags"""
    import copy
    from cerberus_amd import api
    c2 = copy.copy(cfg)
    c2.contact_sensor_type = 2
    c = api.Context(c2, 0)
    yield c
    c.close()


def _force_samples(samples, seed=0):
    rng = np.random.default_rng(seed)
    s = np.array(samples, copy=True)
    s[:, 31:35] = 15.0 + 140.0 * s[:, 31:35] + 4.0 * rng.normal(size=s[:, 31:35].shape)
    return s


def test_preintegrate_contact_sensor_type_2(ctx_force, ocfg, small_window):
    """imu_leg_integration_base.cpp:195-229, 300-317 on the device against the oracle (which tests/test_oracle_vs_reference.py
    pins against the reference's own imu_leg_integration_base.cpp for this model)."""
    import copy
    w = small_window
    o2 = copy.copy(ocfg)
    o2.contact_sensor_type = 2
    smp = _force_samples(w.samples, 3)
    out = ctx_force.preintegrate(smp, w.sample_offsets, w.lin)
    for k in range(10):
        a0, a1 = w.sample_offsets[k], w.sample_offsets[k + 1]
        b = O.preintegrate_imu_leg(o2, smp[a0:a1], w.lin[k])
        a = out[k]
        np.testing.assert_allclose(a[:33], b[:33], rtol=1e-12, atol=1e-14)
        assert _rel(a[33:33 + 961], b[33:33 + 961]) < 1e-11
        assert _rel(a[33 + 961:], b[33 + 961:]) < 1e-10
    assert _rel(out[0][33 + 961:], w.preint[0][33 + 961:]) > 1e-3   # not the flag-based noise model


@pytest.mark.parametrize("force", [False, True])
def test_streaming_preintegration_equals_the_batch_bitwise(ctx, ctx_force, small_window, force):
    """vilo_preint_streams_*: IMULegIntegrationBase objects resident in HBM, push_back()ed in arbitrary pieces (as
    Estimator::processIMULeg does per message, estimator.cpp:619-626), against vilo_preintegrate on the whole interval."""
    from cerberus_amd import api
    c = ctx_force if force else ctx
    w = small_window
    smp = _force_samples(w.samples, 5) if force else w.samples
    batch = c.preintegrate(smp, w.sample_offsets, w.lin)
    rng = np.random.default_rng(11)
    pool = api.PreintStreams(c, 16)
    ids = rng.permutation(16)[:10]
    first = np.stack([smp[w.sample_offsets[k]] for k in range(10)])
    pool.reset(ids, first, w.lin)
    np.testing.assert_array_equal(pool.read(ids)[:, 0], 0.0)                         # sum_dt of a fresh object
    cursor = [int(w.sample_offsets[k]) + 1 for k in range(10)]                        # element 0 is the constructor's measurement
    while any(cursor[k] < w.sample_offsets[k + 1] for k in range(10)):
        sel, chunks = [], []
        for k in range(10):
            left = int(w.sample_offsets[k + 1]) - cursor[k]
            if left and rng.random() < 0.7:
                n = int(rng.integers(1, min(left, 9) + 1))
                sel.append(k); chunks.append(smp[cursor[k]:cursor[k] + n]); cursor[k] += n
        if not sel:
            continue
        off = np.concatenate([[0], np.cumsum([len(ch) for ch in chunks])])
        pool.push(ids[sel], np.concatenate(chunks), off)
    np.testing.assert_array_equal(pool.read(ids), batch)
    # a reset object starts over; the others keep their state
    pool.reset(ids[:1], first[:1], w.lin[:1])
    again = pool.read(ids)
    assert again[0, 0] == 0.0 and np.array_equal(again[1:], batch[1:])
    pool.close()


def test_streaming_imu_only_preintegration_equals_the_batch_bitwise(ctx, small_window):
    """IntegrationBase objects (USE_LEG = 0) kept on the device, pushed in pieces, against vilo_preintegrate_imu."""
    from cerberus_amd import api
    w = small_window
    lin6 = np.ascontiguousarray(w.lin[:, :6])
    batch = ctx.preintegrate_imu(w.samples, w.sample_offsets, lin6)
    pool = api.PreintStreams(ctx, 10, imu_only=True)
    ids = np.arange(10)
    pool.reset(ids, np.stack([w.samples[w.sample_offsets[k]] for k in range(10)]), lin6)
    for piece in range(3):   # thirds of every interval
        chunks = []
        for k in range(10):
            a, b = int(w.sample_offsets[k]) + 1, int(w.sample_offsets[k + 1])
            cut = [a, a + (b - a) // 3, a + 2 * (b - a) // 3, b]
            chunks.append(w.samples[cut[piece]:cut[piece + 1]])
        pool.push(ids, np.concatenate(chunks), np.concatenate([[0], np.cumsum([len(c) for c in chunks])]))
    np.testing.assert_array_equal(pool.read(ids), batch)
    leg_pool = api.PreintStreams(ctx, 2)
    with pytest.raises(api.ViloError):   # kinds do not mix
        api.Context._check(ctx, api.lib().vilo_preint_streams_read_imu(ctx.h, leg_pool.h, 0 + 1, api.T.iptr(np.zeros(1, np.int32)), None))
    leg_pool.close(); pool.close()


def _cd_to_oracle(cd, F=11):
    if cd < 66:
        return 19 * (cd // 6) + cd % 6
    if cd < 72:
        return 19 * F + (cd - 66)
    if cd < 78:
        return 19 * F + 6 + (cd - 72)
    if cd == 78:
        return 19 * F + 12
    k, c = divmod(cd - 80, 13)
    return 19 * k + 6 + c


def _fresh(cfg, ocfg, **kw):
    from cerberus_amd import synth
    w = synth.make_window(cfg, **kw)
    O.fill_preint(ocfg, w)
    return w


@pytest.mark.parametrize("consts", [(0, 0, 1), (0, 1, 1), (1, 0, 0)])
def test_linearization_and_gauss_newton_step(ctx, cfg, ocfg, consts):
    """One linearisation: landmark blocks, gradient and the regularised Gauss-Newton solution against numpy on
    the oracle's normal equations."""
    from cerberus_amd import api
    w = _fresh(cfg, ocfg, n_landmarks=37, seed=5)
    w.leg_bias_const, w.ex_const, w.td_const = consts
    H, g, cost = O.window_normal_eq(ocfg, w)
    F, L = 11, w.L
    b = api.Batch(ctx, [w])
    b.solve(api.default_solve_opts(True, 1))
    perm = b.fetch(11).astype(int)
    E, gl, wl = b.fetch(1), b.fetch(2), b.fetch(3).reshape(80, L)
    cam_g, cam_dh2, cam_y, lm_y = b.fetch(4), b.fetch(5), b.fetch(6), b.fetch(8)
    st = b.fetch(10)
    lam_idx = 19 * F + 13 + perm
    np.testing.assert_allclose(E, np.diag(H)[lam_idx], rtol=1e-11)
    np.testing.assert_allclose(gl, g[lam_idx], rtol=1e-10, atol=1e-10 * np.abs(g).max())
    active = np.ones(224, bool)
    active[79] = False; active[223] = False
    if consts[1]:
        active[66:78] = False
    if consts[2]:
        active[78] = False
    if consts[0]:
        for k in range(11):
            active[80 + 13 * k + 9:80 + 13 * k + 13] = False
    cds = np.array([cd for cd in range(223) if cd != 79])
    oidx = np.array([_cd_to_oracle(cd) for cd in cds])
    for a in range(79):
        if not active[a] and np.all(wl[a] == 0.0):
            continue   # a constant block's coupling row may be left out of the solve passes (the compact visual form does not form td's)
        np.testing.assert_allclose(wl[a], H[_cd_to_oracle(a), lam_idx], rtol=1e-10, atol=1e-10 * np.abs(H[:, lam_idx]).max(), err_msg="w row %d" % a)
    act = active[cds]
    np.testing.assert_allclose(cam_g[cds][act], g[oidx][act], rtol=1e-9, atol=1e-9 * np.abs(g).max())
    assert np.all(cam_g[cds][~act] == 0)
    # regularised GN solution with Jacobi scaling (iteration 0): (H + mu Dhat^2) y = g on the active set
    sel = np.concatenate([oidx[act], lam_idx])
    Hs, gs = H[np.ix_(sel, sel)], g[sel]
    s = 1.0 / (1.0 + np.sqrt(np.diag(Hs)))
    dh2 = np.clip(s * s * np.diag(Hs), 1e-6, 1e32) / (s * s)
    y = np.linalg.solve(Hs + 1e-8 * np.diag(dh2), gs)
    ycam = y[:act.sum()]; ylm = y[act.sum():]
    np.testing.assert_allclose(cam_dh2[cds][act], dh2[:act.sum()], rtol=1e-10)
    scale = np.abs(y).max()
    assert np.abs(cam_y[cds][act] - ycam).max() < 1e-7 * scale, np.abs(cam_y[cds][act] - ycam).max() / scale
    assert np.abs(lm_y - ylm).max() < 1e-7 * scale
    # dogleg scalars
    gt2 = float(np.sum(gs * gs / dh2)); gnn2 = float(np.sum(dh2 * y * y)); gy = float(gs @ y)
    v = gs / dh2
    q = float(v @ Hs @ v)
    np.testing.assert_allclose([st[5], st[6], -st[7], st[8]], [gt2, gnn2, gy, q], rtol=1e-7)
    np.testing.assert_allclose(st[24], cost, rtol=1e-10)  # cost_trace[0] = cost at the initial point
    b.close()


@pytest.mark.parametrize("iters", [1, 3, 12])
def test_solve_parity(ctx, cfg, ocfg, iters):
    """States and cost trace after an equal number of trust-region iterations."""
    from cerberus_amd import api, synth
    w_g = _fresh(cfg, ocfg, n_landmarks=40, seed=11)
    w_o = _fresh(cfg, ocfg, n_landmarks=40, seed=11)
    summ = ctx.solve_windows([w_g], api.default_solve_opts(True, iters))[0]
    osum = O.solve_window(ocfg, w_o, O.default_opts(True, iters))
    ct_g = np.array([summ.cost_trace[i] for i in range(iters + 1)]); ct_o = np.array([osum.cost_trace[i] for i in range(iters + 1)])
    rt_g = np.array([summ.radius_trace[i] for i in range(iters + 1)]); rt_o = np.array([osum.radius_trace[i] for i in range(iters + 1)])
    print("cost gpu", ct_g, "\ncost orc", ct_o, "\nradius gpu", rt_g, "\nradius orc", rt_o)
    assert summ.iterations == osum.iterations and summ.num_successful == osum.num_successful
    # states after an equal number of iterations <= 1e-8
    np.testing.assert_allclose(ct_g, ct_o, rtol=1e-8)
    np.testing.assert_allclose(rt_g, rt_o, rtol=1e-8)
    worst = 0.0
    for name, a, bb in zip(["pose", "sb", "lb", "ex", "td", "lam"], w_g.state_arrays(), w_o.state_arrays()):
        err = np.abs(a - bb).max() / max(1.0, np.abs(bb).max())
        worst = max(worst, err)
        assert err < 1e-8, (name, err)
    print("MEASURED test_solve_parity[%d]: states %.2e, cost %.2e" % (iters, worst, np.abs(ct_g / ct_o - 1).max()))


def test_solve_skips_long_intervals(ctx, cfg, ocfg):
    """estimator.cpp:1118 / :1164: no IMU(-leg) factor for an interval with sum_dt > 10 s."""
    from cerberus_amd import api
    w_g = _fresh(cfg, ocfg, n_landmarks=40, seed=13)
    w_o = _fresh(cfg, ocfg, n_landmarks=40, seed=13)
    for w in (w_g, w_o):
        w.preint[4, 0] = 11.0   # sum_dt of interval (4, 5)
    summ = ctx.solve_windows([w_g], api.default_solve_opts(True, 4))[0]
    osum = O.solve_window(ocfg, w_o, O.default_opts(True, 4))
    assert summ.iterations == osum.iterations and summ.num_successful == osum.num_successful
    np.testing.assert_allclose(summ.final_cost, osum.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), np.abs(a - bb).max()   # measured: 1e-10 .. 1e-9


def _truncate(w, F):
    """First F frames of a synthetic window (the reference solves such windows while frame_count < WINDOW_SIZE: no prior
    yet, leg biases constant, estimator.cpp:1074-1082)."""
    keep = [l for l in range(w.L) if w.lm_start_frame[l] + 2 <= F]   # at least two observations left
    obs, st, off, sf = [], [], [0], []
    for l in keep:
        o0, o1 = w.lm_obs_offset[l], w.lm_obs_offset[l + 1]
        K = min(o1 - o0, F - w.lm_start_frame[l])
        obs.append(w.obs[o0:o0 + K]); st.append(w.obs_is_stereo[o0:o0 + K]); off.append(off[-1] + K); sf.append(w.lm_start_frame[l])
    w.L, w.n_obs = len(keep), off[-1]
    w.lm_start_frame = np.array(sf, np.int32); w.lm_obs_offset = np.array(off, np.int32)
    w.obs = np.ascontiguousarray(np.concatenate(obs)); w.obs_is_stereo = np.ascontiguousarray(np.concatenate(st))
    w.inv_depth = np.ascontiguousarray(w.inv_depth[keep])
    w.F = F
    w.prior.struct.valid = 0
    w.leg_bias_const = 1
    return w


@pytest.mark.parametrize("F", [4, 8])
def test_solve_partial_window(ctx, cfg, ocfg, F):
    from cerberus_amd import api
    w_g = _truncate(_fresh(cfg, ocfg, n_landmarks=60, seed=17), F)
    w_o = _truncate(_fresh(cfg, ocfg, n_landmarks=60, seed=17), F)
    summ = ctx.solve_windows([w_g], api.default_solve_opts(True, 4))[0]
    osum = O.solve_window(ocfg, w_o, O.default_opts(True, 4))
    assert summ.iterations == osum.iterations and summ.num_successful == osum.num_successful
    np.testing.assert_allclose(summ.final_cost, osum.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        assert np.abs(a[:F] - bb[:F]).max() < 1e-8 * max(1.0, np.abs(bb[:F]).max()) if a.shape[0] == 11 else np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max())


def test_solve_config2_window_and_tolerances(ctx, cfg, ocfg):
    """Full-size config-2 window (200 landmarks), Ceres termination rules enabled."""
    from cerberus_amd import api
    w_g = _fresh(cfg, ocfg, n_landmarks=200, seed=20260925)
    w_o = _fresh(cfg, ocfg, n_landmarks=200, seed=20260925)
    summ = ctx.solve_windows([w_g], api.default_solve_opts(False, 12))[0]
    osum = O.solve_window(ocfg, w_o, O.default_opts(False, 12))
    assert (summ.iterations, summ.termination) == (osum.iterations, osum.termination)
    np.testing.assert_allclose(summ.final_cost, osum.final_cost, rtol=1e-8)
    worst = max(np.abs(a - bb).max() / max(1.0, np.abs(bb).max()) for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()))
    print("MEASURED test_solve_config2_window_and_tolerances: states %.2e, cost %.2e" % (worst, abs(summ.final_cost / osum.final_cost - 1)))
    assert worst < 1e-8, worst


def test_headline_kernel_set_vs_oracle(ctx, cfg, ocfg):
    """The exact kernel set bench.py times at its default size, against the oracle: a batch of more than 1024 config-2 windows (seeds
    20260925 + i, 200 landmarks, 500 Hz: the bench's windows) takes the producer / consumer visual kernel, the compact assembly and the
    three-stage solver (k_chain / k_solve_mid / k_backsub) — below that size other forms of the same kernels run, so the smaller tests do
    not see this combination. 12 fixed iterations; first, middle and last window against the oracle at SURVEY 8(c)'s 1e-8."""
    from cerberus_amd import api, synth
    W = 1100
    ws = [synth.make_window(cfg, params=synth.default_params(n_landmarks=200, seed=20260925 + i)) for i in range(W)]
    ctx.preintegrate_windows(ws)
    assert api.lib().vilo_get_solver_form(ctx.h) == -1   # (the batch size chooses)
    opts = api.default_solve_opts(True, 12)
    api.lib().vilo_set_profiling(ctx.h, 1)
    try:
        summ = ctx.solve_windows(ws, opts)
        import ctypes as C
        ms = (C.c_double * 32)(); launches = (C.c_longlong * 32)()
        nk = api.lib().vilo_get_kernel_times(ctx.h, ms, launches, 32)
        api.lib().vilo_kernel_name.restype = C.c_char_p
        ran = {api.lib().vilo_kernel_name(i).decode() for i in range(nk) if launches[i] > 0}
    finally:
        api.lib().vilo_set_profiling(ctx.h, 0)
    assert {"k_chain", "k_solve_mid", "k_backsub", "k_assemble", "k_visual_linearize"} <= ran, ran
    assert all(s.iterations == 12 for s in summ)
    worst = 0.0
    for i in (0, W // 2, W - 1):
        w_o = synth.make_window(cfg, params=synth.default_params(n_landmarks=200, seed=20260925 + i))
        w_o.preint[...] = ws[i].preint   # (the records the GPU integrated: K1 has its own goldens)
        so = O.solve_window(ocfg, w_o, O.default_opts(True, 12))
        assert (summ[i].iterations, summ[i].num_successful) == (so.iterations, so.num_successful)
        np.testing.assert_allclose(summ[i].final_cost, so.final_cost, rtol=1e-8)
        for a, bb in zip(ws[i].state_arrays(), w_o.state_arrays()):
            if a.size:
                worst = max(worst, np.abs(a - bb).max() / max(1.0, np.abs(bb).max()))
    print("MEASURED test_headline_kernel_set_vs_oracle: states %.2e" % worst)
    assert worst < 1e-8, worst


def test_the_tail_of_a_headline_sized_batch(ctx, cfg, ocfg):
    """bench.py's default launch is 32 768 windows; the oracle comparisons above look at batches of 1100. Here a batch of that size: 253
    generated windows (the bench's seeds) repeated over the first 32 765 positions as twins (shared inputs, own state arrays — position p
    holds window p % 253, so replicas land in different workgroups, packed waves and arena chunks), three further seeds in the LAST three
    positions. 12 fixed iterations with the bench's kernel set. Checked: (1) every replica's final states and cost are bitwise those of the
    window's first position — a result does not depend on where in a launch the window sits, up to the last workgroup; (2) the first, a
    middle and the last three windows against the oracle at SURVEY 8(c)'s 1e-8; (3) every cost finite and within [0.1, 10] x the median
    (bench.py's `parity_sample.all_windows`)."""
    from cerberus_amd import api, synth
    W, U = 32768, 253
    uniq = [synth.make_window(cfg, params=synth.default_params(n_landmarks=200, seed=20260925 + i)) for i in range(U + 3)]
    ctx.preintegrate_windows(uniq)
    ws = [uniq[p] if p < U else uniq[p % U].twin() for p in range(W - 3)] + uniq[U:]
    b = api.Batch(ctx, ws)
    try:
        b.prepare()
        b.solve(api.default_solve_opts(True, 12))
        summ = b.download()
    finally:
        b.close()
    assert all(s.iterations == 12 for s in summ)
    costs = np.array([s.final_cost for s in summ])
    med = np.median(costs)
    assert np.isfinite(costs).all() and (costs >= 0.1 * med).all() and (costs <= 10 * med).all(), (costs.min(), med, costs.max())
    mism = 0
    for p in range(U, W - 3):
        same = summ[p].final_cost == summ[p % U].final_cost and all(np.array_equal(a, bb) for a, bb in zip(ws[p].state_arrays(), ws[p % U].state_arrays()))
        mism += 0 if same else 1
    assert mism == 0, "%d of %d replicas differ from their window's first position" % (mism, W - 3 - U)
    worst = 0.0
    for p in (0, 1, W // 2, W - 3, W - 2, W - 1):
        src = uniq[U + p - (W - 3)] if p >= W - 3 else uniq[p % U]
        seed = 20260925 + (U + p - (W - 3) if p >= W - 3 else p % U)
        w_o = synth.make_window(cfg, params=synth.default_params(n_landmarks=200, seed=seed))
        w_o.preint[...] = src.preint
        so = O.solve_window(ocfg, w_o, O.default_opts(True, 12))
        assert (summ[p].iterations, summ[p].num_successful) == (so.iterations, so.num_successful)
        np.testing.assert_allclose(summ[p].final_cost, so.final_cost, rtol=1e-8)
        for a, bb in zip(ws[p].state_arrays(), w_o.state_arrays()):
            if a.size:
                worst = max(worst, np.abs(a - bb).max() / max(1.0, np.abs(bb).max()))
    print("MEASURED test_the_tail_of_a_headline_sized_batch: states %.2e over positions 0, 1, W/2, W-3..W-1 of %d" % (worst, W))
    assert worst < 1e-8, worst


def test_path_parity_with_the_oracles_own_preintegration(ctx, cfg, ocfg):
    """The WHOLE path against the oracle with nothing shared but the inputs: contact preintegration of the raw samples (K1) + sqrt_info +
    12 trust-region iterations on the GPU, against the oracle integrating the same samples itself (O.fill_preint) and solving. The other solve
    tests hand both sides the same records (bench.py's parity_sample and test_headline_kernel_set_vs_oracle: the GPU's; most others: the
    oracle's), which pins the solver; K1 has its own goldens. What the two integrations differ by (1e-13 relative in the 31 x 31 covariance,
    whose inverse square root enters every IMU residual) shows in the states at the level of the solver's own rounding: measured 2.3e-10,
    tolerance 1e-8 like the solver-only tests; the measured value is printed."""
    from cerberus_amd import api, synth
    W = 4
    ws = [synth.make_window(cfg, params=synth.default_params(n_landmarks=200, seed=20260925 + i)) for i in range(W)]
    ctx.preintegrate_windows(ws)          # K1 on the GPU
    summ = ctx.solve_windows(ws, api.default_solve_opts(True, 12))
    worst = 0.0
    for i in range(W):
        w_o = synth.make_window(cfg, params=synth.default_params(n_landmarks=200, seed=20260925 + i))
        O.fill_preint(ocfg, w_o)          # the oracle's own integration of the same samples
        so = O.solve_window(ocfg, w_o, O.default_opts(True, 12))
        assert (summ[i].iterations, summ[i].num_successful) == (so.iterations, so.num_successful)
        np.testing.assert_allclose(summ[i].final_cost, so.final_cost, rtol=1e-8)
        for a, bb in zip(ws[i].state_arrays(), w_o.state_arrays()):
            if a.size:
                worst = max(worst, np.abs(a - bb).max() / max(1.0, np.abs(bb).max()))
    print("MEASURED test_path_parity_with_the_oracles_own_preintegration: states %.2e" % worst)
    assert worst < 1e-8, worst


def test_host_pipeline_gives_the_one_batch_answer(cfg, ocfg):
    """vilo_solve_windows cut into sub-batches over internal lanes (vilo_set_host_pipeline): bit for bit the one batch's states, costs and
    summaries — ragged windows, an uneven last share, sub-batches that by themselves would take the small batches' kernels —; a window with a NaN state fails alone; a refused window anywhere leaves every state
    of the call as it was (estimator.cpp:848-901: the call takes host arrays and either optimises them or does not)."""
    import ctypes as C
    from cerberus_amd import api, _ctypes as T
    specs = [dict(n_landmarks=40, seed=1), dict(n_landmarks=7, seed=2), dict(n_landmarks=90, seed=3), dict(n_landmarks=40, seed=4, with_prior=False)]
    base = [_fresh(cfg, ocfg, **s) for s in specs]
    N = 263        # (a call is only cut if as ONE batch it takes the full batch's kernel set: more than 256 windows with landmarks)

    def crowd():
        return [base[i % 4].twin() for i in range(N)]
    opts = api.default_solve_opts(True, 4)
    one, piped = crowd(), crowd()
    c1 = api.Context(cfg, 0); c1.set_host_pipeline(0, 0)
    c3 = api.Context(cfg, 0); c3.set_host_pipeline(3, 40)     # 263 windows -> 7 sub-batches of 38, ..., 35 over three lanes: small batches by themselves
    try:
        s1 = c1.solve_windows(one, opts)
        s3 = c3.solve_windows(piped, opts)
        for a, b in zip(s1, s3):
            assert bytes(a) == bytes(b)
        for wa, wb in zip(one, piped):
            for x, y in zip(wa.state_arrays(), wb.state_arrays()):
                np.testing.assert_array_equal(x, y)
        assert not np.array_equal(piped[0].pose, base[0].pose)       # (it did move)
        # a window with a non-finite observation fails by itself, in whatever sub-batch it sits
        sick = crowd()
        sick[17].pose[4, 1] = np.nan
        s = c3.solve_windows(sick, opts)
        assert s[17].termination == 2 and all(x.termination != 2 for i, x in enumerate(s) if i != 17)
        for i in (0, 16, 18, N - 1):
            for x, y in zip(sick[i].state_arrays(), one[i].state_arrays()):
                np.testing.assert_array_equal(x, y)
        # Estimator::optimization() as a whole (solve, gauge fix, marginalisation) takes the same lanes: states and priors of the one batch
        from cerberus_amd.synth import PriorData

        def optimise(c, ws):
            descs, states, summ, priors = (T.WindowDesc * N)(), (T.WindowState * N)(), (T.SolveSummary * N)(), (T.Prior * N)()
            outs = [PriorData() for _ in ws]
            for i, w in enumerate(ws):
                descs[i], states[i] = w.desc(T)
                priors[i] = outs[i].struct
            fl = (C.c_int * N)(*[i % 2 for i in range(N)])
            c._check(api.lib().vilo_optimize_windows(c.h, N, descs, states, C.byref(opts), fl, priors, summ))
            for i, o in enumerate(outs):
                o.struct = priors[i]
            return outs
        o1, o3 = crowd(), crowd()
        p1, p3 = optimise(c1, o1), optimise(c3, o3)
        assert sum(p.n > 0 for p in p1) >= N - N // 4 - 1     # (MARGIN_SECOND_NEW of a window without a prior leaves none)
        for wa, wb, pa, pb in zip(o1, o3, p1, p3):
            for x, y in zip(wa.state_arrays(), wb.state_arrays()):
                np.testing.assert_array_equal(x, y)
            assert pa.blocks() == pb.blocks() and pa.n == pb.n and pa.struct.valid == pb.struct.valid
            np.testing.assert_array_equal(pa.J0_matrix(), pb.J0_matrix())
            np.testing.assert_array_equal(pa.r0[:pa.n], pb.r0[:pb.n])
        # a refused window in the LAST sub-batch: the sub-batches before it had been solved and downloaded by then
        bad = crowd()
        before = [w.clone_state() for w in bad]
        bad[N - 2].lm_obs_offset = bad[N - 2].lm_obs_offset.copy(); bad[N - 2].lm_obs_offset[3] = bad[N - 2].lm_obs_offset[5] + 1
        descs, states, summ = (T.WindowDesc * N)(), (T.WindowState * N)(), (T.SolveSummary * N)()
        for i, w in enumerate(bad):
            descs[i], states[i] = w.desc(T)
        rc = api.lib().vilo_solve_windows(c3.h, N, descs, states, C.byref(opts), summ)
        assert rc == -2 and b"landmark observation table" in api.lib().vilo_last_error(c3.h)
        for w, k in zip(bad, before):
            for x, y in zip(w.state_arrays(), k):
                np.testing.assert_array_equal(x, y)
    finally:
        c1.close(); c3.close()


def test_host_records_arrive_as_what_the_solver_reads(ctx, cfg):
    """Host windows' IMU-leg records cross PCIe as 739 of their 1955 doubles (scalars, bias columns of the Jacobian's first 21 rows, upper
    triangle of the covariance) and are laid out as records again on the device: every entry the preparation reads is the caller's, the
    covariance's lower triangle mirrors the upper one, what no factor reads is zero — for the last window of a batch as for the first."""
    from cerberus_amd import api, synth, _ctypes as T
    ws = [synth.make_window(cfg, n_landmarks=20, seed=5100 + i) for i in range(70)]
    ctx.preintegrate_windows(ws)
    b = api.Batch(ctx, ws)
    try:
        for wi in (0, 37, 69):
            dev = b.fetch(13, wi).reshape(10, T.PREINT_DOUBLES)
            host = ws[wi].preint
            np.testing.assert_array_equal(dev[:, :33], host[:, :33])
            Jd, Jh = dev[:, 33:33 + 961].reshape(10, 31, 31), host[:, 33:33 + 961].reshape(10, 31, 31)
            np.testing.assert_array_equal(Jd[:, :21, 21:], Jh[:, :21, 21:])
            assert np.all(Jd[:, 21:, :] == 0) and np.all(Jd[:, :, :21] == 0) and np.abs(Jh[:, :21, :21]).max() > 0
            Cd, Ch = dev[:, 33 + 961:].reshape(10, 31, 31), host[:, 33 + 961:].reshape(10, 31, 31)
            iu = np.triu_indices(31)
            for k in range(10):
                np.testing.assert_array_equal(Cd[k][iu], Ch[k][iu])
                np.testing.assert_array_equal(Cd[k], Cd[k].T)
    finally:
        b.close()


def test_host_pipeline_under_changing_settings_and_refused_windows(cfg, ocfg):
    """One context, forty calls: the number of windows, the lanes and the share size change from call to call (lanes are created once and
    reused), every third call carries a refused window at a random position (first share, last share, anywhere) — the call must say so
    and leave every state as it was — and every good call must give the one batch's answer bit for bit."""
    import ctypes as C
    from cerberus_amd import api, _ctypes as T
    rng = np.random.default_rng(99)
    base = [_fresh(cfg, ocfg, n_landmarks=int(n), seed=700 + i, with_prior=bool(i % 3)) for i, n in enumerate((5, 17, 33, 64, 9))]
    opts = api.default_solve_opts(True, 3)
    c1 = api.Context(cfg, 0); c1.set_host_pipeline(0, 0)
    cp = api.Context(cfg, 0)
    try:
        ref = {}
        for call in range(40):
            N = int(rng.choice([257, 300, 513, 777]))
            lanes, sub = int(rng.integers(2, 7)), int(rng.integers(16, 130))
            cp.set_host_pipeline(lanes, sub)
            ws = [base[(i * 7 + call) % 5].twin() for i in range(N)]
            before = [w.clone_state() for w in ws]
            bad_at = None
            if call % 3 == 2:
                bad_at = int(rng.choice([0, N - 1, int(rng.integers(0, N))]))
                ws[bad_at].lm_start_frame = ws[bad_at].lm_start_frame.copy()
                ws[bad_at].lm_start_frame[0] = 11                    # outside the window
            descs, states, summ = (T.WindowDesc * N)(), (T.WindowState * N)(), (T.SolveSummary * N)()
            for i, w in enumerate(ws):
                descs[i], states[i] = w.desc(T)
            rc = api.lib().vilo_solve_windows(cp.h, N, descs, states, C.byref(opts), summ)
            if bad_at is not None:
                assert rc == -2, (call, rc)
                for w, k in zip(ws, before):
                    for x, y in zip(w.state_arrays(), k):
                        np.testing.assert_array_equal(x, y)
                continue
            assert rc == 0, (call, rc, api.lib().vilo_last_error(cp.h))
            for i in (0, N // 2, N - 1, int(rng.integers(0, N))):
                # the window's answer in a one-batch call that takes the solver form a call of N windows takes (eight waves per window up
                # to 512 windows, the single wave beyond: the two agree to rounding, not bitwise)
                key = ((i * 7 + call) % 5, N > 512)
                if key not in ref:
                    one = [base[key[0]].twin() for _ in range(600 if key[1] else 300)]
                    c1.solve_windows(one, opts)
                    ref[key] = one[0].clone_state()
                for x, y in zip(ws[i].state_arrays(), ref[key]):
                    np.testing.assert_array_equal(x, y)
    finally:
        c1.close(); cp.close()


def test_host_pipeline_on_a_random_crowd(cfg):
    """300 windows of 0 .. 400 landmarks, with and without prior, solved to convergence (not a fixed iteration count): the call cut over
    2 .. 6 lanes into shares of 7 .. 64 windows — shares that as batches of their own would take the small assembly and the frame-parallel
    visual form, which agree with the full batch's kernels to rounding only (3e-10) — gives the one batch's states and summaries bit for bit."""
    from cerberus_amd import api, synth
    rng = np.random.default_rng(5)
    N = 300
    ws = [synth.make_window(cfg, n_landmarks=int(rng.integers(0, 400)), seed=3000 + i, with_prior=bool(rng.integers(0, 4) != 0)) for i in range(N)]
    c1 = api.Context(cfg, 0)
    c1.set_host_pipeline(0, 0)
    try:
        c1.preintegrate_windows(ws)
        s0 = [w.clone_state() for w in ws]
        opts = api.default_solve_opts(False, 12)
        a = c1.solve_windows(ws, opts)
        ra = [w.clone_state() for w in ws]
        assert len({s.iterations for s in a}) > 1          # (the windows do not all stop at the same iteration)
        for lanes, sub in ((2, 7), (3, 16), (4, 64), (6, 11)):
            for w, s in zip(ws, s0):
                w.set_state(s)
            c = api.Context(cfg, 0)
            c.set_host_pipeline(lanes, sub)
            try:
                b = c.solve_windows(ws, opts)
            finally:
                c.close()
            for i, (w, r) in enumerate(zip(ws, ra)):
                assert bytes(a[i]) == bytes(b[i]), (lanes, sub, i)
                for x, y in zip(w.state_arrays(), r):
                    np.testing.assert_array_equal(x, y)
    finally:
        c1.close()


def test_batch_of_windows_matches_single(ctx, cfg, ocfg):
    """Independent windows in one batch give the same answers as solved alone; ragged landmark counts,
    a window without prior, >64 landmarks per start frame (multi-chunk groups)."""
    from cerberus_amd import api
    specs = [dict(n_landmarks=40, seed=1), dict(n_landmarks=7, seed=2), dict(n_landmarks=500, seed=3), dict(n_landmarks=40, seed=4, with_prior=False)]
    ws = [_fresh(cfg, ocfg, **s) for s in specs]
    singles = [_fresh(cfg, ocfg, **s) for s in specs]
    opts = api.default_solve_opts(True, 5)
    sb = ctx.solve_windows(ws, opts)
    for w1, s1 in zip(singles, sb):
        s = ctx.solve_windows([w1], opts)[0]
        assert s.final_cost == s1.final_cost    # no atomics anywhere on the path: a window's result does not depend on its batch
    for w_b, w_s in zip(ws, singles):
        for a, bb in zip(w_b.state_arrays(), w_s.state_arrays()):
            np.testing.assert_array_equal(a, bb)
    # and against the oracle for the ragged ones
    for spec, w_b in zip(specs[1:], ws[1:]):
        w_o = _fresh(cfg, ocfg, **spec)
        O.solve_window(ocfg, w_o, O.default_opts(True, 5))
        for a, bb in zip(w_b.state_arrays(), w_o.state_arrays()):
            assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), np.abs(a - bb).max()   # measured: 1e-10 .. 1e-9


@pytest.mark.parametrize("td_const", [1, 0])
def test_small_and_large_batches_linearise_identically(ctx, cfg, ocfg, td_const):
    """Up to 256 packed waves a batch is linearised frame-parallel (k_visual_linearize_tpar + k_visual_reduce: one workgroup per
    (packed wave, frame) so that a few windows still fill the chip), above that by one wave per packed wave. Same window, both forms,
    bit for bit: the LINEARISATION of a window does not depend on how many windows share the call. (The per-linearisation solver has several
    forms chosen by batch size that agree to rounding, not bitwise — tests/test_solver_forms.py; vilo_set_solver_form pins one where a
    robot must get the same answer alone and inside a fleet of any size.) td_const = 1 (estimate_td: 0, the reference's
    configuration): a left-camera factor's rows take one Gram tile; td_const = 0: all three — the second case also against the oracle."""
    from cerberus_amd import api
    opts = api.default_solve_opts(False, 12)

    def fresh(seed):
        w = _fresh(cfg, ocfg, n_landmarks=60, seed=seed)
        w.td_const = td_const
        return w
    alone = [fresh(300 + i) for i in range(3)]
    for w in alone:
        ctx.solve_windows([w], opts)                                   # 2 packed waves: frame-parallel form
    crowd = [fresh(300 + i) for i in range(3)] + [fresh(900 + i) for i in range(160)]   # > 256 packed waves: walking form
    ctx.solve_windows(crowd, opts)
    for w1, w2 in zip(alone, crowd[:3]):
        for a, b in zip(w1.state_arrays(), w2.state_arrays()):
            np.testing.assert_array_equal(a, b)
    if not td_const:
        w_o = fresh(300)
        O.solve_window(ocfg, w_o, O.default_opts(False, 12))
        for name, a, bb in zip(["pose", "sb", "lb", "ex", "td", "lam"], crowd[0].state_arrays(), w_o.state_arrays()):
            assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), (name, np.abs(a - bb).max())


def test_size_independent_properties(ctx, cfg):
    """Config-2 sized batch: cost never increases along accepted steps, rejected steps leave the state
    untouched, and re-solving from the solution gains almost nothing."""
    from cerberus_amd import api, synth
    ws = [synth.make_window(cfg, n_landmarks=200, seed=900 + i) for i in range(8)]
    ctx.preintegrate_windows(ws)
    opts = api.default_solve_opts(True, 12)
    summ = ctx.solve_windows(ws, opts)
    for s in summ:
        ct = np.array([s.cost_trace[i] for i in range(s.iterations + 1)])
        assert np.all(np.diff(ct) <= 1e-12 * ct[0])
        assert s.final_cost < s.initial_cost
    summ2 = ctx.solve_windows(ws, api.default_solve_opts(True, 2))
    for s, s2 in zip(summ, summ2):
        assert abs(s2.initial_cost - s.final_cost) <= 1e-9 * s.final_cost
        assert s2.final_cost >= s2.initial_cost * (1 - 2e-3)   # 12 iterations is not full convergence


@pytest.mark.parametrize("mode", [0, 1])
def test_marginalize(ctx, cfg, ocfg, mode):
    """MarginalizationInfo::marginalize: only J0^T J0 and J0^T r0 are ordering/sign invariant (SURVEY §8a note 12)."""
    from cerberus_amd.synth import PriorData
    w = _fresh(cfg, ocfg, n_landmarks=60, seed=21)
    pg, po = PriorData(), PriorData()
    ctx.marginalize(w, mode, pg)
    rc, m, A, bvec = O.marginalize(ocfg, w, mode, po, want_A=True)
    assert rc == 0 and pg.struct.valid == 1
    assert pg.blocks() == po.blocks()
    n = pg.n
    assert n == po.n and n in (80, 86)
    np.testing.assert_array_equal(pg.x0[:7 * 40], po.x0[:7 * 40])
    Jg, Jo = pg.J0_matrix(), po.J0_matrix()
    Ag, Ao = Jg.T @ Jg, Jo.T @ Jo
    assert np.abs(Ag - Ao).max() < 1e-6 * np.abs(Ao).max(), np.abs(Ag - Ao).max() / np.abs(Ao).max()  # eps * cond(Amm): both sides use the eigen pseudo-inverse
    bg, bo = Jg.T @ pg.r0[:n], Jo.T @ po.r0[:n]
    assert np.abs(bg - bo).max() < 1e-6 * np.abs(bo).max()
    # independent numpy Schur complement of the oracle's A, b
    Amm, Amr, Arr = A[:m, :m], A[:m, m:], A[m:, m:]
    Ai = np.linalg.pinv(0.5 * (Amm + Amm.T), rcond=0, hermitian=True)
    As = Arr - Amr.T @ Ai @ Amr
    assert np.abs(Ag - As).max() < 1e-6 * np.abs(As).max()
    # per kept block pair, every entry in units of the blocks' own diagonals, against the 60-digit Schur complement of the oracle's A, b
    # (tests/marg_exact.py; tests/test_golden.py::test_what_fp64_inputs_allow_for_margin_old measures the floor FP64 inputs set: ~6e-6)
    from marg_exact import block_table, exact_schur, scaled_errors
    He, ge = exact_schur(A, bvec, m)
    eh, eb, _, _ = scaled_errors(pg, He, ge, block_table(po))
    oh, ob, _, _ = scaled_errors(po, He, ge, block_table(po))
    print("MEASURED test_marginalize mode %d: HIP vs exact Schur complement H %.2e, b %.2e of the blocks' diagonals (oracle: %.2e, %.2e)" % (mode, eh, eb, oh, ob))
    assert max(eh, eb) < (2e-5 if mode == 0 else 1e-11), (eh, eb)


@pytest.mark.parametrize("mode", [0, 1])
def test_marginalize_block_elimination_equals_the_eigen_pseudo_inverse_path(ctx, cfg, ocfg, mode, monkeypatch):
    """k_marginalize_lds eliminates landmarks, then the dense frame-0 dims by Cholesky, after certifying lambda_min(Amm) > eps;
    k_marginalize (the fallback for rank-deficient Amm) forms the eps-thresholded eigen pseudo-inverse of the full Amm as
    marginalization_factor.cpp:281-286 does. On a well-conditioned window both give the same prior information."""
    from cerberus_amd import api
    from cerberus_amd.synth import PriorData
    w = _fresh(cfg, ocfg, n_landmarks=60, seed=23)
    pf, pg = PriorData(), PriorData()
    ctx.marginalize(w, mode, pf)
    assert api.lib().vilo_debug_marg_general_count(ctx.h) == 0
    monkeypatch.setenv("VILO_MARG_GENERAL", "1")
    ctx.marginalize(w, mode, pg)
    assert api.lib().vilo_debug_marg_general_count(ctx.h) == 1
    assert pf.blocks() == pg.blocks() and pf.n == pg.n
    Jf, Jg = pf.J0_matrix(), pg.J0_matrix()
    Af, Ag = Jf.T @ Jf, Jg.T @ Jg
    assert np.abs(Af - Ag).max() < 1e-6 * np.abs(Ag).max()
    bf, bg = Jf.T @ pf.r0[:pf.n], Jg.T @ pg.r0[:pg.n]
    assert np.abs(bf - bg).max() < 1e-6 * np.abs(bg).max()
    np.testing.assert_allclose(pf.r0[:pf.n] @ pf.r0[:pf.n], pg.r0[:pg.n] @ pg.r0[:pg.n], rtol=1e-6)


def test_marginalize_rank_deficient_block_takes_the_eigen_path(ctx, cfg, ocfg):
    """A genuinely rank-deficient Amm (three zero eigenvalues, see conftest.rank_deficient_window): k_marginalize_lds cannot certify
    lambda_min(Amm) > eps, the window goes to k_marginalize (eps-thresholded eigen pseudo-inverse, marginalization_factor.cpp:281-286),
    and the prior it leaves is the oracle's (which tests/test_oracle_vs_reference.py pins against the compiled reference on this window)."""
    from conftest import rank_deficient_window
    from cerberus_amd import api
    from cerberus_amd.synth import PriorData
    w = rank_deficient_window(cfg, ocfg)
    pg, po = PriorData(), PriorData()
    ctx.marginalize(w, 0, pg)
    assert api.lib().vilo_debug_marg_general_count(ctx.h) == 1
    rc, m, _, _ = O.marginalize(ocfg, w, 0, po)
    assert rc == 0 and m == 7 and pg.blocks() == po.blocks() and pg.n == po.n == 19
    Jg, Jo = pg.J0_matrix(), po.J0_matrix()
    Ag, Ao = Jg.T @ Jg, Jo.T @ Jo
    assert np.abs(Ag - Ao).max() < 1e-8 * np.abs(Ao).max(), np.abs(Ag - Ao).max() / np.abs(Ao).max()
    bg, bo = Jg.T @ pg.r0[:pg.n], Jo.T @ po.r0[:po.n]
    assert np.abs(bg - bo).max() < 1e-8 * np.abs(bo).max(), np.abs(bg - bo).max() / np.abs(bo).max()


def test_marginalize_many_dropped_landmarks(ctx, cfg, ocfg):
    """700 landmarks, 100 of them anchored in frame 0: the landmark elimination of k_marginalize_lds walks four 32-wide tiles."""
    from cerberus_amd import api
    from cerberus_amd.synth import PriorData
    w = _fresh(cfg, ocfg, n_landmarks=700, seed=29)
    assert int((w.lm_start_frame == 0).sum()) == 100
    pg, po = PriorData(), PriorData()
    ctx.marginalize(w, 0, pg)
    assert api.lib().vilo_debug_marg_general_count(ctx.h) == 0
    rc, m, _, _ = O.marginalize(ocfg, w, 0, po)
    assert rc == 0 and m == 119 and pg.blocks() == po.blocks() and pg.n == po.n == 86
    Jg, Jo = pg.J0_matrix(), po.J0_matrix()
    Ag, Ao = Jg.T @ Jg, Jo.T @ Jo
    assert np.abs(Ag - Ao).max() < 1e-6 * np.abs(Ao).max()
    bg, bo = Jg.T @ pg.r0[:86], Jo.T @ po.r0[:86]
    assert np.abs(bg - bo).max() < 1e-6 * np.abs(bo).max()


def test_marginalize_seed_sweep(ctx, cfg, ocfg):
    """A' -> J0, r0 by pivoted Cholesky + one-sided Jacobi (prior_factor_lds) over many windows: with and without a prior (the latter
    semi-definite: fewer columns than dimensions), few and many landmarks, both flags, every one against the oracle's eigen route;
    and J0 r0 consistent with itself: J0^T J0 is what the rows say, the kept rows are mutually orthogonal (they are sqrt(S) v^T)."""
    from cerberus_amd.synth import PriorData
    worst = 0.0
    for k in range(24):
        kw = dict(n_landmarks=(12, 60, 200, 700)[k % 4], seed=4000 + k, with_prior=(k % 3 != 0))
        w = _fresh(cfg, ocfg, **kw)
        for mode in ((0, 1) if kw["with_prior"] else (0,)):   # MARGIN_SECOND_NEW without a prior has nothing to carry over
            pg, po = PriorData(), PriorData()
            ctx.marginalize(w, mode, pg)
            rc = O.marginalize(ocfg, w, mode, po)[0]
            assert rc == 0 and pg.struct.valid == 1 and pg.blocks() == po.blocks() and pg.n == po.n, (kw, mode)
            n = pg.n
            Jg, Jo = pg.J0_matrix(), po.J0_matrix()
            Ag, Ao = Jg.T @ Jg, Jo.T @ Jo
            e_a = np.abs(Ag - Ao).max() / np.abs(Ao).max()
            bg, bo = Jg.T @ pg.r0[:n], Jo.T @ po.r0[:n]
            e_b = np.abs(bg - bo).max() / np.abs(bo).max()
            worst = max(worst, e_a, e_b)
            # eps * cond(Amm) with cond(Amm) up to 1e12 here: the oracle inverts Amm through its eigen pseudo-inverse like the reference; against a
            # numpy Schur complement of the oracle's own A the two sides sit equally far (tools: 1e-7 .. 6e-6)
            assert e_a < 1e-5 and e_b < 1e-5, (kw, mode, e_a, e_b)
            G = Jg @ Jg.T                       # rows of J0 = sqrt(S_i) v_i^T: orthogonal, |row|^2 = S_i
            d = np.sqrt(np.maximum(np.diag(G), 1e-300))
            C_ = np.abs(G / np.outer(d, d) - np.eye(n))[np.ix_(np.diag(G) > 0, np.diag(G) > 0)]
            assert C_.max() < 1e-6, (kw, mode, C_.max())
    print("worst relative deviation of J0^T J0 / J0^T r0 from the oracle over the sweep: %.2e" % worst)


def test_prior_factor_form_carries_the_same_information(cfg, ocfg):
    """vilo_set_prior_form(VILO_PRIOR_FACTOR): where A' has full rank and lambda_min(A') > eps is certified, the marginalisation leaves the
    pivoted Cholesky factor (J0 = X^T, r0 = X^-1 b) instead of sqrt(S) V^T (marginalization_factor.cpp:297-305) — an orthogonal
    transformation of it. Everything a solve takes from the prior is the same to rounding: J0^T J0, J0^T r0 per entry in units of the
    diagonal, |r0|^2; the rows themselves are NOT orthogonal any more (that is how the test knows the path was taken). A window without
    a prior (A' semi-definite: the gauge directions, as in every real sequence) gives a factor of n - 4 columns and as many non-zero rows
    as the eigen form keeps. (What a solve makes of either form: tests/test_rosbag.py replays a sequence in both, every window against
    the oracle at 1e-8.)"""
    from cerberus_amd import api
    from cerberus_amd.synth import PriorData
    ce, cf = api.Context(cfg, 0), api.Context(cfg, 0)
    cf.set_prior_form("factor")
    assert api.lib().vilo_set_prior_form(cf.h, 7) != 0
    try:
        worst = 0.0
        for k in range(8):
            kw = dict(n_landmarks=(12, 60, 200, 700)[k % 4], seed=5000 + k, with_prior=(k % 4 != 3))
            w = _fresh(cfg, ocfg, **kw)
            for mode in ((0, 1) if kw["with_prior"] else (0,)):
                pe, pf = PriorData(), PriorData()
                ce.marginalize(w, mode, pe)
                cf.marginalize(w, mode, pf)
                assert pe.blocks() == pf.blocks() and pe.n == pf.n
                n = pe.n
                Je, Jf = pe.J0_matrix(), pf.J0_matrix()
                if not kw["with_prior"]:
                    kept_e, kept_f = int((np.abs(Je).max(axis=1) > 0).sum()), int((np.abs(Jf).max(axis=1) > 0).sum())
                    assert kept_f <= n - 4 and abs(kept_e - kept_f) <= 2, (kept_e, kept_f, n)
                He, Hf = Je.T @ Je, Jf.T @ Jf
                d = np.sqrt(np.diag(He))
                e_h = (np.abs(He - Hf) / np.outer(d, d)).max()
                e_b = (np.abs(Je.T @ pe.r0[:n] - Jf.T @ pf.r0[:n]) / d).max() / max(1.0, np.abs(Je.T @ pe.r0[:n] / d).max())
                e_c = abs(pe.r0[:n] @ pe.r0[:n] - pf.r0[:n] @ pf.r0[:n]) / (pe.r0[:n] @ pe.r0[:n])
                worst = max(worst, e_h, e_b, e_c)
                # semi-definite A': the eigen form projects b orthogonally onto the eigenvectors it keeps, the factor form reproduces b on the
                # pivot rows; b's component along the four gauge directions is zero in exact arithmetic and rounding noise here (measured: up to
                # 9e-6 of the largest whitened gradient entry at 700 landmarks), which is what the two forms differ by
                tol = (1e-11, 1e-9, 1e-9) if kw["with_prior"] else (1e-9, 1e-4, 1e-6)
                assert e_h < tol[0] and e_b < tol[1] and e_c < tol[2], (kw, mode, e_h, e_b, e_c)
                nz = np.abs(Jf).max(axis=1) > 0
                G = Jf[nz] @ Jf[nz].T
                dg = np.sqrt(np.diag(G))
                assert np.abs(G / np.outer(dg, dg) - np.eye(int(nz.sum()))).max() > 1e-3, "the factor form's rows are not mutually orthogonal"
        print("MEASURED factor form vs eigen form of the prior: worst deviation %.2e" % worst)
    finally:
        ce.close(); cf.close()


def test_optimize_windows_is_solve_plus_gauge_fix_plus_marginalize(ctx, cfg, ocfg):
    """vilo_optimize_windows (one device batch) against the three separate entry points, and against the oracle's chain."""
    import ctypes as C
    from cerberus_amd import api, _ctypes as T
    from cerberus_amd.synth import PriorData
    opts = api.default_solve_opts(True, 5)
    ws = [_fresh(cfg, ocfg, n_landmarks=50, seed=60 + i) for i in range(3)]
    ref = [_fresh(cfg, ocfg, n_landmarks=50, seed=60 + i) for i in range(3)]
    flags = [0, 1, 0]
    # separate calls
    p_sep = [PriorData() for _ in ref]
    for w, f, p in zip(ref, flags, p_sep):
        before = w.clone_state()
        ctx.solve_windows([w], opts)
        ctx.gauge_fix(before, w)
        ctx.marginalize(w, f, p)
    # one call
    W = len(ws)
    descs = (T.WindowDesc * W)(); states = (T.WindowState * W)(); summ = (T.SolveSummary * W)(); priors = (T.Prior * W)()
    p_one = [PriorData() for _ in ws]
    for i, w in enumerate(ws):
        descs[i], states[i] = w.desc(T)
        priors[i] = p_one[i].struct
    fl = (C.c_int * W)(*flags)
    ctx._check(api.lib().vilo_optimize_windows(ctx.h, W, descs, states, C.byref(opts), fl, priors, summ))
    for i in range(W):
        for a, b in zip(ws[i].state_arrays(), ref[i].state_arrays()):
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())     # the gauge fix runs on the batch's state block
        p_one[i].struct = priors[i]
        assert p_one[i].blocks() == p_sep[i].blocks()
        Ja, Jb = p_one[i].J0_matrix(), p_sep[i].J0_matrix()
        assert np.abs(Ja.T @ Ja - Jb.T @ Jb).max() < 1e-9 * np.abs(Jb.T @ Jb).max()
    # the oracle's chain on the first window
    wo = _fresh(cfg, ocfg, n_landmarks=50, seed=60)
    before = wo.clone_state()
    O.solve_window(ocfg, wo, O.default_opts(True, 5))
    O.gauge_fix(before, wo)
    po = PriorData()
    assert O.marginalize(ocfg, wo, 0, po)[0] == 0
    Jg, Jo = p_one[0].J0_matrix(), po.J0_matrix()
    assert np.abs(Jg.T @ Jg - Jo.T @ Jo).max() < 1e-6 * np.abs(Jo.T @ Jo).max()


def test_marginalize_config2_batch(ctx, cfg, ocfg):
    """256 config-2 windows (200 landmarks) in one vilo_marginalize call, both flags: every window takes the LDS path, the
    information matrices are symmetric positive semi-definite with the 4-dim gauge null space cut off at eps, and three of them agree
    with the oracle."""
    import ctypes as C
    from cerberus_amd import api, synth, _ctypes as T
    from cerberus_amd.synth import PriorData
    W = 256
    ws = [synth.make_window(cfg, n_landmarks=200, seed=900 + i) for i in range(W)]
    ctx.preintegrate_windows(ws)
    descs = (T.WindowDesc * W)(); states = (T.WindowState * W)(); priors = (T.Prior * W)()
    outs = [PriorData() for _ in range(W)]
    for i, w in enumerate(ws):
        descs[i], states[i] = w.desc(T)
        priors[i] = outs[i].struct
    for mode, n_expect in ((0, 86), (1, 80)):
        ctx._check(api.lib().vilo_marginalize(ctx.h, W, descs, states, mode, priors))
        assert api.lib().vilo_debug_marg_general_count(ctx.h) == 0
        for i in range(W):
            outs[i].struct = priors[i]
            assert priors[i].valid == 1 and priors[i].n == n_expect
        for i in (0, 100, 255):
            J = outs[i].J0_matrix()
            A = J.T @ J
            assert np.isfinite(J).all() and np.isfinite(outs[i].r0[:n_expect]).all()
            ev = np.linalg.eigvalsh(A)
            assert ev.min() > -1e-9 * ev.max() and (np.abs(J).sum(axis=1) == 0).sum() <= 6   # rows of dropped eigenvalues are zero
            po = PriorData()
            O.fill_preint(ocfg, ws[i])
            assert O.marginalize(ocfg, ws[i], mode, po)[0] == 0
            Jo = po.J0_matrix()
            assert np.abs(A - Jo.T @ Jo).max() < 1e-5 * np.abs(Jo.T @ Jo).max()


def test_marginalized_prior_feeds_next_solve(ctx, cfg, ocfg):
    """Prior produced by GPU marginalisation of one window drives the solve of the next (GPU vs oracle)."""
    from cerberus_amd import api
    from cerberus_amd.synth import PriorData
    w0 = _fresh(cfg, ocfg, n_landmarks=50, seed=31)
    p = PriorData()
    ctx.marginalize(w0, 0, p)
    w_g = _fresh(cfg, ocfg, n_landmarks=50, seed=32)
    w_o = _fresh(cfg, ocfg, n_landmarks=50, seed=32)
    for w in (w_g, w_o):
        w.prior = p.copy()
    sg = ctx.solve_windows([w_g], api.default_solve_opts(True, 6))[0]
    so = O.solve_window(ocfg, w_o, O.default_opts(True, 6))
    np.testing.assert_allclose(sg.final_cost, so.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), np.abs(a - bb).max()   # measured: 1e-10 .. 1e-9


def _vins(cfg, ocfg, **kw):
    """USE_LEG = 0 (config/a1_config/hardware_a1_vins_config.yaml): IMUFactor instead of IMULegFactor, no leg-bias blocks."""
    w = _fresh(cfg, ocfg, with_prior=False, **kw)
    w.use_leg = 0
    return w


def test_solve_without_leg_factors(ctx, cfg, ocfg):
    from cerberus_amd import api
    w_g, w_o = _vins(cfg, ocfg, n_landmarks=40, seed=51), _vins(cfg, ocfg, n_landmarks=40, seed=51)
    lb0 = w_g.leg_bias.copy()
    sg = ctx.solve_windows([w_g], api.default_solve_opts(True, 5))[0]
    so = O.solve_window(ocfg, w_o, O.default_opts(True, 5))
    assert sg.iterations == so.iterations and sg.num_successful == so.num_successful
    np.testing.assert_allclose(sg.final_cost, so.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), np.abs(a - bb).max()   # measured: 1e-10 .. 1e-9
    np.testing.assert_array_equal(w_g.leg_bias, lb0)   # not part of the problem


def test_marginalize_and_next_solve_without_leg_factors(ctx, cfg, ocfg):
    from cerberus_amd import api
    from cerberus_amd.synth import PriorData
    w0 = _vins(cfg, ocfg, n_landmarks=50, seed=61)
    pg, po = PriorData(), PriorData()
    ctx.marginalize(w0, 0, pg)
    rc, m, A, bvec = O.marginalize(ocfg, w0, 0, po, want_A=True)
    assert rc == 0 and pg.struct.valid == 1 and pg.blocks() == po.blocks()
    n = pg.n
    assert n == po.n
    Jg, Jo = pg.J0_matrix(), po.J0_matrix()
    Ag, Ao = Jg.T @ Jg, Jo.T @ Jo
    assert np.abs(Ag - Ao).max() < 1e-6 * np.abs(Ao).max()
    assert np.abs(Jg.T @ pg.r0[:n] - Jo.T @ po.r0[:n]).max() < 1e-6 * np.abs(Jo.T @ po.r0[:n]).max()
    # no prior: A' is semi-definite (the gauge directions). The eigenvalues the reference keeps (> 1e-8) and the squared column norms
    # the one-sided Jacobi of k_marginalize_lds keeps agree in number up to the ones that are rounding noise around the threshold,
    # and the part of r0 that matters (|r0|^2 = b^T A'^+ b) agrees
    kept_g, kept_o = int((np.abs(Jg).max(axis=1) > 0).sum()), int((np.abs(Jo).max(axis=1) > 0).sum())
    sv = np.linalg.eigvalsh(Ao)
    noise = int(((sv > 1e-10) & (sv < 1e-6 * sv[-1] * 1e-6)).sum())
    assert abs(kept_g - kept_o) <= noise + 0, (kept_g, kept_o, noise, sv[:8])
    w_g, w_o = _vins(cfg, ocfg, n_landmarks=50, seed=62), _vins(cfg, ocfg, n_landmarks=50, seed=62)
    for w in (w_g, w_o):
        w.prior = pg.copy()
    sg = ctx.solve_windows([w_g], api.default_solve_opts(True, 5))[0]
    so = O.solve_window(ocfg, w_o, O.default_opts(True, 5))
    np.testing.assert_allclose(sg.final_cost, so.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), np.abs(a - bb).max()   # measured: 1e-10 .. 1e-9


@pytest.mark.parametrize("with_prior", [True, False])
def test_solve_window_without_landmarks(ctx, cfg, ocfg, with_prior):
    """Empty visual input (no feature has 4 observations yet): IMU-leg factors (+ prior) only."""
    from cerberus_amd import api

    def mk():
        w = _fresh(cfg, ocfg, n_landmarks=1, seed=77, with_prior=with_prior)
        w.L, w.n_obs = 0, 0
        w.lm_start_frame = np.zeros(0, np.int32); w.lm_obs_offset = np.zeros(1, np.int32)
        w.obs = np.zeros((0, 11)); w.obs_is_stereo = np.zeros(0, np.uint8); w.inv_depth = np.zeros(0)
        return w
    w_g, w_o = mk(), mk()
    sg = ctx.solve_windows([w_g], api.default_solve_opts(True, 4))[0]
    so = O.solve_window(ocfg, w_o, O.default_opts(True, 4))
    assert (sg.iterations, sg.num_successful) == (so.iterations, so.num_successful)
    np.testing.assert_allclose(sg.final_cost, so.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        if a.size:
            assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max())


def test_solve_window_at_the_feature_cap(ctx, cfg, ocfg):
    """NUM_OF_F = 1000 landmarks (parameters.h:24; BASELINE configs[2] size): 8003 observations, 21 chunks, 16 packed waves."""
    from cerberus_amd import api
    w_g = _fresh(cfg, ocfg, n_landmarks=1000, seed=91)
    w_o = _fresh(cfg, ocfg, n_landmarks=1000, seed=91)
    sg = ctx.solve_windows([w_g], api.default_solve_opts(True, 12))[0]
    so = O.solve_window(ocfg, w_o, O.default_opts(True, 12))
    assert (sg.iterations, sg.num_successful) == (so.iterations, so.num_successful) and sg.iterations == 12
    np.testing.assert_allclose(sg.final_cost, so.final_cost, rtol=1e-8)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        assert np.abs(a - bb).max() < 1e-8 * max(1.0, np.abs(bb).max()), np.abs(a - bb).max()   # measured: 1e-10 .. 1e-9


def test_gauge_fix(ctx, cfg, ocfg):
    from cerberus_amd import api
    w_g = _fresh(cfg, ocfg, n_landmarks=30, seed=41)
    w_o = _fresh(cfg, ocfg, n_landmarks=30, seed=41)
    before = w_g.clone_state()
    ctx.solve_windows([w_g], api.default_solve_opts(True, 3))
    w_o.set_state(w_g.clone_state())
    ctx.gauge_fix(before, w_g)
    O.gauge_fix(before, w_o)
    for a, bb in zip(w_g.state_arrays(), w_o.state_arrays()):
        np.testing.assert_allclose(a, bb, atol=1e-12)
    # yaw and position of frame 0 are restored
    np.testing.assert_allclose(w_g.pose[0, :3], before[0][0, :3], atol=1e-12)
