"""CPU tests of the oracle itself: analytic Jacobians vs central differences in the local (⊞)
parameterisation — automating the reference's print-only checkers
(IMULegFactor::checkJacobian imu_leg_factor.cpp:7-171, Projection*Factor::check
projectionTwoFrameOneCamFactor.cpp:152-272) — plus independent numpy re-derivations."""
import numpy as np
import pytest

from conftest import rand_pose
from oracle import oracle_py as O

RF = np.array([0.1805, -0.047, -0.0838, 0.21])


def test_kinematics_fd():
    rng = np.random.default_rng(0)
    for _ in range(10):
        q = np.array([0.2, 0.8, -1.6]) + 0.3 * rng.normal(size=3)
        lc = 0.21 + 0.01 * rng.normal()
        k = O.kin(q, lc, RF)
        eps = 1e-6
        Jn = np.zeros((3, 3)); dJn = np.zeros((9, 3))
        for c in range(3):
            dq = np.zeros(3); dq[c] = eps
            kp, km = O.kin(q + dq, lc, RF), O.kin(q - dq, lc, RF)
            Jn[:, c] = (kp["f"] - km["f"]) / (2 * eps)
            dJn[:, c] = ((kp["J"] - km["J"]) / (2 * eps)).T.reshape(9)  # vec column-major
        kp, km = O.kin(q, lc + eps, RF), O.kin(q, lc - eps, RF)
        np.testing.assert_allclose(k["J"], Jn, atol=1e-9)
        np.testing.assert_allclose(k["dJ_dq"], dJn, atol=1e-8)
        np.testing.assert_allclose(k["df_drho"], (kp["f"] - km["f"]) / (2 * eps), atol=1e-9)
        np.testing.assert_allclose(k["dJ_drho"], ((kp["J"] - km["J"]) / (2 * eps)).T.reshape(9), atol=1e-8)


def test_kinematics_closed_form():
    # independent numpy statement of the chain in A1Kinematics.cpp:58-66
    rng = np.random.default_rng(1)
    q = rng.normal(size=3)
    lc = 0.2
    ox, oy, d, lt = RF
    s, c = np.sin, np.cos
    p = np.array([ox - lt * s(q[1]) - lc * s(q[1] + q[2]),
                  oy + d * c(q[0]) + lt * c(q[1]) * s(q[0]) + lc * c(q[1]) * c(q[2]) * s(q[0]) - lc * s(q[0]) * s(q[1]) * s(q[2]),
                  d * s(q[0]) - lt * c(q[0]) * c(q[1]) - lc * c(q[0]) * c(q[1]) * c(q[2]) + lc * c(q[0]) * s(q[1]) * s(q[2])])
    np.testing.assert_allclose(O.kin(q, lc, RF)["f"], p, atol=1e-15)


def _local_plus(x, d):
    return O.pose_plus(x, d) if x.size == 7 else x + d


def _fd_jac(fn, params, nres, eps=1e-6):
    """Central differences of fn(params)->r wrt local increments of every block."""
    out = []
    for k, x in enumerate(params):
        ls = 6 if x.size == 7 else x.size
        J = np.zeros((nres, ls))
        for c in range(ls):
            d = np.zeros(ls); d[c] = eps
            pp = list(params); pm = list(params)
            pp[k] = _local_plus(x, d); pm[k] = _local_plus(x, -d)
            J[:, c] = (fn(pp) - fn(pm)) / (2 * eps)
        out.append(J)
    return out


def _proj_setup(rng, kind):
    pose_i = rand_pose(rng, 0.3); pose_j = rand_pose(rng, 0.3)
    pose_j[3:] = pose_i[3:] + 0.05 * rng.normal(size=4); pose_j[3:] /= np.linalg.norm(pose_j[3:])
    ex0 = np.array([0.1, 0.025, 0.11, 0.5, -0.5, 0.5, -0.5]); ex1 = ex0.copy(); ex1[1] = -0.025
    ex0[3:] += 0.01 * rng.normal(size=4); ex0[3:] /= np.linalg.norm(ex0[3:])
    ex1[3:] += 0.01 * rng.normal(size=4); ex1[3:] /= np.linalg.norm(ex1[3:])
    lam = np.array([1.0 / rng.uniform(2, 10)]); td = np.array([0.003])
    obs = np.concatenate([[rng.uniform(-.5, .5), rng.uniform(-.5, .5), 1.0], [rng.uniform(-.5, .5), rng.uniform(-.5, .5), 1.0],
                          rng.normal(size=4) * 0.2, [0.003, 0.001]])
    params = [[pose_i, pose_j, ex0, lam, td], [pose_i, pose_j, ex0, ex1, lam, td], [ex0, ex1, lam, td]][kind]
    return obs, params


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_projection_jacobians_fd(ocfg, kind):
    rng = np.random.default_rng(10 + kind)
    for _ in range(5):
        obs, params = _proj_setup(rng, kind)
        r, Js = O.eval_proj(kind, ocfg, obs, params)
        fd = _fd_jac(lambda p: O.eval_proj(kind, ocfg, obs, p, False)[0], params, 2)
        for k, (J, Jn) in enumerate(zip(Js, fd)):
            ls = Jn.shape[1]
            np.testing.assert_allclose(J[:, :ls], Jn, rtol=2e-6, atol=2e-5 * np.abs(Jn).max(), err_msg="block %d" % k)
            if J.shape[1] == 7:
                assert np.all(J[:, 6] == 0.0)


def test_projection_1f2c_lambda_quirk(ocfg):
    # projectionOneFrameTwoCamFactor.cpp:119 uses pts_i (not pts_i_td) in the inverse-depth Jacobian
    rng = np.random.default_rng(3)
    obs, params = _proj_setup(rng, 2)
    obs[10] = 0.02  # td_i far from td so pts_i_td != pts_i
    _, Js = O.eval_proj(2, ocfg, obs, params)
    fd = _fd_jac(lambda p: O.eval_proj(2, ocfg, obs, p, False)[0], params, 2)
    assert np.abs(Js[2][:, 0] - fd[2][:, 0]).max() > 1e-3 * np.abs(fd[2]).max()  # the quirk is visible
    obs2 = obs.copy(); obs2[6:8] = 0.0  # zero velocity -> pts_i_td == pts_i -> exact again
    _, Js2 = O.eval_proj(2, ocfg, obs2, params)
    fd2 = _fd_jac(lambda p: O.eval_proj(2, ocfg, obs2, p, False)[0], params, 2)
    np.testing.assert_allclose(Js2[2], fd2[2], rtol=1e-6)


def test_projection_numpy_rederivation(ocfg):
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(4)
    obs, params = _proj_setup(rng, 1)
    pose_i, pose_j, ex0, ex1, lam, td = params
    R = lambda p: Rot.from_quat(p[3:7]).as_matrix()
    pts_i = obs[0:3] - (td[0] - obs[10]) * np.array([obs[6], obs[7], 0])
    pts_j = obs[3:6] - (td[0] - obs[11]) * np.array([obs[8], obs[9], 0])
    Pw = R(pose_i) @ (R(ex0) @ (pts_i / lam[0]) + ex0[:3]) + pose_i[:3]
    pc = R(ex1).T @ (R(pose_j).T @ (Pw - pose_j[:3]) - ex1[:3])
    r_np = 460.0 / 1.5 * (pc[:2] / pc[2] - pts_j[:2])
    r, _ = O.eval_proj(1, ocfg, obs, params, False)
    np.testing.assert_allclose(r, r_np, rtol=1e-11, atol=1e-11)


def _imu_params(rng, w, k, leg=True, dbg_zero=False):
    pi, pj = w.pose[k].copy(), w.pose[k + 1].copy()
    sbi, sbj = w.speed_bias[k].copy(), w.speed_bias[k + 1].copy()
    lbi, lbj = w.leg_bias[k].copy(), w.leg_bias[k + 1].copy()
    if dbg_zero:
        sbi[6:9] = w.lin[k][3:6]
    return [pi, sbi, lbi, pj, sbj, lbj] if leg else [pi, sbi, pj, sbj]


@pytest.mark.parametrize("leg", [True, False])
def test_imu_factor_jacobians_fd(ocfg, small_window, leg):
    w = small_window
    rng = np.random.default_rng(5)
    for k in (0, 4, 9):
        for dbg_zero, tol in ((True, 2e-5), (False, 5e-3)):
            params = _imu_params(rng, w, k, leg, dbg_zero)
            pre = w.preint[k] if leg else w.preint_imu[k]
            ev = (lambda p, j=True: O.eval_imu_leg(ocfg, pre, p, j)) if leg else (lambda p, j=True: O.eval_imu(ocfg, pre, p, j))
            nres = 31 if leg else 15
            r, Js = ev(params)
            # un-whiten for a meaningful comparison scale: compare U^-1 J
            cov = (pre[33 + 961:] if leg else pre[17 + 225:]).reshape(nres, nres)
            U = O.sqrt_info(cov)
            Ui = np.linalg.inv(U)
            fd = _fd_jac(lambda p: ev(p, False)[0], params, nres, eps=1e-7)
            for b, (J, Jn) in enumerate(zip(Js, fd)):
                ls = Jn.shape[1]
                A, B = Ui @ J[:, :ls], Ui @ Jn
                np.testing.assert_allclose(A, B, atol=tol * max(1.0, np.abs(B).max()), err_msg="k=%d block %d" % (k, b))


def test_sqrt_info_properties(small_window):
    w = small_window
    for k in range(10):
        cov = w.preint[k][33 + 961:].reshape(31, 31)
        U = O.sqrt_info(cov, 0)
        assert np.allclose(U, np.triu(U))
        M = U @ cov @ U.T  # = I when U^T U = cov^-1
        np.testing.assert_allclose(M, np.eye(31), atol=1e-6)
        # the reference-literal route (inverse with partial pivoting + LLT) is the same matrix, row by row (see the next test)
        U1 = O.sqrt_info(cov, 1)
        rows = np.linalg.norm(U1 - U, axis=1) / np.linalg.norm(U, axis=1)
        assert rows.max() < 1e-12, rows.max()


def test_sqrt_info_routes_against_100_digit_arithmetic(cfg, small_window):
    """LLT(cov^-1).matrixL()^T (imu_leg_factor.cpp:197-198) evaluated in 100-digit arithmetic from the FP64 covariance: the default route
    (Cholesky of the index-reversed covariance + triangular inverse) and the reference's route taken literally (inverse, then LLT) both give
    it to a few 1e-15 — row by row for the factor, per diagonal for the information it stands for. The covariance's raw condition number
    (1e13 .. 1e14 for a trot, 1e20 with every foot in the air at some sample) is units — variances of 1e-11 beside ones of 0.1 or 10e10 —:
    after diagonal equilibration it is ~ 15, which is what these factorisations see. One integration step alone leaves a rank-deficient
    covariance (no sqrt_info exists)."""
    import mpmath as mp
    from cerberus_amd import synth
    w = small_window
    recs = [(w.preint[k][33 + 961:].reshape(31, 31), "trot") for k in (0, 5, 9)]
    # an interval with three samples of every foot in the air (imu_leg_integration_base.cpp:354-358: uncertainties 10e10)
    a0, a1 = int(w.sample_offsets[2]), int(w.sample_offsets[3])
    smp = np.array(w.samples[a0:a1], copy=True)
    smp[5:8, 31:35] = 0.0
    recs.append((O.preintegrate_imu_leg(cfg, smp, w.lin[2])[33 + 961:].reshape(31, 31), "feet in the air"))
    for cov, what in recs:
        d = np.sqrt(np.diag(cov))
        eq = np.linalg.eigvalsh(cov / np.outer(d, d))
        raw = np.linalg.eigvalsh(0.5 * (cov + cov.T))
        assert eq[-1] / eq[0] < 50 and raw[-1] / raw[0] > 1e12, (what, eq[-1] / eq[0], raw[-1] / raw[0])
        with mp.workdps(100):
            Cm = mp.matrix(cov.tolist())
            inv = mp.inverse((Cm + Cm.T) / 2)
            Ue = np.array(mp.cholesky(inv).T.tolist(), dtype=np.float64)
            Le = np.array(inv.tolist(), dtype=np.float64)
        dl = np.sqrt(np.diag(Le))
        for mode in (0, 1):
            U = O.sqrt_info(cov, mode)
            rows = np.linalg.norm(U - Ue, axis=1) / np.linalg.norm(Ue, axis=1)
            info = np.abs(U.T @ U - Le) / np.outer(dl, dl)
            assert rows.max() < 2e-14 and info.max() < 5e-14, (what, mode, rows.max(), info.max())   # measured: 3e-15, 6e-15
    one_step = O.preintegrate_imu_leg(cfg, w.samples[a0:a0 + 2], w.lin[2])[33 + 961:].reshape(31, 31)
    d = np.sqrt(np.diag(one_step))
    eq = np.linalg.eigvalsh(one_step / np.outer(d, d))
    assert eq[0] < 1e-12 * eq[-1]


def test_preintegration_imu_subblock_consistency(cfg, small_window):
    """The 15-dim sub-block of the IMU-leg preintegration equals classic IMU preintegration when
    ACC_N_Z == ACC_N (SURVEY parity note 3)."""
    import ctypes as C
    w = small_window
    c2 = O.config_from(cfg)
    c2.acc_n_z = c2.acc_n
    idx = np.r_[0:9, 21:27]
    for k in (0, 5):
        a, b = w.sample_offsets[k], w.sample_offsets[k + 1]
        pl = O.preintegrate_imu_leg(c2, w.samples[a:b], w.lin[k])
        pi = O.preintegrate_imu(c2, w.samples[a:b], w.lin[k][:6])
        np.testing.assert_allclose(pl[0:11], pi[0:11], rtol=0, atol=1e-15)
        Jl = pl[33:33 + 961].reshape(31, 31)[np.ix_(idx, idx)]
        Ji = pi[17:17 + 225].reshape(15, 15)
        np.testing.assert_allclose(Jl, Ji, atol=1e-14)
        Cl = pl[33 + 961:].reshape(31, 31)[np.ix_(idx, idx)]
        Ci = pi[17 + 225:].reshape(15, 15)
        np.testing.assert_allclose(Cl, Ci, rtol=1e-10, atol=1e-22)


def test_preintegration_bias_jacobians(ocfg, small_window):
    """jacobian sub-blocks (A.4) predict the effect of re-propagating with perturbed ba/bg/rho."""
    w = small_window
    k = 3
    a, b = w.sample_offsets[k], w.sample_offsets[k + 1]
    base = O.preintegrate_imu_leg(ocfg, w.samples[a:b], w.lin[k])
    J = base[33:33 + 961].reshape(31, 31)
    d = np.zeros(10); d[0:3] = [1e-4, -2e-4, 1.5e-4]; d[3:6] = [2e-5, 1e-5, -3e-5]; d[6:10] = [1e-4, -1e-4, 2e-4, 5e-5]
    pert = O.preintegrate_imu_leg(ocfg, w.samples[a:b], w.lin[k] + d)
    dp = pert[1:4] - base[1:4]; dv = pert[8:11] - base[8:11]
    np.testing.assert_allclose(dp, J[0:3, 21:24] @ d[0:3] + J[0:3, 24:27] @ d[3:6], rtol=2e-3, atol=1e-10)
    np.testing.assert_allclose(dv, J[6:9, 21:24] @ d[0:3] + J[6:9, 24:27] @ d[3:6], rtol=2e-3, atol=1e-10)
    for j in range(4):
        de = pert[11 + 3 * j:14 + 3 * j] - base[11 + 3 * j:14 + 3 * j]
        pred = J[9 + 3 * j:12 + 3 * j, 24:27] @ d[3:6] + J[9 + 3 * j:12 + 3 * j, 27 + j] * d[6 + j]
        np.testing.assert_allclose(de, pred, rtol=5e-3, atol=1e-9)
    cov = base[33 + 961:].reshape(31, 31)
    np.testing.assert_allclose(cov, cov.T, rtol=1e-9, atol=1e-25)
    assert np.linalg.eigvalsh(cov).min() > 0


def test_step_F_matches_fd(ocfg, small_window):
    """F (A.3) is the Jacobian of one midpoint step w.r.t. the error state: check the (P,V,eps)x(BA,BG,rho) columns
    through re-propagation of a single step."""
    w = small_window
    s0, s1 = w.samples[1], w.samples[2]
    lin = w.lin[0]
    F, V = O.step_FV(ocfg, s0, s1, [0, 0, 0, 1], lin)
    two = np.stack([s0, s1])
    base = O.preintegrate_imu_leg(ocfg, two, lin)
    eps = 1e-6
    for col, off in ((21, 0), (24, 3)):
        for c in range(3):
            d = np.zeros(10); d[off + c] = eps
            pp = O.preintegrate_imu_leg(ocfg, two, lin + d); pm = O.preintegrate_imu_leg(ocfg, two, lin - d)
            dnum = (pp[1:23] - pm[1:23]) / (2 * eps)  # dp(3) dq(4) dv(3) eps(12)
            np.testing.assert_allclose(dnum[0:3], F[0:3, col + c], atol=1e-9)
            np.testing.assert_allclose(dnum[7:10], F[6:9, col + c], atol=1e-8)
            np.testing.assert_allclose(dnum[10:22], F[9:21, col + c], atol=2e-7)
    assert np.count_nonzero(F) <= 241 + 4 and np.count_nonzero(V) > 0


def test_huber():
    np.testing.assert_allclose(O.huber(1.0, 0.25), [0.25, 1.0, 0.0])
    r = O.huber(1.0, 4.0)
    np.testing.assert_allclose(r, [3.0, 0.5, -0.5 / 8.0])


def test_pose_plus():
    x = np.array([1, 2, 3, 0.1, -0.2, 0.3, 0.9]); x[3:] /= np.linalg.norm(x[3:])
    d = np.array([0.1, 0.2, -0.1, 0.01, -0.02, 0.03])
    out = O.pose_plus(x, d)
    np.testing.assert_allclose(out[:3], x[:3] + d[:3])
    assert abs(np.linalg.norm(out[3:]) - 1) < 1e-15
    from scipy.spatial.transform import Rotation as Rot
    dq = np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0]); dq /= np.linalg.norm(dq)
    ref = (Rot.from_quat(x[3:]) * Rot.from_quat(dq)).as_quat()
    if ref[3] * out[6] < 0:
        ref = -ref
    np.testing.assert_allclose(out[3:], ref, atol=1e-15)


def test_prior_eval(ocfg, small_window):
    w = small_window
    pr = w.prior.struct
    blocks = w.prior.blocks()
    rng = np.random.default_rng(8)
    params = []
    off = 0
    for (bid, size, idx) in blocks:
        x0 = w.prior.x0[off:off + size]
        off += size
        params.append(O.pose_plus(x0, 0.01 * rng.normal(size=6)) if size == 7 else x0 + 0.01 * rng.normal(size=size))
    r, Js = O.eval_prior(pr, params)
    fd = _fd_jac(lambda p: O.eval_prior(pr, p, False)[0], params, pr.n)
    J0 = w.prior.J0_matrix()
    for (bid, size, idx), J, Jn in zip(blocks, Js, fd):
        ls = 6 if size == 7 else size
        np.testing.assert_array_equal(J[:, :ls], J0[:, idx:idx + ls])
        np.testing.assert_allclose(J[:, :ls], Jn, atol=2e-2 * np.abs(J0).max())  # dtheta=2vec(dq) is first order


def test_non_finite_start_fails_like_ceres(cfg, ocfg):
    """IterationZero with a non-finite evaluation: FAILURE, parameters untouched (the device side: tests/test_error_paths.py)."""
    from cerberus_amd import synth
    w = synth.make_window(cfg, n_landmarks=20, seed=3)
    O.fill_preint(ocfg, w)
    w.state_arrays()[0][4, 1] = np.nan
    before = w.clone_state()
    with pytest.raises(FloatingPointError):
        O.solve_window(ocfg, w, O.default_opts(True, 3))
    for a, b in zip(w.state_arrays(), before):
        np.testing.assert_array_equal(a, b)
