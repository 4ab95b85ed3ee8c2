"""Golden vectors: outputs of the REFERENCE's own factor code (compiled from /root/reference into oracle/_ref/libref.so
and frozen by tests/golden/make_golden.py). CPU tests pin the oracle to them; `-m gpu` tests compare the HIP path,
called through the C-ABI, with the reference outputs directly. Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

from cerberus_amd import synth
from oracle import oracle_py as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))
RF = np.array([0.1805, -0.047, -0.0838, 0.21])
PROJ_SIZES = [[7, 7, 7, 1, 1], [7, 7, 7, 7, 1, 1], [7, 7, 1, 1]]


def _split(v, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(v[..., o:o + s]); o += s
    return out


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def _per_entry(a, b):
    """largest |a_i - b_i| / |b_i| (entries below 1e-12 of the largest one are held to that floor)"""
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-12 * np.abs(b).max())).max())


def _per_row(A, B):
    return float((np.linalg.norm(A - B, axis=1) / np.linalg.norm(B, axis=1)).max())


@pytest.fixture(scope="module")
def gwin():
    return synth.make_window(synth.default_config(), n_landmarks=int(G["win_landmarks"]), seed=int(G["win_seed"]))


def _info_blocks(prior):
    """(ids, H, b, x0) of a prior with blocks sorted by id — same construction as make_golden.py."""
    from oracle import ref_py as R
    Hb, bb, x0 = R.prior_information(prior)
    ids = sorted(x0)
    return (np.array(ids), np.block([[Hb[(a, c)] for c in ids] for a in ids]), np.concatenate([bb[i] for i in ids]),
            np.concatenate([x0[i] for i in ids]))


# ------------------------------------------------------------------------------------------ oracle vs reference (CPU)
def test_oracle_kinematics():
    for i in range(G["kin_q"].shape[0]):
        k = O.kin(G["kin_q"][i], float(G["kin_lc"][i]), RF)
        for name in ("f", "J", "df_drho", "dJ_dq", "dJ_drho"):
            np.testing.assert_allclose(k[name], G["kin_" + name][i], rtol=0, atol=1e-14)


def test_oracle_preintegration(gwin):
    w, cfg = gwin, O.default_config()
    O.fill_preint(cfg, w)
    np.testing.assert_allclose(w.preint[:, :33], G["preint"][:, :33], rtol=1e-12, atol=1e-14)          # state + linearisation point
    np.testing.assert_allclose(w.preint[:, 33:994], G["preint"][:, 33:994], rtol=1e-10, atol=1e-12)    # jacobian
    for k in range(w.F - 1):
        cov = G["preint"][k, 994:]
        np.testing.assert_allclose(w.preint[k, 994:], cov, rtol=1e-9, atol=1e-12 * np.abs(cov).max())
    np.testing.assert_allclose(w.preint_imu, G["preint_imu"], rtol=1e-10, atol=1e-16)


def test_oracle_imu_factors():
    cfg = O.default_config()
    for k in range(G["imu_params"].shape[0]):
        P = _split(G["imu_params"][k], [7, 9, 4, 7, 9, 4])
        r, J = O.eval_imu_leg(cfg, G["preint"][k], P)
        Jr = G["imuleg_J"][k]
        Jo = np.hstack(J)
        # whitened quantities against the compiled reference's: rounding level (measured 7e-14 per entry, 2e-15 per row; why the covariance's
        # condition number does not enter: tests/test_oracle_factors.py::test_sqrt_info_routes_against_100_digit_arithmetic)
        assert _per_entry(r, G["imuleg_r"][k]) < 1e-11
        assert _per_row(Jo, Jr) < 1e-13
        ri, Ji = O.eval_imu(cfg, G["preint_imu"][k], [P[0], P[1], P[3], P[4]])
        assert _per_entry(ri, G["imu_r"][k]) < 1e-11
        assert _per_row(np.hstack(Ji), G["imu_J"][k]) < 1e-13


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_oracle_projection(kind):
    cfg = O.default_config()
    for i in range(G["proj%d_obs" % kind].shape[0]):
        P = _split(G["proj%d_params" % kind][i], PROJ_SIZES[kind])
        r, J = O.eval_proj(kind, cfg, G["proj%d_obs" % kind][i], P)
        np.testing.assert_allclose(r, G["proj%d_r" % kind][i], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(np.hstack(J), G["proj%d_J" % kind][i], rtol=1e-11, atol=1e-9)


def test_oracle_pose_plus_and_prior(gwin):
    for i in range(G["plus_x"].shape[0]):
        np.testing.assert_allclose(O.pose_plus(G["plus_x"][i], G["plus_d"][i]), G["plus_out"][i], rtol=0, atol=1e-15)
    pr = gwin.prior
    sizes = [pr.struct.block_size[k] for k in range(pr.struct.n_blocks)]
    r, J = O.eval_prior(pr.struct, _split(G["prior_params"], sizes))
    np.testing.assert_allclose(r, G["prior_r"], rtol=1e-12, atol=1e-12 * np.abs(G["prior_r"]).max())
    np.testing.assert_allclose(np.hstack(J), G["prior_J"], rtol=0, atol=1e-13 * np.abs(G["prior_J"]).max())


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_marginalization(gwin, mode):
    w, cfg = gwin, O.default_config()
    w.preint[...] = G["preint"]
    p = synth.PriorData()
    assert O.marginalize(cfg, w, mode, p)[0] == 0
    ids, H, b, x0 = _info_blocks(p)
    tol = 1e-5 if mode == 0 else 1e-11   # MARGIN_OLD: eps * cond(Amm), see test_oracle_vs_reference.test_marginalization; MARGIN_SECOND_NEW: rounding (measured 1e-14)
    np.testing.assert_array_equal(ids, G["marg%d_ids" % mode])
    np.testing.assert_allclose(H, G["marg%d_H" % mode], rtol=0, atol=tol * np.abs(G["marg%d_H" % mode]).max())
    np.testing.assert_allclose(b, G["marg%d_b" % mode], rtol=0, atol=tol * np.abs(G["marg%d_b" % mode]).max())
    np.testing.assert_array_equal(x0, G["marg%d_x0" % mode])


# ------------------------------------------------------------------------------------------ HIP path vs reference (GPU)
@pytest.fixture(scope="module")
def ctx(cfg):
    from cerberus_amd import api
    c = api.Context(cfg, 0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_preintegration_vs_reference(ctx, gwin):
    w = gwin
    ctx.preintegrate_window(w)
    np.testing.assert_allclose(w.preint[:, :33], G["preint"][:, :33], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(w.preint[:, 33:994], G["preint"][:, 33:994], rtol=1e-9, atol=1e-11)
    for k in range(w.F - 1):
        cov = G["preint"][k, 994:]
        np.testing.assert_allclose(w.preint[k, 994:], cov, rtol=1e-8, atol=1e-11 * np.abs(cov).max())
    np.testing.assert_allclose(w.preint_imu, G["preint_imu"], rtol=1e-9, atol=1e-15)


@pytest.mark.gpu
def test_gpu_imu_factors_vs_reference(ctx):
    P = _split(G["imu_params"], [7, 9, 4, 7, 9, 4])
    r, Js = ctx.eval_imu_leg(G["preint"], P)
    for k in range(r.shape[0]):
        Jg, Jr = np.hstack([J[k] for J in Js]), G["imuleg_J"][k]
        # the HIP path's whitened residual and Jacobians against the compiled reference's: measured 8e-14 per entry, 2e-15 per row
        assert _per_entry(r[k], G["imuleg_r"][k]) < 1e-11
        assert _per_row(Jg, Jr) < 1e-13
    r, Js = ctx.eval_imu(G["preint_imu"], [P[0], P[1], P[3], P[4]])
    for k in range(r.shape[0]):
        assert _per_entry(r[k], G["imu_r"][k]) < 1e-11
        assert _per_row(np.hstack([J[k] for J in Js]), G["imu_J"][k]) < 1e-13


@pytest.mark.gpu
def test_gpu_literal_sqrt_info_route_vs_reference(ctx):
    """vilo_set_sqrt_info_mode(1): inverse() + LLT as imu_leg_factor.cpp:197-198 writes it, on the device (Gauss-Jordan with partial pivoting,
    then the lower Cholesky factor, transposed). Both device routes against the outputs of the compiled reference (whose inverse / LLT are
    the test shim's): rounding level — 8e-15 per entry for the literal route, 8e-14 for the default one — and the two routes are different
    code giving the same matrix: they differ from each other, by rounding. (The covariance's condition number of 1e13 .. 1e14 is units; see
    tests/test_oracle_factors.py::test_sqrt_info_routes_against_100_digit_arithmetic.)"""
    from cerberus_amd import api
    P = _split(G["imu_params"], [7, 9, 4, 7, 9, 4])
    r0, J0 = ctx.eval_imu_leg(G["preint"], P)
    assert api.lib().vilo_set_sqrt_info_mode(ctx.h, 1) == 0
    try:
        r1, J1 = ctx.eval_imu_leg(G["preint"], P)
    finally:
        assert api.lib().vilo_set_sqrt_info_mode(ctx.h, 0) == 0
    assert api.lib().vilo_set_sqrt_info_mode(ctx.h, 7) != 0
    worst = 0.0
    for k in range(r0.shape[0]):
        Jl, Jd, Jr = np.hstack([J[k] for J in J1]), np.hstack([J[k] for J in J0]), G["imuleg_J"][k]
        assert _per_entry(r1[k], G["imuleg_r"][k]) < 1e-11
        assert _per_row(Jl, Jr) < 1e-13
        worst = max(worst, _per_row(Jl, Jd), _per_entry(r1[k], r0[k]))
    assert 0.0 < worst < 1e-11, worst   # two different routes (not the same code twice), the same matrix to rounding
    # the oracle's literal route (orc_sqrt_info mode 1) on the same covariances
    for k in range(r0.shape[0]):
        cov = G["preint"][k, 994:].reshape(31, 31)
        U0, U1 = O.sqrt_info(cov, 0), O.sqrt_info(cov, 1)
        assert _per_row(U1, U0) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_gpu_projection_vs_reference(ctx, kind):
    P = _split(G["proj%d_params" % kind], PROJ_SIZES[kind])
    r, Js = ctx.eval_proj(kind, G["proj%d_obs" % kind], P)
    np.testing.assert_allclose(r, G["proj%d_r" % kind], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(np.concatenate(Js, axis=2), G["proj%d_J" % kind], rtol=1e-11, atol=1e-9)


@pytest.mark.gpu
def test_gpu_pose_plus_and_prior_vs_reference(ctx, gwin):
    np.testing.assert_allclose(ctx.pose_plus(G["plus_x"], G["plus_d"]), G["plus_out"], rtol=0, atol=1e-15)
    r, J = ctx.eval_prior(gwin.prior, G["prior_params"][None, :])
    np.testing.assert_allclose(r[0], G["prior_r"], rtol=1e-12, atol=1e-12 * np.abs(G["prior_r"]).max())
    np.testing.assert_allclose(J[0], G["prior_J"], rtol=0, atol=1e-13 * np.abs(G["prior_J"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_marginalization_vs_reference(ctx, gwin, mode):
    w = gwin
    w.preint[...] = G["preint"]
    p = synth.PriorData()
    ctx.marginalize(w, mode, p)
    ids, H, b, x0 = _info_blocks(p)
    tol = 1e-5 if mode == 0 else 1e-11   # (MARGIN_SECOND_NEW drops one well-conditioned pose block: rounding level, measured 1e-14)
    np.testing.assert_array_equal(ids, G["marg%d_ids" % mode])
    np.testing.assert_allclose(H, G["marg%d_H" % mode], rtol=0, atol=tol * np.abs(G["marg%d_H" % mode]).max())
    np.testing.assert_allclose(b, G["marg%d_b" % mode], rtol=0, atol=tol * np.abs(G["marg%d_b" % mode]).max())
    np.testing.assert_array_equal(x0, G["marg%d_x0" % mode])


# ------------------------------------------------------------------------------ force-based contact model (contact_sensor_type 2)
GF = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preint_force_model.npz"))


def _force_inputs(gwin):
    from test_oracle_vs_reference import force_samples
    return force_samples(gwin.samples, seed=int(GF["force_seed"]))


def _check_force_records(pre, rt_state, rt_jac, rt_cov):
    ref = GF["preint"]
    np.testing.assert_allclose(pre[:, :33], ref[:, :33], rtol=rt_state, atol=1e-13)
    np.testing.assert_allclose(pre[:, 33:994], ref[:, 33:994], rtol=rt_jac, atol=1e-11)
    for k in range(ref.shape[0]):
        cov = ref[k, 994:]
        np.testing.assert_allclose(pre[k, 994:], cov, rtol=rt_cov, atol=1e-11 * np.abs(cov).max())


def test_oracle_force_model_preintegration(gwin):
    """tests/golden/preint_force_model.npz: the reference's IMULegIntegrationBase with contact_sensor_type 2
    (imu_leg_integration_base.cpp:195-229, 300-317), frozen by tests/golden/make_golden_rank3.py."""
    import copy
    cfg2 = copy.copy(O.default_config())
    cfg2.contact_sensor_type = 2
    w, smp = gwin, _force_inputs(gwin)
    pre = np.array([O.preintegrate_imu_leg(cfg2, smp[w.sample_offsets[k]:w.sample_offsets[k + 1]], w.lin[k]) for k in range(w.F - 1)])
    _check_force_records(pre, 1e-12, 1e-10, 1e-9)


@pytest.mark.gpu
def test_gpu_force_model_preintegration_vs_reference(gwin):
    import copy
    from cerberus_amd import api
    c2 = copy.copy(synth.default_config())
    c2.contact_sensor_type = 2
    c = api.Context(c2, 0)
    w, smp = gwin, _force_inputs(gwin)
    _check_force_records(c.preintegrate(smp, w.sample_offsets, w.lin), 1e-11, 1e-9, 1e-8)
    c.close()


# ------------------------------------------------------------------------------ repropagate() with the force-based contact model
GR = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preint_force_model_reprop.npz"))


def _check_records(pre, ref, rt_state, rt_jac, rt_cov):
    np.testing.assert_allclose(pre[:, :33], ref[:, :33], rtol=rt_state, atol=1e-13)
    np.testing.assert_allclose(pre[:, 33:994], ref[:, 33:994], rtol=rt_jac, atol=1e-11)
    for k in range(ref.shape[0]):
        cov = ref[k, 994:]
        np.testing.assert_allclose(pre[k, 994:], cov, rtol=rt_cov, atol=1e-11 * np.abs(cov).max())


def test_oracle_repropagation_keeps_the_force_filter(gwin):
    """tests/golden/preint_force_model_reprop.npz: the reference's own repropagate() (imu_leg_integration_base.cpp:62-86) called twice on
    an object that integrated its samples with contact_sensor_type 2. It does not reset foot_force_min / max / window (:29-41 are
    constructor-only), so each pass starts from the filter the previous one left — the frozen passes differ from one another even at
    equal biases, and from a fresh integration by tens of percent in the covariance."""
    import copy
    cfg2 = copy.copy(O.default_config())
    cfg2.contact_sensor_type = 2
    w, smp = gwin, _force_inputs(gwin)
    for name, second in (("same", GR["l1"]), ("vary", GR["l2"])):
        for k in range(w.F - 1):
            s = smp[w.sample_offsets[k]:w.sample_offsets[k + 1]]
            got = O.repropagate_imu_leg(cfg2, s, w.lin[k], [GR["l1"][k], second[k]])
            _check_records(got, GR[name][k], 1e-12, 1e-10, 1e-9)
    # the effect is not small: pass 3 at the biases of pass 2 is another record, and a fresh integration at l1 is neither
    k = 0
    cov2, cov3 = GR["same"][k, 1, 994:], GR["same"][k, 2, 994:]
    fresh = O.preintegrate_imu_leg(cfg2, smp[w.sample_offsets[k]:w.sample_offsets[k + 1]], GR["l1"][k])[994:]
    assert np.abs(cov3 - cov2).max() > 1e-6 * np.abs(cov2).max()
    assert np.abs(fresh - cov2).max() > 1e-2 * np.abs(cov2).max()
    np.testing.assert_array_equal(GR["same"][:, 0], GF["preint"])   # (pass 1 is the record of preint_force_model.npz)


@pytest.mark.gpu
def test_gpu_force_model_repropagation_vs_reference(gwin):
    """k_repropagate on a resident batch (vilo_batch_set_samples) against the reference's repropagate(): the batch's intervals are objects
    that integrated their samples once; a solve of zero iterations evaluates the initial point = one repropagate() at its biases (pass 2),
    the next one another (pass 3) — same biases, and still a different record, because the force filter carries on."""
    import copy
    from cerberus_amd import api
    c2 = copy.copy(synth.default_config())
    c2.contact_sensor_type = 2
    c = api.Context(c2, 0)
    w = synth.make_window(synth.default_config(), n_landmarks=24, seed=5)   # (gwin's twin: this one is modified)
    np.testing.assert_array_equal(w.lin, gwin.lin)
    smp = _force_inputs(gwin)
    w.samples[...] = smp
    w.preint[...] = c.preintegrate(smp, w.sample_offsets, w.lin)
    _check_records(w.preint, GR["same"][:, 0], 1e-11, 1e-9, 1e-8)
    w.speed_bias[:10, 3:9] = GR["l1"][:, :6]
    w.leg_bias[:10, :] = GR["l1"][:, 6:]
    b = api.Batch(c, [w])
    try:
        b.set_samples()
        opts = api.default_solve_opts(True, 0)
        for p in (1, 2):
            b.solve(opts)
            rec = b.fetch(13).reshape(10, -1)
            np.testing.assert_array_equal(rec[:, 23:33], GR["l1"])   # lin_ba, lin_bg, lin_rho of the re-integration
            _check_records(rec, GR["same"][:, p], 1e-11, 1e-9, 1e-8)
        # switching the samples off and on again makes new objects: the first re-integration is pass 2 again
        b.set_samples(False)
        b.set_samples()
        b.solve(opts)
        _check_records(b.fetch(13).reshape(10, -1), GR["same"][:, 1], 1e-11, 1e-9, 1e-8)
    finally:
        b.close()
        c.close()


# ------------------------------------------------------------------------------ contact inputs no gait produces (reference branches)
GE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preint_contact_edges.npz"))


def _edge_inputs(gwin):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_edges as E
    return (E.contact_edges(gwin.samples, gwin.sample_offsets, seed=int(GE["flag_seed"])),
            E.force_edges(gwin.samples, gwin.sample_offsets, seed=int(GE["force_seed"])))


def test_golden_contact_edges_reach_the_branches():
    """What the fixture is for: intervals 1, 8 (whole interval) and 0, 4, 7 (part of it) integrate steps with all four flags 0 —
    imu_leg_integration_base.cpp:354-358 sets the leg-velocity variances to 10e10, so the foot-position covariance is ~1e11 dt^2 per
    step — and intervals 2, 3, 6 decide contact on c >= 0.5 exactly (:183-194): with `>` instead of `>=` interval 6 would be in the air."""
    cov = np.abs(GE["flags"][:, 994:]).max(axis=1)
    assert (cov[[0, 1, 4, 7, 8]] > 1e6).all() and (cov[[2, 3, 6]] < 1.0).all(), cov
    covf = np.abs(GE["force"][:, 994:]).max(axis=1)
    assert (covf[1::2] > 1e6).all() and (covf[0::2] < 1.0).all(), covf


def test_oracle_contact_edges(gwin):
    import copy
    cfg = O.default_config()
    cfg2 = copy.copy(cfg)
    cfg2.contact_sensor_type = 2
    w = gwin
    ce, fe = _edge_inputs(gwin)
    pre = np.array([O.preintegrate_imu_leg(cfg, ce[w.sample_offsets[k]:w.sample_offsets[k + 1]], w.lin[k]) for k in range(w.F - 1)])
    _check_records(pre, GE["flags"], 1e-12, 1e-10, 1e-9)
    pre = np.array([O.preintegrate_imu_leg(cfg2, fe[w.sample_offsets[k]:w.sample_offsets[k + 1]], w.lin[k]) for k in range(w.F - 1)])
    _check_records(pre, GE["force"], 1e-12, 1e-10, 1e-9)


@pytest.mark.gpu
def test_gpu_contact_edges_vs_reference(ctx, gwin):
    """k_preint_imu_leg on the same inputs: all feet in the air (kernels_preint.hip, `10e10` branch), the 0.5 threshold with non-binary c,
    and both again through the force model — and the factor built on such a record (sqrt_info of a covariance with 1e7 entries beside 1e-8
    ones) against the oracle's."""
    import copy
    from cerberus_amd import api
    w = gwin
    ce, fe = _edge_inputs(gwin)
    pre = ctx.preintegrate(ce, w.sample_offsets, w.lin)
    _check_records(pre, GE["flags"], 1e-11, 1e-9, 1e-8)
    c2 = copy.copy(synth.default_config())
    c2.contact_sensor_type = 2
    c = api.Context(c2, 0)
    try:
        _check_records(c.preintegrate(fe, w.sample_offsets, w.lin), GE["force"], 1e-11, 1e-9, 1e-8)
    finally:
        c.close()
    # IMULegFactor on the in-the-air records: whitened residual / Jacobian through the C-ABI against the oracle on the reference's record
    P = _split(G["imu_params"], [7, 9, 4, 7, 9, 4])
    r, Js = ctx.eval_imu_leg(GE["flags"], P)
    cfg = O.default_config()
    for k in range(r.shape[0]):
        ro, Jo = O.eval_imu_leg(cfg, GE["flags"][k], [p[k] for p in P])
        Jo, Jg = np.hstack(Jo), np.hstack([J[k] for J in Js])
        assert _per_entry(r[k], ro) < 1e-11, (k, _per_entry(r[k], ro))      # (raw condition numbers of 1e20 here: units, like everywhere else)
        assert _per_row(Jg, Jo) < 1e-13, (k, _per_row(Jg, Jo))


# ------------------------------------------------------------------------------ gauge fix at the Euler singularity (estimator.cpp:925-934)
GG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gauge_fix_edges.npz"))


def _gauge_cases(fix, w):
    """fix(before_arrays, w): the implementation under test; compares every case of the fixture with what ref_gauge_fix gave."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_gauge as MG
    worst = 0.0
    for n in range(GG["cases"].shape[0]):
        g = {k: GG["%s_%d" % (k, n)] for k in ("before_pose", "before_sb", "after_pose", "after_sb", "after_ex", "fixed_pose", "fixed_sb", "fixed_ex")}
        fp, fs, fe = MG.run_fix(fix, w, g["before_pose"], g["before_sb"], g["after_pose"], g["after_sb"], g["after_ex"])
        for a, b in ((fp, g["fixed_pose"]), (fs, g["fixed_sb"]), (fe, g["fixed_ex"])):
            worst = max(worst, np.abs(a - b).max())
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12, err_msg="case %d: pitch before / after %s" % (n, GG["cases"][n]))
        # what the fix is for: frame 0 keeps its position, and its yaw (regular branch) or its whole rotation (singular branch)
        np.testing.assert_allclose(fp[0, :3], g["before_pose"][0, :3], atol=1e-12)
        if GG["branch_taken"][n]:
            assert min(np.abs(fp[0, 3:] - g["before_pose"][0, 3:]).max(), np.abs(fp[0, 3:] + g["before_pose"][0, 3:]).max()) < 1e-8
    return worst


def test_golden_gauge_cases_cover_both_branches():
    t = GG["branch_taken"]
    assert t.sum() >= 8 and (~t).sum() >= 3
    c = GG["cases"]
    assert any(abs(a) < 89 and abs(b) > 89 for a, b in c[t]) and any(abs(a) > 89 and abs(b) < 89 for a, b in c[t])   # either clause alone


def test_oracle_gauge_fix_at_the_euler_singularity(gwin):
    _gauge_cases(O.gauge_fix, gwin)


@pytest.mark.gpu
def test_gpu_gauge_fix_at_the_euler_singularity(ctx, gwin):
    """vilo_gauge_fix (k_gauge_fix, kernels_marg.hip) with frame 0 pitched to +-89.5 / +-90 / 88.9 -> 89.2 degrees before and / or after
    the solve, against the reference-built fixture."""
    worst = _gauge_cases(ctx.gauge_fix, gwin)
    print("gauge fix, worst absolute difference to the reference-built fixture: %.2e" % worst)


# ------------------------------------------------------------------------------ marginalisation against the exact Schur complement
def _exact_marginal(gwin, mode):
    from marg_exact import block_table, exact_schur
    w, cfg = gwin, O.default_config()
    w.preint[...] = G["preint"]
    po = synth.PriorData()
    rc, m, A, b = O.marginalize(cfg, w, mode, po, want_A=True)
    assert rc == 0
    H, g = exact_schur(A, b, m)
    return po, H, g, block_table(po)


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_marginalization_vs_exact_schur_complement(gwin, mode):
    """The oracle's prior per kept block pair, scaled by the blocks' own diagonals, against the 60-digit Schur complement of the same
    normal equations. MARGIN_OLD: the reference's route — Amm^-1 from an eigen-decomposition of a matrix of condition 1.6e11
    (marginalization_factor.cpp:281-286) — leaves 2.4e-6 in these units (measured; the compiled reference with the build's eigensolver:
    1.3e-4); MARGIN_SECOND_NEW drops one well-conditioned pose: 4e-14."""
    from marg_exact import scaled_errors
    po, H, g, blocks = _exact_marginal(gwin, mode)
    eh, eb, gh, gb = scaled_errors(po, H, g, blocks)
    print("MEASURED oracle vs exact Schur complement, mode %d: H %.2e, b %.2e in units of the blocks' diagonals (of the largest entry: %.2e, %.2e)" % (mode, eh, eb, gh, gb))
    assert max(eh, eb) < (2e-5 if mode == 0 else 1e-11), (eh, eb)


def test_what_fp64_inputs_allow_for_margin_old(gwin):
    """Why MARGIN_OLD is held to 2e-5 and not to 1e-7 in these units: round the INPUT — the assembled normal equations, every entry moved by
    at most one unit in the last place (relative 1.1e-16) — and the exact (60-digit) Schur complement itself moves by ~6e-6 of the blocks'
    diagonals. Not through Amm's conditioning (Jacobi-scaled, its condition number is 9; unscaled 1.6e11): the kept blocks' information
    is what is left of Arr after subtracting a nearly equal Arm Amm^-1 Amr. Two FP64 implementations assemble A with different rounding,
    so none can be held closer to the exact complement of the OTHER's A than this floor — oracle 2.4e-6, HIP path 3.7e-6, the compiled
    reference (with the build's eigensolver) 1.3e-4."""
    from marg_exact import exact_schur
    w, cfg = gwin, O.default_config()
    w.preint[...] = G["preint"]
    po = synth.PriorData()
    rc, m, A, b = O.marginalize(cfg, w, 0, po, want_A=True)
    H, g = exact_schur(A, b, m)
    d = np.sqrt(np.diag(H))
    rng = np.random.default_rng(0)
    E = rng.uniform(-1, 1, size=A.shape)
    H2, g2 = exact_schur(A * (1 + 1.1e-16 * 0.5 * (E + E.T)), b * (1 + 1.1e-16 * rng.uniform(-1, 1, size=b.shape)), m)
    floor_h, floor_b = (np.abs(H2 - H) / np.outer(d, d)).max(), (np.abs(g2 - g) / d).max()
    dm = np.sqrt(np.diag(A[:m, :m]))
    print("MEASURED one-ulp input rounding moves the exact Schur complement by H %.2e, b %.2e (blocks' diagonals); cond(Amm) %.1e, Jacobi-scaled %.1e"
          % (floor_h, floor_b, np.linalg.cond(A[:m, :m]), np.linalg.cond(A[:m, :m] / np.outer(dm, dm))))
    assert 1e-6 < floor_h < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_marginalization_vs_exact_schur_complement(ctx, gwin, mode):
    """vilo_marginalize (k_marginalize_lds: landmarks eliminated first, the dense frame-0 dimensions by Cholesky, no eigen-decomposition of
    Amm) held to the exact Schur complement per kept block pair in units of the blocks' own diagonals, the oracle in the same units beside
    it. MARGIN_OLD: measured 3.7e-6 (oracle 2.4e-6) against a floor of ~6e-6 that the FP64 rounding of the inputs alone sets
    (test_what_fp64_inputs_allow_for_margin_old); MARGIN_SECOND_NEW: 9e-15."""
    from marg_exact import scaled_errors
    po, H, g, blocks = _exact_marginal(gwin, mode)
    p = synth.PriorData()
    ctx.marginalize(gwin, mode, p)
    eh, eb, gh, gb = scaled_errors(p, H, g, blocks)
    oh, ob, _, _ = scaled_errors(po, H, g, blocks)
    print("MEASURED HIP vs exact Schur complement, mode %d: H %.2e, b %.2e in units of the blocks' diagonals (of the largest entry: %.2e, %.2e); "
          "oracle in the same units: %.2e, %.2e" % (mode, eh, eb, gh, gb, oh, ob))
    assert max(eh, eb) < (2e-5 if mode == 0 else 1e-11), (eh, eb)
