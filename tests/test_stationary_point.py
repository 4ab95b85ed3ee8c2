"""What can be said about "the reference Ceres solution" without Ceres: whatever path a trust-region method takes, where it converges the
gradient of the cost vanishes. The cost function is not ours — it is the sum of the reference's OWN residual blocks, and
tests/ref_gradient.py assembles its value and gradient in Python from `Evaluate()` of the compiled reference classes
(oracle/_ref/libref.so: IMULegFactor, the three projection factors, MarginalizationFactor, built from /root/reference), enumerated as
Estimator::optimization adds them, with the Huber loss and the pose parameterisation applied as Ceres applies them. Neither solver under
test is involved in that evaluation. Checked at the state the oracle (CPU) and the HIP path (`-m gpu`) reach after 40 iterations:
  * the cost the solver reports IS the reference's cost at that state (1e-12),
  * the gradient, every entry in units of the square root of its Gauss-Newton diagonal, is ten orders of magnitude below the start's and
    below 1e-4 in absolute terms (the floor of evaluating it in FP64: the IMU rows carry information up to 1e14).
So the HIP solver stops where any convergent minimiser of the reference's problem stops; what stays unpinned is the trajectory there."""
import numpy as np
import pytest

from cerberus_amd import synth
from oracle import oracle_py as O
from oracle import ref_py as R
import ref_gradient as RG

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so not built (needs /root/reference at build time)")
CASES = [(41, 30), (42, 60), (5, 24)]


def _whitened(w, g, h):
    return np.abs(RG.free_gradient(w, g) / np.sqrt(RG.free_gradient(w, h))).max()


def _check(cfg, w, final_cost, start):
    with R.as_oracle():
        c1, g1, h1 = RG.cost_and_gradient(cfg, w)
    wf = _whitened(w, g1, h1)
    assert abs(c1 - final_cost) <= 1e-12 * c1, (c1, final_cost)
    assert wf < 1e-4 and wf < 1e-9 * start, (wf, start)
    return wf


@pytest.mark.parametrize("seed,L", CASES)
def test_the_oracle_stops_at_a_stationary_point_of_the_references_cost(seed, L):
    cfg = O.default_config()
    w = synth.make_window(synth.default_config(), n_landmarks=L, seed=seed)
    O.fill_preint(cfg, w)
    with R.as_oracle():
        c0, g0, h0 = RG.cost_and_gradient(cfg, w)
    assert abs(c0 - O.window_cost(cfg, w)) <= 1e-14 * c0            # (the enumeration is the one the solvers use)
    start = _whitened(w, g0, h0)
    sm = O.solve_window(cfg, w, O.default_opts(True, 40))
    wf = _check(cfg, w, sm.final_cost, start)
    print("MEASURED oracle, seed %d: whitened gradient %.1e -> %.1e" % (seed, start, wf))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,L", CASES)
def test_the_hip_solver_stops_at_a_stationary_point_of_the_references_cost(seed, L):
    from cerberus_amd import api
    cfg, ocfg = synth.default_config(), O.default_config()
    ctx = api.Context(cfg, 0)
    try:
        w = synth.make_window(cfg, n_landmarks=L, seed=seed)
        ctx.preintegrate_window(w)
        with R.as_oracle():
            c0, g0, h0 = RG.cost_and_gradient(ocfg, w)
        start = _whitened(w, g0, h0)
        sm = ctx.solve_windows([w], api.default_solve_opts(True, 40))[0]
        assert abs(sm.initial_cost - c0) <= 1e-10 * c0
        wf = _check(ocfg, w, sm.final_cost, start)
        print("MEASURED HIP solver, seed %d: whitened gradient of the reference's cost %.1e -> %.1e, cost %.6f" % (seed, start, wf, sm.final_cost))
    finally:
        ctx.close()


# ------------------------------------------------------------------------------ one iteration = the Gauss-Newton step of the reference's linearisation
def _dense_gauss_newton_step(cfg, w):
    """One iteration as Ceres 1.14 takes it near a solution, in dense numpy algebra on the compiled reference's Jacobians: Jacobi scaling
    1 / (1 + |column|) (first iteration), D^2 = clamp(diag(Js^T Js), 1e-6, 1e32), (Js^T Js + mu D^2) y = -Js^T r with mu = min_mu = 1e-8;
    the Gauss-Newton point is the step when |D y| <= radius (1e4). Returns (delta in local coordinates, columns, |D y|)."""
    with R.as_oracle():
        r, J, cols = RG.dense_jacobian(cfg, w)
    scale = 1.0 / (1.0 + np.linalg.norm(J, axis=0))
    Js = J * scale
    A, g = Js.T @ Js, Js.T @ r
    D2 = np.clip(np.diag(A), 1e-6, 1e32)
    y = np.linalg.solve(A + 1e-8 * np.diag(D2), -g)
    return scale * y, cols, float(np.sqrt((D2 * y * y).sum()))


def _step_case(seed, L, solve_one):
    """six oracle iterations to get near the solution, then ONE iteration of the solver under test from there against the dense step"""
    cfg = O.default_config()
    w = synth.make_window(synth.default_config(), n_landmarks=L, seed=seed)
    O.fill_preint(cfg, w)
    O.solve_window(cfg, w, O.default_opts(True, 6))
    x6 = w.clone_state()
    delta, cols, dy = _dense_gauss_newton_step(cfg, w)
    assert dy < 1e4                                   # inside the initial trust region: the dogleg step IS the Gauss-Newton step
    w_np, w_sv = (synth.make_window(synth.default_config(), n_landmarks=L, seed=seed) for _ in range(2))
    for v in (w_np, w_sv):
        v.preint[...] = w.preint
        v.set_state(x6)
    RG.apply_step(w_np, cols, delta)
    sm = solve_one(w_sv)
    assert (sm.iterations, sm.num_successful) == (1, 1)
    worst = 0.0
    for a, b, c in zip(w_np.state_arrays(), w_sv.state_arrays(), x6):
        if a.size == 0:
            continue
        step = np.abs(b - c).max()
        err = np.abs(a - b).max()
        assert err <= 1e-6 * step + 1e-12, (err, step)
        worst = max(worst, err / max(step, 1e-300))
    return worst


@pytest.mark.parametrize("seed,L", CASES)
def test_one_oracle_iteration_is_the_dense_gauss_newton_step_of_the_references_jacobian(seed, L):
    cfg = O.default_config()
    worst = _step_case(seed, L, lambda w: O.solve_window(cfg, w, O.default_opts(True, 1)))
    print("MEASURED oracle step vs dense numpy step on the reference's Jacobian, seed %d: %.1e of the step" % (seed, worst))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,L", CASES)
def test_one_hip_iteration_is_the_dense_gauss_newton_step_of_the_references_jacobian(seed, L):
    """The Schur-eliminated, Jacobi-scaled, mu-regularised solve of the HIP kernels (landmarks eliminated, block-tridiagonal chain, Cholesky
    of the 80 x 80 pose system) against ONE dense numpy.linalg.solve on the Jacobian the compiled reference's classes return."""
    from cerberus_amd import api
    ctx = api.Context(synth.default_config(), 0)
    try:
        worst = _step_case(seed, L, lambda w: ctx.solve_windows([w], api.default_solve_opts(True, 1))[0])
        print("MEASURED HIP step vs dense numpy step on the reference's Jacobian, seed %d: %.1e of the step" % (seed, worst))
    finally:
        ctx.close()


# ------------------------------------------------------------------------------ USE_LEG 0: IMUFactor instead of IMULegFactor
def _vins_window(seed, L):
    cfg = O.default_config()
    w = synth.make_window(synth.default_config(), n_landmarks=L, seed=seed, with_prior=False)
    w.use_leg = 0
    O.fill_preint(cfg, w)
    return w


def _vins_step(solve_one, seed=51, L=40):
    """config/a1_config/hardware_a1_vins_config.yaml (USE_LEG 0): near the solution one iteration against the dense step on the reference's
    IMUFactor / projection Jacobians. No prior: the four gauge directions are regularised by mu D^2 alone, so the step is compared where it
    is determined — through the cost it reaches and the non-gauge part (positions relative to frame 0 would need the gauge fixed; the
    candidate cost and model decrease do not)."""
    cfg = O.default_config()
    w = _vins_window(seed, L)
    O.solve_window(cfg, w, O.default_opts(True, 6))
    x6 = w.clone_state()
    with R.as_oracle():
        r, J, cols = RG.dense_jacobian(cfg, w)
        c6, _, _ = RG.cost_and_gradient(cfg, w)
    assert not any(k[0] == 2 for k in cols)          # (no leg-bias block in the problem)
    scale = 1.0 / (1.0 + np.linalg.norm(J, axis=0))
    Js = J * scale
    A, g = Js.T @ Js, Js.T @ r
    D2 = np.clip(np.diag(A), 1e-6, 1e32)
    y = np.linalg.solve(A + 1e-8 * np.diag(D2), -g)
    assert np.sqrt((D2 * y * y).sum()) < 1e4
    w_np = _vins_window(seed, L); w_np.set_state(x6)
    RG.apply_step(w_np, cols, scale * y)
    with R.as_oracle():
        c_np, _, _ = RG.cost_and_gradient(cfg, w_np)
    w_sv = _vins_window(seed, L); w_sv.set_state(x6)
    sm = solve_one(w_sv)
    assert (sm.iterations, sm.num_successful) == (1, 1)
    assert abs(sm.initial_cost - c6) <= 1e-10 * c6
    assert abs(sm.final_cost - c_np) <= 1e-9 * c_np, (sm.final_cost, c_np)
    return abs(sm.final_cost - c_np) / c_np, c6 - c_np


def test_one_oracle_iteration_without_leg_factors_reaches_the_dense_steps_cost():
    cfg = O.default_config()
    err, dec = _vins_step(lambda w: O.solve_window(cfg, w, O.default_opts(True, 1)))
    print("MEASURED USE_LEG 0, oracle: cost after one iteration vs after the dense step on the reference's Jacobian %.1e (decrease %.3g)" % (err, dec))


@pytest.mark.gpu
def test_one_hip_iteration_without_leg_factors_reaches_the_dense_steps_cost():
    from cerberus_amd import api
    ctx = api.Context(synth.default_config(), 0)
    try:
        err, dec = _vins_step(lambda w: ctx.solve_windows([w], api.default_solve_opts(True, 1))[0])
        print("MEASURED USE_LEG 0, HIP: cost after one iteration vs after the dense step on the reference's Jacobian %.1e (decrease %.3g)" % (err, dec))
    finally:
        ctx.close()
