"""A third, dense restatement of one Ceres 1.14 iteration — numpy on the Jacobian the COMPILED REFERENCE's factor classes return
(tests/ref_gradient.py), no Schur elimination, none of the oracle's or the kernels' code — teacher-forced along the oracle's trajectories:
from every state (x_i, radius_i, mu_i) the oracle visits, rejected ones included, one dense iteration must give the oracle's dogleg scalars
(alpha, |D^-1 g|^2, |Gauss-Newton step|^2), the same kind of step (Gauss-Newton / Cauchy-limited / interpolated), the same
model_cost_change, the same candidate cost (evaluated by the reference's residual blocks) and the same decision. The HIP path is tied to
the oracle on exactly these single steps by tests/test_branches.py (`-m gpu`, 1e-8); this file ties the oracle to the documented algorithm
applied to the reference's own linearisation. What it restates (ceres-solver 1.14 internal/ceres/dogleg_strategy.cc,
trust_region_minimizer.cc, as documented): Jacobi scaling 1 / (1 + |column|); D^2 = clamp(diag(Js^T Js), min_lm_diagonal, max_lm_diagonal);
gradient and Cauchy scale in the D-scaled space; (Js^T Js + mu D^2) y = -Js^T r; the traditional dogleg; step accepted iff
(cost - candidate cost) / model_cost_change > 1e-3."""
import ctypes as C

import numpy as np
import pytest

from cerberus_amd import synth
from oracle import oracle_py as O
from oracle import ref_py as R
import ref_gradient as RG

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref.so not built (needs /root/reference at build time)")


def _window(ocfg, seed, L=40, prior=True, **kw):
    prm = synth.default_params(n_landmarks=L, seed=seed, with_prior=prior)
    for k, v in kw.items():
        setattr(prm, k, v)
    w = synth.make_window(synth.default_config(), params=prm)
    O.fill_preint(ocfg, w)
    return w


def _scalars():
    out = (C.c_double * 12)()
    O.lib().orc_last_step_scalars(out)
    return list(out)


def dense_iteration(cfg, w, radius, mu):
    """One iteration from w's state. Returns dict(alpha, gnorm2, gnnorm2, kind, model, cand, accepted) and applies the step to w if accepted."""
    with R.as_oracle():
        r, J, cols = RG.dense_jacobian(cfg, w)
        cost, _, _ = RG.cost_and_gradient(cfg, w)
    scale = 1.0 / (1.0 + np.linalg.norm(J, axis=0))
    Js = J * scale
    A, g = Js.T @ Js, Js.T @ r
    D2 = np.clip(np.diag(A), 1e-6, 1e32)
    D = np.sqrt(D2)
    gs = g / D
    Jg = Js @ (gs / D)
    alpha = (gs @ gs) / (Jg @ Jg)
    gn = np.linalg.solve(A + mu * np.diag(D2), -g) * D
    ngn, ng = np.linalg.norm(gn), np.linalg.norm(gs)
    if ngn <= radius:
        step, kind = gn, 0
    elif alpha * ng >= radius:
        step, kind = -(radius / ng) * gs, 1
    else:
        b_dot_a = -alpha * (gs @ gn)
        a2 = alpha * alpha * ng * ng
        bma2 = a2 - 2.0 * b_dot_a + ngn * ngn
        c = b_dot_a - a2
        d = np.sqrt(c * c + bma2 * (radius * radius - a2))
        beta = (d - c) / bma2 if c <= 0 else (radius * radius - a2) / (d + c)
        step, kind = (-alpha * (1.0 - beta)) * gs + beta * gn, 2
    y = step / D                              # in the Jacobi-scaled coordinates
    model = -(y @ g + 0.5 * (Js @ y) @ (Js @ y))   # model_cost_change = -(g^T y + y^T Js^T Js y / 2)
    before = w.clone_state()
    RG.apply_step(w, cols, scale * y)
    with R.as_oracle():
        cand, _, _ = RG.cost_and_gradient(cfg, w)
    accepted = model > 0 and (cost - cand) / model > 1e-3
    if not accepted:
        w.set_state(before)
    return dict(alpha=alpha, gnorm2=ng * ng, gnnorm2=ngn * ngn, kind=kind, model=model, cand=cand, accepted=int(accepted), cost=cost)


def _along(ocfg, n_iters, radius0, tol, **wkw):
    O.lib().orc_set_initial_mu.argtypes = [C.c_double]
    kinds, decisions, worst = set(), [], 0.0
    try:
        for i in range(n_iters):
            # the oracle's state, radius and mu after i iterations
            w = _window(ocfg, **wkw)
            O.lib().orc_set_initial_mu(1e-8)
            oo = O.default_opts(True, i)
            oo.initial_trust_region_radius = radius0
            sm = O.solve_window(ocfg, w, oo, check=False)
            sc = _scalars()
            radius_i, mu_i = (sm.radius_trace[i], sc[8]) if i else (radius0, 1e-8)
            x_i = w.clone_state()
            # one oracle iteration from there ...
            w_o = _window(ocfg, **wkw)
            w_o.set_state(x_i)
            O.lib().orc_set_initial_mu(mu_i)
            o1 = O.default_opts(True, 1)
            o1.initial_trust_region_radius = radius_i
            s1 = O.solve_window(ocfg, w_o, o1, check=False)
            sc = _scalars()
            # ... and one dense iteration on the reference's Jacobian
            w_d = _window(ocfg, **wkw)
            w_d.set_state(x_i)
            d = dense_iteration(ocfg, w_d, radius_i, mu_i)
            assert d["kind"] == int(sc[10]) and d["accepted"] == s1.num_successful, (i, d["kind"], sc[10], d["accepted"], s1.num_successful)
            for name, mine, theirs in (("alpha", d["alpha"], sc[0]), ("gnorm2", d["gnorm2"], sc[1]), ("gnnorm2", d["gnnorm2"], sc[2]), ("model", d["model"], sc[3]),
                                       ("cand", d["cand"], sc[4]), ("cost", d["cost"], s1.cost_trace[0])):
                err = abs(mine - theirs) / max(abs(theirs), 1e-300)
                worst = max(worst, err)
                assert err < tol, (i, name, mine, theirs, err, radius_i, mu_i)
            for a, b in zip(w_d.state_arrays(), w_o.state_arrays()):
                if a.size:
                    err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
                    worst = max(worst, err)
                    assert err < tol, (i, err)
            kinds.add(d["kind"]); decisions.append(d["accepted"])
    finally:
        O.lib().orc_set_initial_mu(1e-8)
    return kinds, decisions, worst


@pytest.mark.parametrize("radius", [1e-1, 1e4])
def test_dense_iterations_of_the_three_dogleg_kinds(radius):
    """the windows of tests/test_branches.py::test_single_steps_of_the_three_dogleg_kinds"""
    ocfg = O.default_config()
    kw = dict(seed=11) if radius < 1.0 else dict(seed=20, sig_p=0.2, sig_theta=0.08, sig_lambda_rel=0.6, sig_v=0.5)
    kinds, decisions, worst = _along(ocfg, 8, radius, 1e-7, **kw)
    assert (1 in kinds) if radius < 1.0 else len(kinds) >= 2, kinds
    print("MEASURED dense iteration on the reference's Jacobian vs the oracle, radius %g: kinds %s, worst %.1e" % (radius, sorted(kinds), worst))


def test_dense_iterations_along_a_run_of_rejected_steps():
    """the far-off start of test_single_steps_along_a_trajectory_with_rejected_runs (seed 42): the full Gauss-Newton step with radius 1e8 is
    rejected several times in a row, the radius halves, then accepted steps follow — every decision reproduced"""
    ocfg = O.default_config()
    kinds, decisions, worst = _along(ocfg, 10, 1e8, 1e-5, seed=42, sig_p=1.0, sig_theta=0.4, sig_lambda_rel=0.9, sig_v=1.0, sig_ba=0.3, sig_bg=0.05)
    assert 0 in decisions and 1 in decisions, decisions
    print("MEASURED dense iteration vs the oracle along rejected / accepted steps: decisions %s, worst %.1e" % (decisions, worst))
