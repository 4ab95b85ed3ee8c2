"""Worker of test_solver_forms.py: solves a fixed set of windows with the solver form the environment pins (VILO_SOLVER, read once per
process by the library) and prints the results as JSON."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from cerberus_amd import api, synth  # noqa: E402


def window(cfg, seed, L=40, prior=True, **kw):
    prm = synth.default_params(n_landmarks=L, seed=seed, with_prior=prior)
    for k, v in kw.items():
        setattr(prm, k, v)
    return synth.make_window(cfg, params=prm)


def run(ctx, ws, opts):
    ctx.preintegrate_windows(ws)
    b = api.Batch(ctx, ws)
    try:
        b.solve(opts)
        summ = b.download()
        retries = [int(b.fetch(10, i)[18:24].view(np.int32)[10]) for i in range(len(ws))]
    finally:
        b.close()
    return [{"state": [a.tolist() for a in w.state_arrays()], "cost_trace": list(s.cost_trace[:s.iterations + 1]),
             "radius_trace": list(s.radius_trace[:s.iterations + 1]), "iterations": s.iterations, "successful": s.num_successful,
             "termination": s.termination, "retries": r} for w, s, r in zip(ws, summ, retries)]


cfg = synth.default_config()
ctx = api.Context(cfg, 0)
out = {}
out["plain"] = run(ctx, [window(cfg, 42), window(cfg, 48, L=90), window(cfg, 51, L=12), window(cfg, 7, L=200)], api.default_solve_opts(True, 8))
o = api.default_solve_opts(True, 8)
o.min_lm_diagonal = o.max_lm_diagonal = 1e-12
out["escalation"] = run(ctx, [window(cfg, 60, prior=False), window(cfg, 61, prior=False)], o)
# a far-off start with a huge radius: runs of rejected steps, each reusing the linearisation (tests/test_branches.py)
o = api.default_solve_opts(True, 12)
o.initial_trust_region_radius = 1e8
far = dict(sig_p=1.0, sig_theta=0.4, sig_lambda_rel=0.9, sig_v=1.0, sig_ba=0.3, sig_bg=0.05)
out["rejected"] = run(ctx, [window(cfg, 42, **far), window(cfg, 51, **far)], o)
ctx.close()
print("FORMS_JSON " + json.dumps(out))
