"""What the launches behind a converged window cost: one window solved with max_num_iterations = 12 (the reference's setting; the launch
sequence is fixed, kernels of iterations after convergence return at once) against the same window with max_num_iterations = the iterations
it took. GPU time between the solve's two events, best of 20."""
import sys

sys.path.insert(0, ".")
from cerberus_amd import api, synth  # noqa: E402

cfg = synth.default_config()
ctx = api.Context(cfg, 0)
for seed in (11, 12, 13):
    w = synth.make_window(cfg, n_landmarks=170, seed=seed)
    ctx.preintegrate_windows([w])
    s0 = w.clone_state()

    def run(n):
        best, summ = None, None
        for _ in range(20):
            w.set_state(s0)
            b = api.Batch(ctx, [w])
            ms = b.solve(api.default_solve_opts(False, n))
            summ = b.download()[0]
            b.close()
            best = ms if best is None else min(best, ms)
        return best, summ
    t12, s12 = run(12)
    k = s12.iterations
    tk, sk = run(max(1, k))
    tk1, _ = run(max(1, k) + 1)
    print("seed %d: 12 launched, %d iterations done: %.3f ms; %d launched: %.3f ms (%d done, cost %.6g vs %.6g); %d launched: %.3f ms -> %.1f us per iteration's launches behind convergence"
          % (seed, k, t12, k, tk, sk.iterations, sk.final_cost, s12.final_cost, k + 1, tk1, 1e3 * (t12 - tk) / max(1, 12 - k)))
