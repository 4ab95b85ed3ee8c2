"""PCIe- and packing-inclusive rate of the host-buffer entry point vilo_solve_windows: everything from host
structs to host results — batch packing, upload, sqrt_info preparation, 12 iterations, download — as the wall time of ONE C call (the
descriptor arrays are built before the clock starts), for a list of "lanes,sub_windows" settings of the call's sub-batch pipeline
(vilo_set_host_pipeline; "0,0" = one batch)."""
import sys
import time

sys.path.insert(0, ".")
from cerberus_amd import _ctypes as T, api, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
settings = sys.argv[2:] or ["0,0", "2,1024", "3,1024", "4,1024", "3,512", "4,512", "3,2048"]
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
ws = [synth.make_window(cfg, n_landmarks=200, seed=900 + i) for i in range(W)]
ctx.preintegrate_windows(ws)
states0 = [w.clone_state() for w in ws]
opts = api.default_solve_opts(True, 12)
descs, states = (T.WindowDesc * W)(), (T.WindowState * W)()
for i, w in enumerate(ws):
    descs[i], states[i] = w.desc(T)
for st in settings:
    lanes, sub = (int(x) for x in st.split(","))
    ctx.set_host_pipeline(lanes, sub)
    best = None
    for rep in range(4):
        for w, s in zip(ws, states0):
            w.set_state(s)
        t0 = time.perf_counter()
        ctx.solve_window_descs(descs, states, opts)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("vilo_solve_windows, %d windows x 12 iterations, pipeline %-7s host to host: %6.1f ms -> %8.0f window-iterations/s" % (W, st, 1e3 * best, W * 12 / best), flush=True)

# Estimator::optimization() as a whole on the same host windows: vilo_optimize_windows = solve + gauge fix + marginalisation (MARGIN_OLD),
# the priors (74 KB per window) come back to host memory
import ctypes as C  # noqa: E402
from cerberus_amd.synth import PriorData  # noqa: E402
outs = [PriorData() for _ in ws]
priors, summ = (T.Prior * W)(), (T.SolveSummary * W)()
for i, o in enumerate(outs):
    priors[i] = o.struct
fl = (C.c_int * W)(*([0] * W))
for st in settings[:1] + settings[-1:]:
    lanes, sub = (int(x) for x in st.split(","))
    ctx.set_host_pipeline(lanes, sub)
    best = None
    for rep in range(3):
        for w, s in zip(ws, states0):
            w.set_state(s)
        t0 = time.perf_counter()
        ctx._check(api.lib().vilo_optimize_windows(ctx.h, W, descs, states, C.byref(opts), fl, priors, summ))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("vilo_optimize_windows, %d windows (12 iterations + marginalisation), pipeline %-7s host to host: %6.1f ms -> %8.0f windows/s" % (W, st, 1e3 * best, W / best), flush=True)
