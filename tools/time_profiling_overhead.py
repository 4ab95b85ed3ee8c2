import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
from cerberus_amd import api, synth
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
W = 4096
ws = [synth.make_window(cfg, n_landmarks=200, seed=20260925 + i) for i in range(W)]
ctx.preintegrate_windows(ws)
b = api.Batch(ctx, ws)
opts = api.default_solve_opts(True, 12)
L = api.lib()
for prof in (1, 0, 1, 0):
    L.vilo_set_profiling(ctx.h, prof)
    for _ in range(2):
        b.reset(); b.prepare(); b.solve(opts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        b.reset(); b.prepare(); b.solve(opts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("profiling %d: %.3f ms per step -> %.0f window-iterations/s" % (prof, dt * 1e3, W * 12 / dt))
