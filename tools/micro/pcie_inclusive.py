"""What a caller who hands over HOST windows gets (run on the GPU box): vilo_batch_create (pack + upload) + prepare + solve + download per
batch against the resident solve alone. usage: VILO_HOST_TIMING=1 python tools/micro/pcie_inclusive.py [windows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cerberus_amd import api, synth
import bench

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
ws = [bench.make_synth_window(cfg, 200, 500, 20260925 + g) for g in range(W)]
ctx.preintegrate_windows(ws)
opts = api.default_solve_opts(fixed_iterations=True, max_num_iterations=12)
for rep in range(3):
    t0 = time.perf_counter()
    b = api.Batch(ctx, ws)
    t1 = time.perf_counter()
    b.prepare(); gpu_ms = b.solve(opts)
    t2 = time.perf_counter()
    out = b.download() if hasattr(b, "download") else None
    t3 = time.perf_counter()
    print("W=%d create %.1f ms  prepare+solve %.1f ms (gpu %.1f)  download %.1f ms  -> %.0f window-iterations/s with the hand-over, %.0f resident"
          % (W, 1e3 * (t1 - t0), 1e3 * (t2 - t1), gpu_ms, 1e3 * (t3 - t2), 12 * W / (t3 - t0), 12 * W / (t2 - t1)), flush=True)
    del b
