"""Wall time of vilo_batch_prepare (sqrt_info of every preintegration record of a resident batch) against the batch size.
Usage on the GPU box: python tools/time_prepare.py"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from cerberus_amd import api, synth  # noqa: E402

cfg = synth.default_config()
ctx = api.Context(cfg, 0)
base = [synth.make_window(cfg, n_landmarks=40, seed=20260925 + i) for i in range(64)]
ctx.preintegrate_windows(base)
for W in (1, 26, 103, 205, 410, 1024, 4096):
    ws = [base[i % 64] for i in range(W)]
    b = api.Batch(ctx, ws)
    for _ in range(3):
        b.prepare()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        b.prepare()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("W %5d  records %6d  prepare %.3f ms  = %.2f us per record-slot of 1024 SIMDs" % (W, W * 10, dt * 1e3, dt * 1e6 / max(1.0, W * 10 / 1024.0)))
    del b
