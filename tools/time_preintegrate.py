"""Wall time of vilo_preintegrate (IMU-leg contact preintegration, host buffers in and out) for W windows x 10 intervals.
Usage on the GPU box: python tools/time_preintegrate.py [W]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberus_amd import api, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
ws = [synth.make_window(cfg, n_landmarks=8, seed=700 + i) for i in range(W)]
for rep in range(3):
    t0 = time.perf_counter()
    ctx.preintegrate_windows(ws)
    dt = time.perf_counter() - t0
    print("vilo_preintegrate: %d intervals in %.1f ms (host to host, python packing included)" % (10 * W, 1e3 * dt))
