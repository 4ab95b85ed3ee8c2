"""Wall time of vilo_marginalize (MARGIN_OLD) on a batch of windows, host packing and PCIe included (it runs once per
frame, outside the iteration loop that bench.py measures). Usage on the GPU box: python tools/time_marginalize.py [W]"""
import ctypes as C
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberus_amd import api, synth, _ctypes as T  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
ws = [synth.make_window(cfg, n_landmarks=200, seed=500 + i) for i in range(W)]
ctx.preintegrate_windows(ws)
descs = (T.WindowDesc * W)(); states = (T.WindowState * W)(); priors = (T.Prior * W)()
outs = [synth.PriorData() for _ in range(W)]
for i, w in enumerate(ws):
    descs[i], states[i] = w.desc(T)
    priors[i] = outs[i].struct
for mode in (0, 1):
    for rep in range(2):
        t0 = time.perf_counter()
        rc = api.lib().vilo_marginalize(ctx.h, W, descs, states, mode, priors)
        dt = time.perf_counter() - t0
    L = api.lib()
    L.vilo_last_marginalize_ms.restype = C.c_double
    print("mode %d: rc %d, %d windows in %.1f ms = %.3f ms per window (n = %d); kernels %.2f ms, %d windows on the general eigen path"
          % (mode, rc, W, 1e3 * dt, 1e3 * dt / W, priors[0].n, L.vilo_last_marginalize_ms(ctx.h), L.vilo_debug_marg_general_count(ctx.h)))
