"""PCIe- and packing-inclusive rate of the host-buffer entry point vilo_solve_windows: everything from
host structs to host results — batch packing, upload, sqrt_info preparation, 12 iterations, download — per call."""
import sys
import time

sys.path.insert(0, ".")
from cerberus_amd import api, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
ws = [synth.make_window(cfg, n_landmarks=200, seed=900 + i) for i in range(W)]
ctx.preintegrate_windows(ws)
states = [w.clone_state() for w in ws]
opts = api.default_solve_opts(True, 12)
for rep in range(3):
    for w, s in zip(ws, states):
        w.set_state(s)
    t0 = time.perf_counter()
    ctx.solve_windows(ws, opts)
    dt = time.perf_counter() - t0
print("vilo_solve_windows, %d windows x 12 iterations, host to host: %.1f ms -> %.0f window-iterations/s (device-resident loop: %.1f ms)" % (W, 1e3 * dt, W * 12 / dt, api.lib().vilo_last_solve_ms(ctx.h)))
