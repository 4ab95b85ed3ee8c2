"""Shader-clock phase breakdown (clock64 stamps, 100 MHz-independent: s_memtime counts at the shader clock) of one
linearisation: k_build_solve phases, k_visual_linearize (first chunk of the window) and k_imu_linearize (factor 0).
Usage on the GPU box: python tools/phase_clocks.py [windows] [landmarks]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from cerberus_amd import api, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = synth.default_config()
ctx = api.Context(cfg, 0)
ws = [synth.make_window(cfg, n_landmarks=L, seed=20260925 + i) for i in range(W)]
ctx.preintegrate_windows(ws)
b = api.Batch(ctx, ws)
opts = api.default_solve_opts(True, 3)
b.solve(opts)
names = ["tables", "tables", "prior image copy", "Gram scatter (owner-computes)", "masks + tile load", "P5/P6 scaling+q", "landmark Schur || bias chain", "T recurrence + rank-143", "Cholesky 80",
         "triangular solves", "B back-sub", "landmark back-sub+norms", "dogleg+candidate"]
acc = np.zeros(13)
vis = np.zeros(5)
imu = np.zeros(3)
sample = list(range(0, W, max(1, W // 32)))
for w in sample:
    c = b.fetch(12, w).view(np.int64)
    d = np.diff(c[:13].astype(np.float64))
    acc[1:] += d
    vis += c[16:21]
    imu += c[24:27]
acc /= len(sample); vis /= len(sample); imu /= len(sample)
print("k_build_solve phases (cycles, mean over %d windows):" % len(sample))
for i in range(1, 13):
    if names[i] != "-":
        print("  %-28s %10.0f" % (names[i], acc[i]))
print("  %-28s %10.0f" % ("total", acc[1:].sum()))
c2 = b.fetch(12, 0).view(np.int64)
print("scatter sub-phases (window 0, wave 0): visual %d  imu %d  prior-g %d  Bs %d" % (c2[13]-c2[2], c2[14]-c2[13], c2[15]-c2[14], c2[3]-c2[15]))
print("masks: loops %d  barrier %d  tile_load3 %d  diag+barrier %d | P5 %d  q tiles %d  q bias part(t96) %d  P6 loop %d  block sums %d" % (c2[21]-c2[3], c2[22]-c2[21], c2[23]-c2[22], c2[4]-c2[23], c2[27]-c2[4], c2[28]-c2[27], c2[29]-c2[28], c2[30]-c2[29], c2[5]-c2[30]))
print("mask-phase arrival of waves 1..3 relative to wave 0: %d %d %d" % (c2[33]-c2[21], c2[34]-c2[21], c2[35]-c2[21]))
print("Schur pass alone (wave 0, from block sums to its end): %d" % (c2[40]-c2[5]))
print("bias chain frame 5: chol13 %d  substitutions %d  S update %d" % (c2[37]-c2[36], c2[38]-c2[37], c2[39]-c2[38]))
print("k_visual_linearize first chunk: total %.0f  proj %.0f  gram %.0f  (n=%.1f kmax=%.1f)" % tuple(vis))
print("k_imu_linearize factor 0: raw %.0f  whiten %.0f  gram %.0f" % tuple(imu))
