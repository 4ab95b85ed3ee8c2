"""Shader-clock phase breakdown (s_memtime stamps) of one linearisation in k_solve_wave.
Usage on the GPU box: python tools/phase_clocks_wave.py [windows] [landmarks]"""
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
b.solve(api.default_solve_opts(True, 3))
names = ["gathers + scaling of the speed / leg-bias part", "tile load, scaling, q, landmark pass 1", "bias chain + T recurrence + rank update",
         "landmark Schur", "Cholesky 80", "triangular solves", "bias back-substitution", "landmark back-substitution + norms", "dogleg + candidate"]
acc = np.zeros(9)
sample = list(range(0, W, max(1, W // 64)))
for w in sample:
    c = b.fetch(12, w).view(np.int64)
    acc += np.diff(c[:10].astype(np.float64))
acc /= len(sample)
print("k_solve_wave phases (cycles, mean over %d windows of %d):" % (len(sample), W))
for n, a in zip(names, acc):
    print("  %-52s %10.0f" % (n, a))
print("  %-52s %10.0f" % ("total", acc.sum()))
c = b.fetch(12, 0).view(np.int64)
print("chain, frame 5 (window 0): loads + S_k %d  chol13 %d  substitutions %d  S update + operands %d  MFMA (T, V, rank update) %d" %
      (c[16] - c[16], c[17] - c[16], c[18] - c[17], c[19] - c[18], c[20] - c[19]))
print("bias back-substitution (window 0): M / T_A copy %d  c = g - B yP %d  forward sweep %d  backward sweep %d" %
      (c[21] - c[6], c[22] - c[21], c[23] - c[22], c[7] - c[23]))
vis = np.zeros(5)
for w in sample:
    c = b.fetch(12, w).view(np.int64)
    vis += c[28:33].astype(np.float64)
vis /= len(sample)
print("k_visual_linearize, first packed wave of a window (%d lanes, %d frames): total %d cycles, factor evaluation + row staging %d, Gram MFMA pass %d" %
      (vis[3], vis[4], vis[0], vis[1], vis[2]))
imu = np.zeros(3)
for w in sample:
    c = b.fetch(12, w).view(np.int64)
    imu += c[33:36].astype(np.float64)
imu /= len(sample)
print("k_imu_linearize, first factor of a window: sqrt_info operand loads %d cycles, whitening %d, Gram + stores %d" % (imu[0], imu[1], imu[2]))
