"""Shader-clock phase breakdown (profiling build: VILO_BUILD_PROF=1 python __graft_entry__.py, then
VILO_GPU_LIB=cerberus_amd/lib/libvilo_gpu_prof.so python tools/phase_clocks_r3.py [windows ...]) of k_assemble(_c) and of the solver the
batch size selects (k_solve_mw8 up to 512 windows, k_solve_wave beyond; the `k_solve_mw` lines below print only for builds that still carry the
two-wave form of rounds 3 - 4)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from cerberus_amd import api, synth  # noqa: E402

cfg = synth.default_config()
ctx = api.Context(cfg, 0)
for W in [int(a) for a in sys.argv[1:]] or [1, 128, 4096]:
    ws = [synth.make_window(cfg, n_landmarks=200, seed=20260925 + i) for i in range(W)]
    ctx.preintegrate_windows(ws)
    b = api.Batch(ctx, ws)
    b.solve(api.default_solve_opts(True, 3))
    sample = list(range(0, W, max(1, W // 32)))
    C = np.stack([b.fetch(12, w).view(np.int64).astype(np.float64) for w in sample])
    m = C.mean(axis=0)
    print("== %d windows (mean of %d sampled windows, cycles)" % (W, len(sample)))
    if C[0, 30] > 0:   # producer / consumer visual kernel: packed wave 0 (window 0)
        print("k_visual_linearize_pc (packed wave 0, %d frames): consumer %d cycles, %d of them at the step barriers; producer %d, %d at the barriers"
              % (C[0, 32], C[0, 28], C[0, 29], C[0, 30], C[0, 31]))
        if C[0, 60] > 0:
            print("   producer: pair tables %d | factor evaluation (lane 0's stamps) %d | row stores %d" % (C[0, 59], C[0, 60], C[0, 61]))
    if W >= 513 and C[:, 46].max() > 0:   # full batch: k_assemble_pose + k_assemble_bias (kernels_asm_full.hip)
        print("k_assemble_pose: bookkeeping + prior image %d | visual slots (passes) %d | IMU pose blocks %d | gradient + scaling %d | tile image out + q %d | sums %d | total %d"
              % (m[37] - m[36], m[38] - m[37], m[39] - m[38], m[41] - m[39], m[42] - m[41], m[45] - m[42], m[45] - m[36]))
        print("k_assemble_pose bookkeeping: loads landed + image stored %d | dx scattered, H dx partials %d | H dx + cost partials %d | decision %d | state copy until the passes start %d"
              % (m[12] - m[36], m[13] - m[12], m[14] - m[13], m[15] - m[14], m[37] - m[15]))
        print("k_assemble_pose pass loop per wave (wait at the first barrier | stage fill + second barrier | class bodies): " +
              "  ".join("w%d %d|%d|%d" % (w, m[47 + 3 * w], m[48 + 3 * w], m[49 + 3 * w]) for w in range(4)))
        print("k_assemble_bias: prologue + IMU factors (frame loop) %d | scaling + prior rows %d | q of the speed / leg-bias rows + sums %d | total %d"
              % (m[43] - m[40], m[44] - m[43], m[46] - m[44], m[46] - m[40]))
    else:
        d = np.diff(C[:, 36:46], axis=1).mean(axis=0)
        print("k_assemble, visual slots: work between the chunk barriers per wave (0: T1 T2 T3, 1: T8, 2: T5 T6 T4, 3: T7): %d %d %d %d" % tuple(m[12:16]))
        print("k_assemble: prior image %d | visual slots %d | IMU factors (frame loop) %d | diagonal + gradient %d | scaling %d | tile image out + q %d | prior rows %d | q of the speed / leg-bias rows %d | sums %d | total %d"
              % (*d, d.sum()))
    if m[34] > 0:
        print("IMU factor 0 (imu_fused_body / k_imu_linearize): stage %d | raw evaluation on lane 0 %d | whitening + Gram %d cycles" % (m[33], m[34], m[35]))
    if m[46] > 0 and W < 513:
        print("k_assemble_s: trust-region bookkeeping (accept_body) before the assembly: %d cycles" % m[46])
    if W <= 256 and m[16] > 0:
        print("k_solve_mw8 wave B1 (cycles from kernel start): scaling %d | main loop, own work %d, with the tiles' hand-over %d | (unused) %d | Cholesky + solves %d | back-substitutions %d | norms %d | dogleg + candidate %d;  C1: chain role %d, sweeps %d"
              % (m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9]))
    if W <= 256 and m[16] > 0:   # four-wave solver: total and barrier-wait cycles of its waves
        print("k_solve_mw8, cycles at the main loop's step barriers: C1 %d C2 %d | TD %d TU %d | B1 %d B2 %d B3 %d B4 %d" % tuple(m[16:24]))
        print("k_solve_mw8, C1's second frame (cycles from its start): S_k in registers %d | Cholesky-13 %d | [T_A | M] by substitution %d | handed to LDS %d | neighbour's update %d" % tuple(m[24:29]))
        d = np.diff(np.concatenate([[0.0], m[47:58]]))
        print("k_solve_mw8, Cholesky-80 on the factor wave (cycles per phase; block columns 0 .. 4: diagonal tile factored | its panel rows solved + next diagonal tile updated): "
              + " ".join("%d|%d" % (d[2 * j], d[2 * j + 1]) for j in range(5)) + "; backward solve %d; total %d" % (d[10], m[57]))
        print("k_solve_mw8, B3's fourth iteration (cycles from its start): the slice's half trips %d | two frames' rank updates %d" % (m[10], m[11]))
    elif W <= 512:
        print("k_solve_mw wave B: scaling %d | 1/(E + mu d) %d | Schur + rank updates (steps) %d | rhs + Cholesky %d | backward solve %d | wait for barrier %d | landmark back-substitution %d | norms + barrier %d | dogleg + candidate %d | total %d"
              % (m[1] - m[0], m[2] - m[1], m[3] - m[2], m[4] - m[3], m[5] - m[4], m[6] - m[5], m[7] - m[6], m[8] - m[7], m[9] - m[8], m[9] - m[0]))
        print("k_solve_mw wave A: chain %d (from kernel start %d) | idle until y_P %d | c = g_B - B y_P %d | forward sweep %d | backward sweep %d"
              % (m[17] - m[16], m[16] - m[0], m[18] - m[17], m[19] - m[18], m[20] - m[19], m[21] - m[20]))
    elif C[:, 11].max() > 0:   # three-stage form
        print("k_chain: %d" % (m[11] - m[10]))
        print("k_solve_mid: gathers %d | 1/(E + mu d) %d | tile load + rank updates from T(k) %d | landmark Schur %d | Cholesky 80 %d | backward solve %d | total %d"
              % (m[1] - m[0], m[2] - m[1], m[3] - m[2], m[4] - m[3], m[5] - m[4], m[6] - m[5], m[6] - m[0]))
        print("k_backsub: bias back-substitution %d | landmark back-substitution %d | dogleg + candidate %d | total %d"
              % (m[25] - m[24], m[26] - m[25], m[27] - m[26], m[27] - m[24]))
    else:
        names = ["gathers", "tile load + 1/(E + mu d)", "chain", "landmark Schur", "Cholesky 80", "backward solve", "bias back-substitution", "landmark back-substitution", "dogleg + candidate"]
        d = np.diff(C[:, 0:10], axis=1).mean(axis=0)
        print("k_solve_wave: " + " | ".join("%s %d" % (n, v) for n, v in zip(names, d)) + " | total %d" % d.sum())
    b.close()
