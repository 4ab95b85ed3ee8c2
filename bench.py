#!/usr/bin/env python
"""bench.py — Gauss-Newton (dogleg trust-region) iterations/s of the sliding-window VILO solve on MI355X.

Contract (see task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches
one rank per GPU through torch.distributed.run. A "step" is one pass of the hot path
(Estimator::optimization()'s solve half, estimator.cpp:1054-1245) over one batch of synthetic windows per GPU:
device-side state reset + 12 fixed trust-region iterations on every window. Windows are independent, so ranks
share nothing (weak scaling, no data-path collective); `value` = window-iterations of all ranks / max-over-ranks
time. Inputs are resident in HBM before the timed region (batch created + preintegrated during setup).

Workload = BASELINE.json configs[1]: synthetic 10-KF x 200-landmark window, A1 4-leg contact preintegration,
500 Hz IMU/leg samples; `--windows` independent instances per GPU (default 32768; config 4 batches 1024 over 8 GPUs).
`--config 3` = BASELINE.json configs[2]: 1000 landmarks (NUM_OF_F, parameters.h:24), 400 Hz samples (27 per interval), and every
iteration integrates all 10 intervals of every window again (IMULegIntegrationBase::repropagate, imu_leg_integration_base.cpp:62-86)
at the biases of the point it linearises, sqrt_info of the new covariances included.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS = 12  # max_num_iterations of the reference (config yaml:86), fixed so every window does identical work


def algorithmic_bytes(sum_k, L, F=11, n_prior=86):
    """compulsory HBM traffic of one window-iteration."""
    return 88 * sum_k + 16 * L + 8 * (20 * F + 15 + L) + 10 * 8696 + 8 * (n_prior * n_prior + n_prior + 98) + 8 * (20 * F + 15 + L)


ALG_FLOPS_PER_WINDOW_ITERATION = {200: 13.7e6, 1000: 47.0e6}   # FP64 flops of one window-iteration (FMA = 2)
ITERATION_KERNELS = ("k_visual_linearize", "k_imu_raw", "k_imu_linearize", "k_assemble", "k_assemble_bias", "k_chain", "k_solve_mid", "k_backsub", "k_solve_wave")   # launched once per iteration (k_assemble = k_assemble_pose: the bookkeeping is its first phase; k_accept runs once per solve, for the last candidate)
REPROPAGATION_KERNELS = ("k_repropagate", "k_prepare_preint")   # config 3: once per iteration as well
# K1 re-propagation adds the raw samples (280 B each) to the compulsory traffic and ~15 Mflop (sparse-aware; 75 Mflop as
# dense 31 x 31 products) to the flops of one window-iteration
REPROPAGATION_FLOPS = 15.0e6


def sustained_mfma_tflops():
    """(all SIMDs fed by four waves, one wave per SIMD) FP64-MFMA TFLOP/s by wall clock from the committed output of tools/micro/mfma_f64_rate.hip
    (15 independent accumulators per wave); (None, None) if the file is not there."""
    import re
    try:
        txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round3_mfma_f64_rate.txt")).read()
        part = txt.split("-- 15 independent accumulators per wave")[1]
        rate = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"waves per CU\s+(\d+):.*?->\s+([0-9.]+) TFLOP/s", part)}
        return rate.get(16), rate.get(4)
    except Exception:
        return None, None


def kernels_sha16():
    """Fingerprint of the kernel sources (cerberus_amd/csrc): profiles/*_pmc.json carries the one of the tree its counters were collected
    on, so a bench line can say whether the committed counter evidence belongs to the kernels that ran."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cerberus_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PROFILE_ROUND = "round6"
# windows per GPU of the headline line: the kernels of an iteration are launched once for all windows of the batch, and every kernel's tail (its
# last partial round of workgroups) and launch gap is paid once per batch — measured on one MI355X with the round-5 kernels: 4096 windows 1.77 M,
# 8192 1.835 M, 12288 1.864 M, 16384 1.877 M, 24576 1.901 M, 32768 1.911 M window-iterations/s (36 GB of the 288 GB resident)
DEFAULT_WINDOWS = 32768
STRONG_TOTAL = 1024   # BASELINE configs[3]


def profile_evidence(W_run, tag=""):
    """Counter evidence of the committed rocprofv3 passes (tools/profile_gpu.sh, tools/profile_sq.sh; `windows_per_dispatch` of the
    file says at which batch size): calibrated HBM bytes per dispatch and the matrix-core busy fraction per kernel. Per-window figures, so they scale."""
    out = {"pmc": None, "mfma": None}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc%s.json" % (PROFILE_ROUND, tag))))
        Wp = pmc.get("windows_per_dispatch", 4096)
        out["pmc"] = {k: v * W_run / Wp for k, v in pmc["hbm_bytes_per_dispatch"].items()}
        out["pmc_kernel_us"] = {k: v["avg_us"] for k, v in pmc["kernel_trace"].items()}
        out["pmc_kernels_sha16"] = pmc.get("kernels_sha16")
        out["pmc_commit"] = pmc.get("commit")
    except Exception:
        pass
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "%s_mfma%s.json" % (PROFILE_ROUND, tag))))
        util = {}
        for k, c in sq["kernels"].items():
            us = (out.get("pmc_kernel_us") or {}).get(k)
            if us and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                # busy cycles summed over the chip's 1024 SIMDs / (kernel cycles x 1024); GRBM_GUI_ACTIVE is summed over the 8 XCDs
                util[k] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        out["mfma"] = util
    except Exception:
        pass
    return out


def make_synth_window(cfg, n_landmarks, rate, seed):
    from cerberus_amd import synth
    prm = synth.default_params(n_landmarks=n_landmarks, seed=seed)
    prm.imu_rate_hz = float(rate)
    return synth.make_window(cfg, params=prm)


def cpu_info():
    """the host the CPU figures were timed on."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count()}


def reference_evaluate_timing(cfg, n_landmarks, rate, budget_s=4.0):
    """The COMPILED REFERENCE's own factor classes (oracle/_ref/libref.so: the Cerberus sources built against the Eigen / Ceres stand-ins)
    timed on one host core: one pass = Evaluate() with Jacobians of every cost function of a config-2 window, which is what one Ceres
    iteration asks of them at least once. Not a Ceres timing (no loss correction, elimination or dogleg in it) and not real Eigen — the part
    of the reference's per-iteration cost that CAN be run here, stated beside the "port" figure."""
    from oracle import oracle_py as O
    from oracle import ref_py as R
    if not R.available():
        return None
    ocfg = O.config_from(cfg)
    w = make_synth_window(cfg, n_landmarks, rate, 777000)
    O.fill_preint(ocfg, w)
    t1, nf = R.time_evaluate(ocfg, w, reps=2)
    reps = max(3, min(2000, int(budget_s / max(t1, 1e-6))))
    t, nf = R.time_evaluate(ocfg, w, reps=reps)
    return {"ms_per_evaluation_pass": 1e3 * t, "residual_blocks_per_pass": nf, "us_per_residual_block": 1e6 * t / max(nf, 1), "passes_timed": reps,
            "upper_bound_iters_per_s": 1.0 / t, "cores": 1, "kind": "reference",
            "what": "every cost function of one config-2 window (prior + 10 IMULegFactor + %d projection factors) Evaluate()d with Jacobians by the compiled "
                    "reference's classes; Ceres' own share of an iteration is not in it, Eigen is the build's stand-in" % (nf - 11)}


def cpu_baseline(cfg, n_landmarks, budget_s=15.0, rate=500, repropagate=False):
    """The oracle (CPU restatement of the reference path, scalar FP64, 1 thread) on windows of the same workload."""
    import contextlib
    from oracle import oracle_py as O
    ocfg = O.config_from(cfg)
    opts = O.default_opts(fixed_iterations=True, max_num_iterations=ITERS)
    opts.recompute_sqrt_info = 1
    t_total, n_it, n_win = 0.0, 0, 0
    while t_total < budget_s and n_win < 256:
        w = make_synth_window(cfg, n_landmarks, rate, 777000 + n_win)
        O.fill_preint(ocfg, w)
        with (O.repropagation(w) if repropagate else contextlib.nullcontext()):
            t0 = time.perf_counter()
            sm = O.solve_window(ocfg, w, opts)
            t_total += time.perf_counter() - t0
        n_it += sm.iterations
        n_win += 1
    return {"value": n_it / t_total, "unit": "GN iters/s", "cores": 1, "kind": "port", **cpu_info(),
            "sample": "%d synthetic %d-landmark %d Hz windows x %d iterations%s, oracle/liboracle.so (g++ -O3), 1 thread, %.1f s"
                      % (n_win, n_landmarks, rate, ITERS, ", every factor evaluation integrating its interval again" if repropagate else "", t_total)}


def marginalize_timing(ctx, cfg, windows, n_cpu=8):
    """The marginalisation half of Estimator::optimization() (estimator.cpp:1247-1455; once per frame, outside the iteration metric):
    GPU kernel time of vilo_marginalize over a batch (linearisation + MARGIN_OLD marginalisation, HIP events) beside the oracle's
    marginalize on the same windows. The reference builds the Hessian on 4 pthreads (marginalization_factor.cpp:246-275) and
    eigen-decomposes on one; the oracle is single-threaded throughout, which the `cores` field says."""
    import ctypes as C
    from cerberus_amd import api, synth, _ctypes as T
    from oracle import oracle_py as O
    W = len(windows)
    descs = (T.WindowDesc * W)(); states = (T.WindowState * W)(); priors = (T.Prior * W)()
    outs = [synth.PriorData() for _ in range(W)]
    for i, w in enumerate(windows):
        descs[i], states[i] = w.desc(T)
        priors[i] = outs[i].struct
    L = api.lib()
    L.vilo_last_marginalize_ms.restype = C.c_double
    for _ in range(2):
        rc = L.vilo_marginalize(ctx.h, W, descs, states, 0, priors)
    gpu_ms = L.vilo_last_marginalize_ms(ctx.h)
    ocfg = O.config_from(cfg)
    n_cpu = min(n_cpu, W)
    cpu = {}
    for nt in (1, 4):   # 4: the reference's NUM_THREADS for the Hessian build (marginalization_factor.h:22, .cpp:246-275); the eigen part is single-threaded there too
        O.lib().orc_set_marginalize_threads(nt)
        t0 = time.perf_counter()
        for w in windows[:n_cpu]:
            O.marginalize(ocfg, w, 0, synth.PriorData())
        cpu[nt] = 1e3 * (time.perf_counter() - t0) / n_cpu
    O.lib().orc_set_marginalize_threads(1)
    return {"mode": "MARGIN_OLD", "gpu_ms_per_window": gpu_ms / W, "gpu_batch": W, "gpu_ms_batch": gpu_ms, "cpu_ms_per_window": cpu[4], "cpu_cores": 4,
            "cpu_ms_per_window_1_thread": cpu[1],
            "cpu_sample": "%d windows, oracle/liboracle.so; Hessian build on 4 threads like the reference (cpu_ms_per_window) and on 1 (cpu_ms_per_window_1_thread), "
                          "Schur complement + eigen-decompositions on one thread in both" % n_cpu, "rc": int(rc)}


def _all_cores_child(k, n_landmarks, t_end, q):
    from cerberus_amd import synth
    from oracle import oracle_py as O
    cfg = synth.default_config()
    ocfg = O.config_from(cfg)
    opts = O.default_opts(fixed_iterations=True, max_num_iterations=ITERS)
    opts.recompute_sqrt_info = 1
    w = synth.make_window(cfg, n_landmarks=n_landmarks, seed=778000 + k % 16)
    O.fill_preint(ocfg, w)
    init = w.clone_state()
    n = 0
    t0 = time.time()
    while time.time() < t_end:
        w.set_state([a.copy() for a in init])
        n += O.solve_window(ocfg, w, opts).iterations
    q.put((n, t0, time.time()))


def cpu_all_cores_main(n_landmarks, budget_s):
    """Runs in a fresh interpreter (no torch / HIP state to fork): one process per core, one window each."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # a container's CPU quota, not the visible CPU count, is what "every core" means (cgroup v2 cpu.max = "quota period")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    t_end = time.time() + 3.0 + budget_s          # 3 s for every child to build its window
    ps = [ctx.Process(target=_all_cores_child, args=(k, n_landmarks, t_end, q)) for k in range(cores)]
    for p_ in ps:
        p_.start()
    res = [q.get() for _ in ps]
    for p_ in ps:
        p_.join()
    n_it = sum(r[0] for r in res)
    span = max(r[2] for r in res) - min(r[1] for r in res)
    print(json.dumps({"value": n_it / span, "unit": "GN iters/s", "cores": cores, "kind": "port",
                      "sample": "one config-2 window per process, %d processes, %d iterations in %.1f s wall, oracle/liboracle.so" % (cores, n_it, span)}))


def cpu_baseline_all_cores(n_landmarks, budget_s=6.0):
    """SURVEY §8(d) "all-cores mode": one window per core, every core of the box (own process each: the oracle's allocator
    traffic serialises threads of one process)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-all-cores", str(n_landmarks), str(budget_s)], capture_output=True, text=True,
                       timeout=120)
    return json.loads(r.stdout.strip().splitlines()[-1])


def sample_indices(W, per_place=3):
    """Which windows of a batch of W the parity sample looks at: the first, the middle and the LAST `per_place` — the tail of a launch is
    where a 32-bit stride, an arena-chunk boundary or a partial last round of workgroups would show."""
    idx = list(range(min(per_place, W))) + [W // 2 - 1 + k for k in range(per_place)] + [W - per_place + k for k in range(per_place)]
    return sorted({i for i in idx if 0 <= i < W})


def parity_sample(cfg, windows, ids, landmarks, rate, summ, idx=None, repropagate=False):
    """Checker leg (outside the timed region): the final states of the first, middle and last windows of the timed batch (`idx`,
    sample_indices) — as the timed kernels left them — against the oracle run on the same seeds for the same fixed number of iterations;
    and every window's final cost finite and within [0.1, 10] x the batch's median. The preintegration records are the ones the
    GPU integrated (K1 has its own goldens); everything after them is recomputed by the oracle."""
    import contextlib
    import numpy as np
    from oracle import oracle_py as O
    ocfg = O.config_from(cfg)
    opts = O.default_opts(fixed_iterations=True, max_num_iterations=ITERS)
    idx = sample_indices(len(windows)) if idx is None else [i for i in idx if i < len(windows)]
    max_state, max_cost = 0.0, 0.0
    for i in idx:
        w_ref = make_synth_window(cfg, landmarks, rate, 20260925 + ids[i])
        w_ref.preint[...] = windows[i].preint
        with (O.repropagation(w_ref) if repropagate else contextlib.nullcontext()):
            osum = O.solve_window(ocfg, w_ref, opts)
        for a, b in zip(windows[i].state_arrays(), w_ref.state_arrays()):
            max_state = max(max_state, float(np.abs(a - b).max() / max(1.0, np.abs(b).max())))
        max_cost = max(max_cost, abs(summ[i].final_cost - osum.final_cost) / max(abs(osum.final_cost), 1e-300))
    costs = np.array([s.final_cost for s in summ], dtype=np.float64)
    med = float(np.median(costs))
    outliers = [int(i) for i in np.nonzero(~(np.isfinite(costs) & (costs >= 0.1 * med) & (costs <= 10.0 * med)))[0]]
    states_finite = all(bool(np.isfinite(a).all()) for w in windows for a in w.state_arrays())
    return {"windows": len(idx), "window_indices": idx, "max_state_err": max_state, "max_cost_rel": max_cost, "tolerance": 1e-8,
            "all_windows": {"n": int(costs.size), "final_cost_median": med, "final_cost_min": float(np.nanmin(costs)), "final_cost_max": float(np.nanmax(costs)),
                            "outside_0.1x_10x_of_median": len(outliers), "first_outliers": outliers[:8], "states_finite": states_finite,
                            "ok": not outliers and states_finite},
            "what": "SOLVER parity (the oracle is handed the preintegration records the GPU integrated; K1 has its own goldens and "
                    "tests/test_gpu_parity.py::test_path_parity_with_the_oracles_own_preintegration covers the whole path): final states (relative to max(1, |state block|)) and final cost of the first, middle and last windows of the timed batch (window_indices) after the last timed step vs "
                    "oracle/liboracle.so on the same seeds, %d fixed iterations%s; all_windows: every window's downloaded final cost and states" % (ITERS, ", intervals integrated again per evaluation" if repropagate else "")}


def host_inclusive_block(ctx, cfg, lib, opts, landmarks, rate, W=4096, reps=3):
    """What a caller who hands over HOST windows gets (`value` of the line is measured with the batch resident): wall time of ONE
    vilo_solve_windows call on W windows in host memory, as a C++ caller pays it — the harness's descriptor building in Python is outside
    it. The call cuts itself into sub-batches over four internal lanes (vilo_set_host_pipeline): packing, PCIe and solver overlap. Beside
    it the same windows as one batch (vilo_batch_create + vilo_batch_solve + vilo_batch_download) with the library's own phase times."""
    import numpy as np
    from cerberus_amd import api
    from cerberus_amd import _ctypes as T
    ws = [make_synth_window(cfg, landmarks, rate, 70260925 + i) for i in range(W)]
    ctx.preintegrate_windows(ws)
    keep = [w.clone_state() for w in ws]
    lib.vilo_last_download_ms.restype = C.c_double
    descs, states = (T.WindowDesc * W)(), (T.WindowState * W)()
    for i, w in enumerate(ws):
        descs[i], states[i] = w.desc(T)

    def restore():
        for w, k in zip(ws, keep):
            w.set_state(k)
    one = None
    for _ in range(reps):
        restore()
        b = api.Batch(ctx, ws)                       # vilo_batch_create: pack + upload + sqrt_info
        cm = (C.c_double * 4)(); by = C.c_double(0.0)
        lib.vilo_last_create_ms(ctx.h, cm, C.byref(by))
        t0 = time.perf_counter()
        b.solve(opts)                                # (create has prepared sqrt_info once, like one vilo_solve_windows call)
        solve_ms = 1e3 * (time.perf_counter() - t0)
        b.download()
        dl_ms = float(lib.vilo_last_download_ms(ctx.h))
        b.close()
        tot = cm[0] + solve_ms + dl_ms
        r = {"ms": {"create": cm[0], "pack": cm[1], "alloc_and_upload": cm[2], "records_up_and_sqrt_info": cm[3], "solve": solve_ms, "download": dl_ms, "total": tot},
             "value": W * ITERS / (tot * 1e-3), "bytes_to_device": by.value,
             "h2d_gbps_over_upload_phases": by.value / max(1e-9, (cm[2] + cm[3]) * 1e-3) / 1e9}
        if one is None or r["value"] > one["value"]:
            one = r
    final_one = [np.concatenate([np.ravel(a) for a in w.state_arrays()]) for w in (ws[0], ws[W // 2], ws[-1])]
    best = None
    for _ in range(reps + 1):                        # (the first call creates the lanes and their staging)
        restore()
        t0 = time.perf_counter()
        summ = ctx.solve_window_descs(descs, states, opts)
        ms = 1e3 * (time.perf_counter() - t0)
        assert sum(s.iterations for s in summ) == W * ITERS
        best = ms if best is None else min(best, ms)
    same = all(np.array_equal(a, np.concatenate([np.ravel(x) for x in w.state_arrays()])) for a, w in zip(final_one, (ws[0], ws[W // 2], ws[-1])))
    # Estimator::optimization() as a whole on the same host windows: solve + gauge fix + marginalisation (MARGIN_OLD), the priors back in host memory
    from cerberus_amd.synth import PriorData
    outs = [PriorData() for _ in ws]
    priors, summ_o = (T.Prior * W)(), (T.SolveSummary * W)()
    for i, o in enumerate(outs):
        priors[i] = o.struct
    fl = (C.c_int * W)(*([0] * W))
    opt_ms = None
    for _ in range(2):
        restore()
        t0 = time.perf_counter()
        ctx._check(lib.vilo_optimize_windows(ctx.h, W, descs, states, C.byref(opts), fl, priors, summ_o))
        ms = 1e3 * (time.perf_counter() - t0)
        opt_ms = ms if opt_ms is None else min(opt_ms, ms)
    assert all(priors[i].valid == 1 and priors[i].n == 86 for i in (0, W // 2, W - 1))
    return {"windows": W, "value": W * ITERS / (best * 1e-3), "unit": "GN window-iterations/s", "ms": best,
            "h2d_gbps": one["bytes_to_device"] / (best * 1e-3) / 1e9, "bytes_to_device": one["bytes_to_device"],
            "bitwise_equal_to_one_batch": bool(same), "as_one_batch": one,
            "optimize_windows": {"ms": opt_ms, "windows_per_s": W / (opt_ms * 1e-3),
                                 "what": "ONE vilo_optimize_windows call on the same host windows: %d iterations, gauge fix, MARGIN_OLD marginalisation, the next priors (86 x 86) back in host memory" % ITERS},
            "what": ("ONE vilo_solve_windows call (%d fixed iterations) on %d config-2 windows handed over in host memory, wall time of the call (best of %d): "
                     "sub-batches of 1024 windows through four lanes of the context — host threads pack while the DMA engines and the solver work on "
                     "the lanes before; `as_one_batch`: the same windows as a single batch, phase by phase" % (ITERS, W, reps))}


def config3_block(ctx, cfg, lib, opts, W=256, steps=2):
    """BASELINE configs[2] as a side block of the default line (the driver only runs the default command): 1000 landmarks, 400 Hz,
    every iteration integrates all 10 intervals again. Same measurement as the headline: `steps` timed steps after one warm-up step that
    carries the per-kernel event pairs; value = window-iterations / wall time of the timed steps."""
    import numpy as np
    from cerberus_amd import api
    t0 = time.perf_counter()
    windows = [make_synth_window(cfg, 1000, 400, 40260925 + i) for i in range(W)]
    ctx.preintegrate_windows(windows)
    batch = api.Batch(ctx, windows)
    batch.set_samples()
    setup_s = time.perf_counter() - t0

    def table():
        ms = (C.c_double * 32)(); launches = (C.c_longlong * 32)()
        nk = lib.vilo_get_kernel_times(ctx.h, ms, launches, 32)
        lib.vilo_kernel_name.restype = C.c_char_p
        return {lib.vilo_kernel_name(i).decode(): {"ms_total": ms[i], "launches": int(launches[i]), "avg_ms": ms[i] / max(1, launches[i])} for i in range(nk)}
    lib.vilo_set_profiling(ctx.h, 1)   # (also clears the per-kernel table)
    batch.reset(); batch.solve(opts)
    kern = {k: v for k, v in table().items() if v["launches"]}
    lib.vilo_set_profiling(ctx.h, 0)
    import torch
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        batch.reset(); batch.solve(opts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    summ = batch.download()
    assert sum(s.iterations for s in summ) == W * ITERS
    n_samples = int(windows[0].sample_offsets[-1])
    b_alg = algorithmic_bytes(int(windows[0].n_obs), 1000) + 280 * n_samples
    dom = max(kern, key=lambda k: kern[k]["ms_total"])
    achieved = b_alg * W / (kern[dom]["avg_ms"] * 1e-3) / 1e9
    out = {"workload": "BASELINE configs[2]: 10-KF x 1000-landmark window, 400 Hz (%d samples per window), all 10 intervals integrated again in every "
                       "iteration + their sqrt_info; %d windows, %d timed steps of %d fixed iterations" % (n_samples, W, steps, ITERS),
           "value": W * ITERS * steps / dt, "unit": "GN window-iterations/s", "ms_per_step": 1e3 * dt / steps, "windows": W, "steps": steps,
           "kernels": kern, "setup_s": setup_s,
           "roofline": {"bound": "hbm", "kernel": dom, "kernel_avg_ms": kern[dom]["avg_ms"], "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                        "algorithmic_bytes_per_window_iteration": b_alg},
           # (parity_sample regenerates window i from seed 20260925 + ids[i]: these windows were drawn from 40260925 + i)
           "parity_sample": parity_sample(cfg, windows, [20000000 + i for i in range(W)], 1000, 400, summ, idx=[0, W - 1], repropagate=True)}
    batch.close()
    return out


def go1_config(cfg):
    """The reference's Go1 parameter set where it differs from the A1 one in what this path reads
    (config/go1_config/hardware_go1_vilo_config.yaml:25-30 against config/a1_config/hardware_a1_vilo_config.yaml): the foot force sensors
    decide contact (contact_sensor_type 2: the adaptive force model of imu_leg_integration_base.cpp:195-229); the leg geometry is the same
    (estimator.cpp:141-156, lower_leg_length 0.21 in both files). The calf lengths are estimated on line (BASELINE configs[4]: "online rho
    calibration" — optimize_leg_bias 1; the yaml file as shipped has it off)."""
    c = type(cfg).from_buffer_copy(cfg)
    c.contact_sensor_type = 2
    return c


def replay_block(device, n_images=170, cpu_budget_s=10.0, seed=505, keep_dir=None, prior_form="factor"):
    """BASELINE configs[4] / [0] (a bag replayed through the estimator, launch/dataset/run_campus_bag_vilo.launch:3-6, src/main.cpp:95-202) with
    the stand-in this container allows: a synthetic Go1-parameter stream (trot, 500 Hz IMU + joints + foot forces, 15 Hz stereo features) written
    as a ROS bag, read back through the bag reader (cerberus_amd/host/vilo_rosbag.cpp) and fed message by message through the node's logic
    (cerberus_amd/rosbag.py) into vilo::SlidingWindow on one GPU: per image the new interval's preintegration, the window's solve, the gauge fix and
    the marginalisation. Beside it the CPU oracle on the windows the replay dumped (solve on one thread + marginalisation with the Hessian build on
    four, marginalization_factor.h:22): the reported CPU baseline of this configuration."""
    import shutil
    import tempfile
    import numpy as np
    from cerberus_amd import api, rosbag, sequence, synth, window_io
    cfg5 = go1_config(synth.default_config())
    ctx5 = api.Context(cfg5, device=device)
    # nobody looks at the rows of the prior's J0 in a replay (it is only ever used through J0^T J0, J0^T r0, |r0|^2): the marginalisation may
    # leave the pivoted Cholesky factor where no eigenvalue would be dropped (include/vilo_gpu.h: vilo_set_prior_form) instead of sqrt(S) V^T
    ctx5.set_prior_form(prior_form)
    tmp = keep_dir or tempfile.mkdtemp(prefix="vilo_replay_")
    dumps = os.path.join(tmp, "windows")
    os.makedirs(dumps, exist_ok=True)
    mp = mp_raw = sw = None
    try:
        stream = sequence.Stream(cfg5, seed=seed)
        frames = [stream.next() for _ in range(n_images)]
        rng = np.random.default_rng(seed)
        for f in frames:   # contact flags of the synthetic gait -> foot forces in newtons, as the Go1's sensors report them
            f["forces"] = 15.0 + 140.0 * f["samples"][:, 31:35] + 4.0 * rng.normal(size=(len(f["samples"]), 4))
        bag = os.path.join(tmp, "go1_synthetic.bag")
        t_start = frames[0]["header"] - len(frames[0]["samples"]) / 500.0
        t0 = time.perf_counter()
        msgs = rosbag.write_stream_bag(bag, frames, t_start)
        write_s = time.perf_counter() - t0
        L = api.lib()
        L.vilo_last_solve_ms.restype = C.c_double; L.vilo_last_marginalize_ms.restype = C.c_double
        tr0 = frames[0]["truth"]

        def one_replay(dump_dir):
            """the bag through the estimator once; dump_dir: every solved window is written out (with its result) for the oracle, which also
            makes the prior and the preintegration records travel through host memory between images instead of staying on the device"""
            nonlocal mp, mp_raw, sw
            mp = mp_raw = sw = None
            sw = sequence.SlidingWindow(ctx5, cfg5, use_leg=1, optimize_leg_bias=1, dump_dir=dump_dir)
            sw.set_extrinsics(*stream.extrinsics())
            sw.init_first_pose(tr0[0:3], sequence.quat_to_R(tr0[3:7]).ravel(), tr0[7:10])   # (the synthetic robot is already walking at the first stamp)
            mp = sequence.MeasurementProcessor(sw)
            mp_raw = mp   # (busy_ms: the time inside the library's entry points — intake of every message, preintegration, solve, marginalisation, slide)
            est_s = [0.0]
            per_image = []

            def timed(fn):
                def wrapper(*a, **k):
                    t = time.perf_counter()
                    r = fn(*a, **k)
                    est_s[0] += time.perf_counter() - t
                    return r
                return wrapper
            mp.input_sample, mp.input_feature, mp.process = timed(mp.input_sample), timed(mp.input_feature), timed(mp.process)
            last = [0.0, 0.0]

            def on_image(k, t):
                st = sw.state()
                busy = mp_raw.busy_ms()
                per_image.append(dict(t=t, est_ms=busy - last[1], py_ms=1e3 * (est_s[0] - last[0]), solve_ms=float(L.vilo_last_solve_ms(ctx5.h)) if st["n_optimizations"] else 0.0,
                                      marg_ms=float(L.vilo_last_marginalize_ms(ctx5.h)) if st["n_optimizations"] else 0.0, n_opt=int(st["n_optimizations"]),
                                      rho=st["Rho"][api.T.F - 2].copy(), p=st["Ps"][api.T.F - 2].copy(), feats=int(st["feature_count"])))
                last[0], last[1] = est_s[0], busy
            t0 = time.perf_counter()
            cnt = rosbag.replay(rosbag.BagReader(bag), mp, contact_sensor_type=2, on_image=on_image)
            return per_image, cnt, time.perf_counter() - t0
        # (1) the product path, timed: prior and preintegration records device-resident between images (vilo_optimize_windows_resident);
        # (2) the same bag again with every window dumped for the oracle below — the prior and the records then travel through host
        #     memory between images (what a caller who keeps them on the host pays), timed beside it
        per_image, cnt, wall_s = one_replay(None)
        per_dump, cnt_dump, wall_dump = one_replay(dumps)
        same = float(max(np.abs(a["p"] - b["p"]).max() for a, b in zip(per_image, per_dump))) if len(per_image) == len(per_dump) else float("inf")
        steady = [r for r in per_image if r["n_opt"] > 12]     # the window is full and the prior has settled
        if not steady:
            return {"error": "the replay produced no steady images", "counters": cnt}
        est_ms = float(np.mean([r["est_ms"] for r in steady]))
        py_ms = float(np.mean([r["py_ms"] for r in steady]))
        solve_ms = float(np.mean([r["solve_ms"] for r in steady]))
        marg_ms = float(np.mean([r["marg_ms"] for r in steady]))
        truth = frames[-1]["truth"]
        rho_err = float(np.abs(per_image[-1]["rho"] - truth[16:20]).max())
        rho_err0 = float(np.abs(0.21 - truth[16:20]).max())
        pos_err = float(np.linalg.norm(per_image[-1]["p"] - truth[0:3]))
        # ---- CPU oracle on the dumped windows (checker leg / reported baseline: never part of the product path) ----
        from oracle import oracle_py as O
        ocfg = O.config_from(cfg5)
        files = sorted(os.listdir(dumps))
        O.lib().orc_set_marginalize_threads(4)
        cpu_solve = cpu_marg = 0.0
        n_cpu, worst = 0, 0.0
        for fn in files:
            if cpu_solve + cpu_marg > cpu_budget_s:
                break
            _, w, after, ref, flag = window_io.load(os.path.join(dumps, fn))
            before = w.clone_state()
            t1 = time.perf_counter()
            sm = O.solve_window(ocfg, w, O.default_opts(False, 12))
            cpu_solve += time.perf_counter() - t1
            O.gauge_fix(before, w)
            if after is not None:
                for a, b in zip(w.state_arrays(), after):
                    if a.size:
                        worst = max(worst, float(np.abs(a - b).max() / max(1.0, np.abs(b).max())))
                w.set_state(after)
            t1 = time.perf_counter()
            O.marginalize(ocfg, w, flag, synth.PriorData())
            cpu_marg += time.perf_counter() - t1
            n_cpu += 1
        O.lib().orc_set_marginalize_threads(1)
        return {
            "workload": "BASELINE configs[4] stand-in (no real bag in the container): synthetic Go1-parameter stream (contact_sensor_type 2: foot forces, "
                        "optimize_leg_bias 1: calf lengths estimated on line; 500 Hz IMU / joints / forces, 15 Hz stereo features), %d images written as a "
                        "rosbag-v2 file (%d messages, %d bytes), read back through the bag reader and replayed message by message through the node's "
                        "logic into the sliding-window estimator on one GPU" % (n_images, len(msgs), os.path.getsize(bag)),
            "value": 1e3 / est_ms, "unit": "images/s (estimator time per image = wall time inside the library's entry points, what a C++ node pays: every message in, preintegration push, batch build, solve, gauge fix, marginalisation, slide)",
            "images": len(per_image), "steady_images": len(steady), "prior_form": prior_form,
            "resident": "prior (vilo_prior_pool) and preintegration records (vilo_preint_streams) stay on the device between images in the timed replay",
            "host_carried_replay": {"ms_per_image_estimator": float(np.mean([r["est_ms"] for r in per_dump if r["n_opt"] > 12])) if any(r["n_opt"] > 12 for r in per_dump) else None,
                                    "what": "the same bag with prior and records carried through host memory and every window written to a file for the oracle check",
                                    "largest_position_difference_to_the_resident_replay_m": same},
            "ms_per_image": {"estimator": est_ms, "solve_gpu": solve_ms, "marginalise_gpu": marg_ms,
                             "preintegration_push_batch_build_and_host_bookkeeping": est_ms - solve_ms - marg_ms,
                             "estimator_through_the_python_wrappers": py_ms,
                             "whole_replay_wall_per_image_including_python_bag_reading": 1e3 * wall_s / max(1, len(per_image))},
            "camera_rate_hz": 15.0, "real_time_factor": (1e3 / est_ms) / 15.0,
            "rho_error_m": {"final": rho_err, "at_start": rho_err0}, "position_error_m_final": pos_err,
            "mean_features_in_window": float(np.mean([r["feats"] for r in steady])),
            "counters": cnt, "bag_write_s": write_s,
            "cpu_baseline": {"value": n_cpu / (cpu_solve + cpu_marg) if n_cpu else None, "unit": "images/s", "cores": 4, "kind": "port",
                             "sample": "the first %d dumped windows of this replay through oracle/liboracle.so: solve on 1 thread (%.1f ms per window), "
                                       "marginalisation with the Hessian build on 4 threads like the reference (%.1f ms per window)"
                                       % (n_cpu, 1e3 * cpu_solve / max(1, n_cpu), 1e3 * cpu_marg / max(1, n_cpu)),
                             "max_state_difference_gpu_vs_oracle": worst},
        }
    finally:
        mp = mp_raw = sw = None   # (the estimator objects go before their context)
        import gc
        gc.collect()
        ctx5.close()
        if not keep_dir:
            shutil.rmtree(tmp, ignore_errors=True)


class ControlPlane:
    """The two control collectives the bench contract asks for (barrier, max-over-ranks of the elapsed time) plus the shard report. The data
    path has no collective (windows are independent), so which library carries these three calls changes nothing that is measured."""

    def __init__(self, dist, group, backend, note):
        self.dist, self.group, self.backend, self.note = dist, group, backend, note

    def device(self, local_rank):
        return "cuda:%d" % local_rank if self.backend == "nccl" else "cpu"

    def barrier(self, local_rank):
        if self.dist is None:
            return
        if self.backend == "nccl":
            self.dist.barrier(group=self.group, device_ids=[local_rank])
        else:
            self.dist.barrier(group=self.group)

    def max_over_ranks(self, x, local_rank):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=self.device(local_rank))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def all_gather(self, vec, local_rank, world):
        import torch
        mine = torch.tensor(vec, dtype=torch.float64, device=self.device(local_rank))
        out = [torch.zeros(len(vec), dtype=torch.float64, device=mine.device) for _ in range(world)]
        self.dist.all_gather(out, mine, group=self.group)
        return [[float(v) for v in x.tolist()] for x in out]


def init_control_plane(rank, world, local_rank, want=None, probe_timeout_s=120.0):
    """One rank per GPU. The default process group is gloo (CPU tensors; it needs nothing from the GPUs and always comes up on 127.0.0.1).
    RCCL ("nccl" on ROCm) is then tried as a second group with one probe all-reduce on the rank's own GPU, in a thread with a deadline;
    the ranks agree over gloo on whether EVERY probe succeeded. If so the barrier and the max-over-ranks reduction run over RCCL, otherwise
    over gloo — a box where RCCL cannot start (two ranks on one GPU: "Duplicate GPU detected"; a missing IPC mode) still produces the line.
    VILO_BENCH_BACKEND=gloo skips the probe."""
    import datetime
    import threading
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    want = want or os.environ.get("VILO_BENCH_BACKEND", "nccl")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if want != "nccl":
        return ControlPlane(dist, None, "gloo", "VILO_BENCH_BACKEND=%s" % want)
    # Agree over gloo FIRST on whether a probe can be attempted at all: dist.new_group is a collective every rank must enter, so no rank
    # may go into it while a peer has already decided against (no GPU visible to it) — the others would sit in it until the deadline. The
    # same exchange finds two ranks on one device ("Duplicate GPU detected" inside RCCL otherwise, after a communicator is half built).
    have = 1 if torch.cuda.is_available() and torch.cuda.device_count() > 0 else 0
    mine = torch.tensor([have, local_rank if have else -1 - rank], dtype=torch.int64)
    every = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(every, mine)
    devs = [int(e[1]) for e in every]
    if not all(int(e[0]) for e in every):
        return ControlPlane(dist, None, "gloo", "RCCL unavailable, control collectives over gloo: a rank sees no GPU (agreed before any rank built a communicator)")
    if len(set(devs)) != world:
        return ControlPlane(dist, None, "gloo", "RCCL unavailable, control collectives over gloo: ranks share a device %r (RCCL refuses duplicate GPUs; agreed before any rank built a communicator)" % (devs,))
    g = dist.new_group(backend="nccl", timeout=datetime.timedelta(minutes=60))   # every rank enters: agreed above; lazy — no communicator yet
    res = {"ok": 0, "err": "probe did not return within %.0f s" % probe_timeout_s}

    def probe():   # the first collective builds the communicator: that is what can fail or hang, so it runs under a deadline
        try:
            t = torch.ones(1, dtype=torch.float64, device="cuda:%d" % local_rank)
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize()
            if int(t.item()) != world:
                raise RuntimeError("probe all-reduce returned %r, expected %d" % (t.item(), world))
            res.update(ok=1, err=None)
        except BaseException as e:   # (DistBackendError, RuntimeError, ... : anything means "not over RCCL on this box")
            res.update(ok=0, err=repr(e)[:300])
    th = threading.Thread(target=probe, daemon=True)
    th.start()
    th.join(probe_timeout_s)
    flag = torch.tensor([res["ok"] if not th.is_alive() else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # gloo, main thread, after the probe thread returned or was given up: every rank learns whether every probe succeeded
    if int(flag.item()) == 1:
        return ControlPlane(dist, g, "nccl", "RCCL probe all-reduce succeeded on every rank")
    return ControlPlane(dist, None, "gloo", "RCCL unavailable, control collectives over gloo: %s" % (res["err"] or "a peer's probe failed"))


def self_launch(n):
    """Re-run this command under torch.distributed.run with one rank per GPU of this node (what the driver does for N > 1)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: --gpus %d without a launcher: " % n + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5), help="2: BASELINE configs[1] (200 landmarks, 500 Hz; the headline metric); "
                    "3: BASELINE configs[2] (1000 landmarks, 400 Hz, K1 re-propagation inside every iteration); "
                    "5: BASELINE configs[4] stand-in: a synthetic Go1-parameter bag replayed through the sliding-window estimator (--images)")
    ap.add_argument("--images", type=int, default=170, help="--config 5 and the `replay` side block: images of the replayed stream")
    ap.add_argument("--no-replay", action="store_true", help="skip the `replay` side block of the default line")
    ap.add_argument("--windows", type=int, default=0, help="independent windows per GPU (default 32768; 1024 with --config 3)")
    ap.add_argument("--total-windows", type=int, default=0, help="BASELINE configs[3] mode: this many windows in total, window w on GPU w mod N (strong scaling)")
    ap.add_argument("--landmarks", type=int, default=0, help="default 200 (1000 with --config 3)")
    ap.add_argument("--rate", type=int, default=0, help="IMU / leg sample rate of the synthetic windows (Hz): default 500 (400 with --config 3)")
    ap.add_argument("--streams", type=int, default=2, help="resident batches solved concurrently in the multi-stream side figure (default 2: `two_streams`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the BASELINE configs[3] (1024 windows in total) side block")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE configs[2] side block of the default line")
    ap.add_argument("--single-window-latency", action="store_true", help="also time a batch of one window (default at N = 1)")
    ap.add_argument("--no-single-window", action="store_true", help="skip the one-window timing (rocprofv3 runs: keeps per-kernel averages pure)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1), so
        # that a bare invocation can never measure one GPU and label it N
        return self_launch(args.gpus)
    exit_code = 0
    if args.config == 5:
        # one robot's frames depend on each other through the prior: "replicas only" — one GPU, no sharding
        if args.gpus != 1:
            print("bench.py: --config 5 is a single stream on one GPU", file=sys.stderr)
            return 2
        r = replay_block(int(os.environ.get("LOCAL_RANK", "0")), n_images=args.images, cpu_budget_s=0.0 if args.no_cpu_baseline else 15.0)
        line = {"metric": "images/s, bag replay through the sliding-window VILO estimator (BASELINE configs[4] stand-in)", "value": r.get("value"),
                "unit": "images/s", "n_gpus": 1, "steps": r.get("steady_images"), "warmup": 12, "ms_per_step": (r.get("ms_per_image") or {}).get("estimator"),
                "higher_is_better": True, "scaling": "replicas only", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": r.get("workload")}, "roofline": None, "cpu_baseline": r.get("cpu_baseline"), "replay": r}
        print(json.dumps(line))
        return 0 if "error" not in r else 3
    rp = args.config == 3
    args.landmarks = args.landmarks or (1000 if rp else 200)
    args.rate = args.rate or (400 if rp else 500)
    args.windows = args.windows or (1024 if rp else DEFAULT_WINDOWS)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to label one as the other" % (args.gpus, world), file=sys.stderr)
        return 2
    import torch
    dist = None
    ctl = ControlPlane(None, None, "none", "single rank")
    if world > 1:
        import torch.distributed as dist
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        ctl = init_control_plane(rank, world, local_rank)
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)

    import numpy as np
    from cerberus_amd import api, synth
    cfg = synth.default_config()
    ctx = api.Context(cfg, device=local_rank)
    if args.total_windows > 0:
        # BASELINE configs[3] as stated: N_total independently seeded windows, window w -> GPU (w mod N)
        assert args.total_windows % world == 0, "--total-windows must be a multiple of the number of ranks"
        W = args.total_windows // world
        ids = [rank + world * i for i in range(W)]
        scaling = "strong"
        workload = "BASELINE configs[3] (%d windows in total, window w on GPU w mod %d)" % (args.total_windows, world)
    else:
        W = args.windows
        ids = [rank * W + i for i in range(W)]
        scaling = "weak"
        workload = "BASELINE configs[2]" if rp else "BASELINE configs[1]" if args.landmarks == 200 else "BASELINE configs[2]-sized"
    t0 = time.perf_counter()
    windows = [make_synth_window(cfg, args.landmarks, args.rate, 20260925 + g) for g in ids]
    ctx.preintegrate_windows(windows)   # K1 on the GPU: contact preintegration of all 10 * W intervals

    def make_batch(c, ws):
        b = api.Batch(c, ws)
        if rp:
            b.set_samples()   # samples resident: every iteration integrates the intervals again at the point it linearises
        return b
    batch = make_batch(ctx, windows)
    # the batch is resident: of the host copies only the first 256 windows are read again (parity sample, marginalisation timing, the
    # single window) — 0.5 MB per window that 8 ranks x 32 768 windows would otherwise hold on one host
    sample_idx = [0, W - 1] if rp else sample_indices(W)
    for i, w in enumerate(windows[256:], 256):
        rec = w.preint if i in sample_idx else None
        w.release_inputs()
        w.preint = rec
    setup_s = time.perf_counter() - t0
    opts = api.default_solve_opts(fixed_iterations=True, max_num_iterations=ITERS)
    lib = api.lib()

    def barrier():
        torch.cuda.synchronize()
        ctl.barrier(local_rank)
        torch.cuda.synchronize()

    # Warm-up steps run with an event pair around every kernel: the per-kernel table of the line comes from them. The timed steps
    # keep the pair around the dominant kernel only (what `roofline.achieved` is defined on): the ~80 other pairs cost the stream
    # 0.6 ms per step. Without warm-up steps the timed ones carry the full set.
    def kernel_table():
        ms = (C.c_double * 32)()
        launches = (C.c_longlong * 32)()
        nk = lib.vilo_get_kernel_times(ctx.h, ms, launches, 32)
        lib.vilo_kernel_name.restype = C.c_char_p
        return {lib.vilo_kernel_name(i).decode(): {"ms_total": ms[i], "launches": int(launches[i]), "avg_ms": ms[i] / max(1, launches[i])} for i in range(nk)}

    lib.vilo_set_profiling(ctx.h, 1)
    warm_counted = args.warmup
    for wi in range(args.warmup):
        batch.reset()
        if not rp:
            batch.prepare()
        batch.solve(opts)
        if wi == 0 and args.warmup >= 2:
            # a kernel's first launch in a process carries the loading of its code (milliseconds in front of k_init_state, the first
            # kernel of a solve): the table is taken over the warm-up steps behind the first
            lib.vilo_set_profiling(ctx.h, 1)
            warm_counted = args.warmup - 1
    kern_warm = kernel_table() if args.warmup > 0 else None
    dom_kind = None
    if kern_warm:
        names = list(kern_warm)
        dom_name = max((k for k in names if kern_warm[k]["launches"] > 0), key=lambda k: kern_warm[k]["ms_total"])
        dom_kind = names.index(dom_name)
    lib.vilo_set_profiling(ctx.h, 1 if dom_kind is None else 2 + dom_kind)
    barrier()
    t0 = time.perf_counter()
    gpu_ms = 0.0
    for _ in range(args.steps):
        batch.reset()
        if not rp:
            batch.prepare()           # sqrt_info of the preintegration records: the reference pays it in every IMULegFactor::Evaluate
        gpu_ms += batch.solve(opts)   # returns after the stream has drained (HIP event); config 3: re-propagation + sqrt_info inside, per iteration
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        elapsed = ctl.max_over_ranks(elapsed, local_rank)
        # which windows each rank solved (first / last seed and a checksum of its first window's observations): shows the shards are disjoint
        shards = ctl.all_gather([float(20260925 + ids[0]), float(20260925 + ids[-1]), float(windows[0].obs.sum())], local_rank, world)
        shard_info = [[int(x[0]), int(x[1]), float(x[2])] for x in shards]
    else:
        shard_info = [[20260925 + ids[0], 20260925 + ids[-1], float(windows[0].obs.sum())]]

    # per-kernel GPU time (HIP events on the solver's stream): the dominant kernel over the timed region, the others over the warm-up steps
    kern_timed = kernel_table()
    for v in kern_timed.values():
        v["steps"] = args.steps
    if kern_warm:
        for v in kern_warm.values():
            v["steps"] = warm_counted
        kern = dict(kern_warm)
        kern[dom_name] = kern_timed[dom_name]
    else:
        kern = kern_timed
    summ = batch.download()
    iters_done = sum(s.iterations for s in summ)
    assert iters_done == W * ITERS, "every window must run the fixed iteration count"
    final_cost = float(np.mean([s.final_cost for s in summ]))

    # BASELINE configs[3] as stated, beside the weak-scaling headline: 1024 windows IN TOTAL, window w on GPU w mod N (128 per GPU on 8). Every
    # rank solves its share with the same barrier / max-over-ranks timing as the headline; reported as a side block, so a 1 / 2 / 4 / 8-GPU
    # sweep of the default command shows the strong-scaling curve too.
    strong = None
    if args.total_windows == 0 and not rp and args.landmarks == 200 and not args.no_strong and STRONG_TOTAL % world == 0:
        Ws = STRONG_TOTAL // world
        wins_s = [make_synth_window(cfg, args.landmarks, args.rate, 50260925 + rank + world * i) for i in range(Ws)]
        ctx.preintegrate_windows(wins_s)
        bs = make_batch(ctx, wins_s)
        lib.vilo_set_profiling(ctx.h, 0)
        for _ in range(2):
            bs.reset(); bs.prepare(); bs.solve(opts)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            bs.reset(); bs.prepare(); bs.solve(opts)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        barrier()
        el = ctl.max_over_ranks(el, local_rank)
        summ_s = bs.download()
        assert sum(s_.iterations for s_ in summ_s) == Ws * ITERS
        bs.close()
        strong = {"workload": "BASELINE configs[3]: %d independent config-2 windows in total, window w on GPU w mod %d" % (STRONG_TOTAL, world),
                  "scaling": "strong", "total_windows": STRONG_TOTAL, "windows_per_gpu": Ws, "n_gpus": world, "steps": args.steps,
                  "value": STRONG_TOTAL * ITERS * args.steps / el, "value_per_gpu": STRONG_TOTAL * ITERS * args.steps / el / world,
                  "unit": "GN window-iterations/s", "ms_per_step": 1e3 * el / args.steps}
    # Small batches per GPU (BASELINE configs[3] on 8 GPUs is 128 windows per GPU; the reference itself solves one window per image): the
    # same timed loop on batches of 128 and 256 windows, reported as a side block of the default single-GPU line.
    small = None
    if world == 1 and args.total_windows == 0 and not rp and args.landmarks == 200 and not args.no_strong:
        small = {}
        lib.vilo_set_profiling(ctx.h, 0)
        for Wsm in (128, 256):
            wsm = [make_synth_window(cfg, args.landmarks, args.rate, 60260925 + i) for i in range(Wsm)]
            ctx.preintegrate_windows(wsm)
            bsm = make_batch(ctx, wsm)
            for _ in range(3):
                bsm.reset(); bsm.prepare(); bsm.solve(opts)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                bsm.reset(); bsm.prepare(); bsm.solve(opts)
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            assert sum(s_.iterations for s_ in bsm.download()) == Wsm * ITERS
            bsm.close()
            small[str(Wsm)] = {"windows": Wsm, "value": Wsm * ITERS * args.steps / el, "unit": "GN window-iterations/s", "ms_per_step": 1e3 * el / args.steps,
                               "us_per_iteration": 1e6 * el / args.steps / ITERS}
    host_inc = None
    if world == 1 and args.total_windows == 0 and not rp and args.landmarks == 200 and not args.no_strong:
        try:
            host_inc = host_inclusive_block(ctx, cfg, lib, opts, args.landmarks, args.rate)
        except Exception as e:   # (a side block: never the reason a line is lost)
            host_inc = {"error": repr(e)}
    if rank == 0:
        unit_work = world * W * ITERS * args.steps        # window-iterations of the whole job
        value = unit_work / elapsed
        sum_k = int(windows[0].n_obs)
        n_samples = int(windows[0].sample_offsets[-1])
        b_alg = algorithmic_bytes(sum_k, args.landmarks) + (280 * n_samples if rp else 0)
        dom = max((k for k in kern if kern[k]["launches"] > 0), key=lambda k: kern[k]["ms_total"])
        dom_avg_s = kern[dom]["avg_ms"] * 1e-3
        achieved = b_alg * W / dom_avg_s / 1e9             # one launch of the dominant kernel covers W window-iterations
        iter_ms = sum(v["ms_total"] / (v["steps"] * ITERS) for v in kern.values() if v["launches"] > 0)
        ev = profile_evidence(W, "_config3" if rp else "") if (rp or args.landmarks == 200) else {"pmc": None, "mfma": None}
        traffic = ev["pmc"].get(dom) if ev["pmc"] else None
        it_kernels = ITERATION_KERNELS + (REPROPAGATION_KERNELS if rp else ())
        it_traffic = sum(ev["pmc"].get(k, 0.0) for k in it_kernels) if ev["pmc"] else None
        alg_flops = ALG_FLOPS_PER_WINDOW_ITERATION.get(args.landmarks)
        if alg_flops and rp:
            alg_flops += REPROPAGATION_FLOPS
        tag = "_config3" if rp else ""
        sust = sustained_mfma_tflops()
        out = {
            "metric": "GN iters/sec, 10-KF x 200-landmark VILO window; 1/2/4/8-GPU batch throughput" if not rp else
                      "GN iters/sec, 10-KF x 1000-landmark VILO window + 400 Hz IMU preintegration re-propagated every iteration (BASELINE configs[2])",
            "value": value, "unit": "GN window-iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "value_per_gpu": value / world, "ranks_in_process_group": (dist.get_world_size() if dist is not None else 1),
            "control_backend": ctl.backend, "control_backend_note": ctl.note,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload + ": synthetic 10-KF x %d-landmark window, 4-leg contact preintegration (%d Hz), "
                                   "%d independent windows per GPU; one step = state reset + %s + %d fixed dogleg iterations%s"
                                   % (args.landmarks, args.rate, W, "nothing else" if rp else "sqrt_info of the %d preintegration records" % (10 * W), ITERS,
                                      ", each integrating all %d intervals (%d samples per window) again at the point it linearises and "
                                      "recomputing their sqrt_info" % (10 * W, n_samples) if rp else ""),
                       "windows_per_gpu": W, "total_windows": W * world, "iterations_per_step": ITERS, "observations_per_window": sum_k,
                       "parallelism": "independent windows sharded over ranks, no collective",
                       "shards": shard_info},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic,
                         "traffic_source": ({"file": "profiles/%s_pmc%s.json" % (PROFILE_ROUND, tag),
                                             "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on a 1 GiB copy; per-window figure scaled to this batch",
                                             "collected_at_commit": ev.get("pmc_commit"), "kernels_sha16_of_profile": ev.get("pmc_kernels_sha16"),
                                             "kernels_sha16_of_this_run": kernels_sha16(),
                                             "stale": ev.get("pmc_kernels_sha16") != kernels_sha16()} if traffic else None),
                         "kernel": dom, "kernel_avg_ms": kern[dom]["avg_ms"],
                         "rocprof_summary": "profiles/%s_rocprof_summary%s.txt (rocprofv3 --kernel-trace --stats of this command, tools/profile_gpu.sh)" % (PROFILE_ROUND, tag),
                         "algorithmic_bytes_per_window_iteration": b_alg,
                         "whole_iteration": {"gbps": b_alg * W / (iter_ms * 1e-3) / 1e9, "frac": b_alg * W / (iter_ms * 1e-3) / 1e9 / 8000.0,
                                             "traffic_bytes_per_window_iteration": it_traffic / W if it_traffic else None,
                                             "traffic_over_algorithmic": it_traffic / W / b_alg if it_traffic else None},
                         # the same iteration against the FP64 ceiling (78.6 TFLOP/s, vector and matrix rate alike on gfx950): SURVEY 8(d) prices
                         # the path in bytes, but at ~45 flop/B the FP64 units bound it
                         "fp64": {"bound": "mfma", "peak": 78.6, "unit": "TFLOP/s", "algorithmic_flops_per_window_iteration": alg_flops,
                                  "whole_iteration_tflops": alg_flops * W / (iter_ms * 1e-3) / 1e12 if alg_flops else None,
                                  "whole_iteration_frac": alg_flops * W / (iter_ms * 1e-3) / 1e12 / 78.6 if alg_flops else None,
                                  # what the matrix cores sustain by wall clock with every SIMD issuing FP64 MFMAs back to back (the shader clock
                                  # drops under that load): tools/micro/mfma_f64_rate.hip, profiles/round3_mfma_f64_rate.txt
                                  "sustained_mfma_tflops": sust[0], "sustained_mfma_tflops_one_wave_per_simd": sust[1],
                                  "sustained_mfma_source": "profiles/round3_mfma_f64_rate.txt (15 accumulators per wave: 16 / 4 waves per CU), parsed at run time",
                                  "whole_iteration_frac_of_sustained": alg_flops * W / (iter_ms * 1e-3) / 1e12 / sust[0] if (alg_flops and sust[0]) else None,
                                  "mfma_util": ev["mfma"],
                                  "mfma_util_source": "profiles/%s_mfma%s.json: SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), rocprofv3 --pmc" % (PROFILE_ROUND, tag) if ev["mfma"] else None},
                         # calibrated HBM bytes per window of every kernel of the iteration against the algorithmic bytes of the whole iteration
                         "traffic_per_kernel": ({k: {"bytes_per_window": ev["pmc"][k] / W, "over_algorithmic": ev["pmc"][k] / W / b_alg}
                                                 for k in it_kernels if k in ev["pmc"]} if ev["pmc"] else None)},
            "kernels": kern,
            "kernels_note": "HIP events on the solver's stream; `steps` = the steps an entry was measured over: the dominant kernel over the timed steps, "
                            "the others over the warm-up steps behind the first, which carries the kernels' code loading (the timed steps carry the dominant kernel's event pairs only)",
            "gpu_ms_per_step": gpu_ms / args.steps, "setup_s": setup_s, "mean_final_cost": final_cost,
        }
        if strong:
            out["strong_scaling"] = strong
        if small:
            out["small_batches"] = small
        if host_inc:
            out["host_inclusive"] = host_inc
        if not args.no_cpu_baseline:
            # checker leg, outside the timed region: did the timed kernels produce the reference's states?
            out["parity_sample"] = parity_sample(cfg, windows, ids, args.landmarks, args.rate, summ, idx=sample_idx, repropagate=rp)
        if (args.single_window_latency or world == 1) and not args.no_single_window:   # absolute rate of ONE window on one GPU
            b1 = make_batch(ctx, windows[:1])
            lib.vilo_set_profiling(ctx.h, 0)
            for _ in range(3):
                b1.reset(); b1.solve(opts)
            t1 = time.perf_counter()
            for _ in range(20):
                b1.reset(); b1.solve(opts)
            out["single_window_iters_per_s"] = 20 * ITERS / (time.perf_counter() - t1)
            b1.close()
        if world == 1 and not args.no_single_window:
            # S resident batches of the same size solved concurrently, each on its own context = HIP stream, one host thread each:
            # the small kernels and the tails of the large ones of one batch run in the gaps of the others. Reported beside `value`,
            # not as `value`: per-launch kernel durations (the roofline block above) are only meaningful without a co-runner.
            try:
                import threading
                lib.vilo_set_profiling(ctx.h, 0)
                S = max(2, args.streams)
                Ws = min(W, 16384)   # (windows per stream: bounded so that the side block's set-up stays under half a minute)
                ctxs, batches = [], []
                for k in range(S):
                    ck = api.Context(cfg, device=local_rank)
                    wk = [make_synth_window(cfg, args.landmarks, args.rate, 30260925 + 100000 * k + i) for i in range(Ws)]
                    ck.preintegrate_windows(wk)
                    ctxs.append(ck); batches.append(make_batch(ck, wk))
                    del wk

                def run(b, n):
                    for _ in range(n):
                        b.reset()
                        if not rp:
                            b.prepare()
                        b.solve(opts)
                for b in batches:
                    run(b, 2)
                t2 = time.perf_counter()
                th = [threading.Thread(target=run, args=(b, args.steps)) for b in batches]
                [t.start() for t in th]; [t.join() for t in th]
                dt2 = time.perf_counter() - t2
                out["two_streams" if S == 2 else "multi_stream"] = {
                    "value": S * Ws * ITERS * args.steps / dt2, "unit": "GN window-iterations/s", "streams": S, "windows_per_gpu": S * Ws,
                    "note": "%d batches of %d windows on %d HIP streams, same kernels; not the headline value" % (S, Ws, S)}
                for b, ck in zip(batches, ctxs):
                    b.close(); ck.close()
            except Exception as e:
                out["two_streams"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and not rp and not args.no_config3:
            try:
                out["config3"] = config3_block(ctx, cfg, lib, opts)
                c3 = cpu_baseline(cfg, 1000, budget_s=4.0, rate=400, repropagate=True)
                out["config3"]["cpu_baseline"] = c3
            except Exception as e:
                out["config3"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and not rp and not args.no_replay:
            try:
                out["replay"] = replay_block(local_rank, n_images=args.images)
            except Exception as e:
                out["replay"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["marginalize"] = marginalize_timing(ctx, cfg, windows[:256], n_cpu=2 if rp else 8)
            except Exception as e:
                out["marginalize"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            # rank 0, one thread, after the timed region (at N > 1 the other ranks are done with their GPUs by now and only wait for this
            # rank at the process group's teardown): every line the driver records carries the CPU figure beside `value`
            out["cpu_baseline"] = cpu_baseline(cfg, args.landmarks, budget_s=15.0 if world == 1 else 10.0, rate=args.rate, repropagate=rp)
            if world == 1 and not rp:
                try:
                    out["cpu_baseline"]["reference_evaluate"] = reference_evaluate_timing(cfg, args.landmarks, args.rate)
                except Exception as e:   # (the library is built from /root/reference where that exists and travels prebuilt)
                    out["cpu_baseline"]["reference_evaluate"] = {"error": repr(e)}
            if world == 1:
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.landmarks) if not rp else None
                except Exception as e:   # the single-thread number above is the contract; this one is extra information
                    out["cpu_baseline_all_cores"] = {"error": repr(e)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        bad = [k for k in ("parity_sample",) if isinstance(out.get(k), dict) and max(out[k]["max_state_err"], out[k]["max_cost_rel"]) > out[k]["tolerance"]]
        if isinstance(out.get("parity_sample"), dict) and not out["parity_sample"]["all_windows"]["ok"]:
            bad.append("parity_sample.all_windows")
        c3p = (out.get("config3") or {}).get("parity_sample") if isinstance(out.get("config3"), dict) else None
        if c3p and max(c3p["max_state_err"], c3p["max_cost_rel"]) > c3p["tolerance"]:
            bad.append("config3.parity_sample")
        if bad:
            print("bench.py: PARITY SAMPLE FAILED (%s): the timed kernels did not reproduce the oracle's states" % ", ".join(bad), file=sys.stderr)
            exit_code = 3
    batch.close()
    ctx.close()
    if dist is not None:
        sys.stdout.flush(); sys.stderr.flush()
        if ctl.backend == "gloo" and "unavailable" in ctl.note:
            # a failed RCCL probe may have left a communicator half-built (or its thread waiting for a peer): nothing to tear down cleanly
            try:
                dist.barrier()
            except Exception:
                pass
            os._exit(exit_code)
        dist.destroy_process_group()
    return exit_code


if __name__ == "__main__" and len(sys.argv) >= 2 and sys.argv[1] == "--cpu-all-cores":
    cpu_all_cores_main(int(sys.argv[2]), float(sys.argv[3]))
    sys.exit(0)

if __name__ == "__main__":
    sys.exit(main())
