#!/usr/bin/env python
"""bench.py — Gauss-Newton (dogleg trust-region) iterations/s of the sliding-window VILO solve on MI355X.

Contract (see task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches
one rank per GPU through torch.distributed.run. A "step" is one pass of the hot path
(Estimator::optimization()'s solve half, estimator.cpp:1054-1245) over one batch of synthetic windows per GPU:
device-side state reset + 12 fixed trust-region iterations on every window. Windows are independent, so ranks
share nothing (weak scaling, no data-path collective); `value` = window-iterations of all ranks / max-over-ranks
time. Inputs are resident in HBM before the timed region (batch created + preintegrated during setup).

Workload = BASELINE.json configs[1]: synthetic 10-KF x 200-landmark window, A1 4-leg contact preintegration,
500 Hz IMU/leg samples; `--windows` independent instances per GPU (config 4 batches 1024 over 8 GPUs).
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


def algorithmic_flops_build_solve(L, F=11, n_prior=86):
    """FP64 flops of one k_build_solve launch per window: landmark Schur complement on the lower
    triangle (L rank-1 updates of the 80x80 block), the speed/leg-bias elimination (13x13 block chain, coupling rows,
    rank-143 update), the 80x80 Cholesky and the solves. 1 FMA = 2 flops."""
    tri = 80 * 81 // 2
    schur = 2 * L * tri + 2 * L * 80 * 3            # rank-1 updates + rhs / q / back-substitution products
    chain = F * (13 ** 3 // 3 + 3 * 13 ** 3)       # chol13 + three 13x13 triangular solves / products per frame
    coupling = F * 2 * (2 * 13 * 13 * 81)          # T(k) = M_k [B_k | rhs] - G_k T(k+1)
    rank = 2 * 13 * F * tri                        # C -= sum_k T_B^T T_B
    chol = 80 ** 3 // 3 + 2 * 80 * 80
    return schur + 2 * chain + coupling + rank + chol


def cpu_baseline(cfg, n_landmarks, budget_s=15.0):
    """The oracle (CPU restatement of the reference path, scalar FP64, 1 thread) on windows of the same workload."""
    import numpy as np  # noqa: F401
    from cerberus_amd import synth
    from oracle import oracle_py as O
    ocfg = O.config_from(cfg)
    opts = O.default_opts(fixed_iterations=True, max_num_iterations=ITERS)
    opts.recompute_sqrt_info = 1
    t_total, n_it, n_win = 0.0, 0, 0
    while t_total < budget_s and n_win < 256:
        w = synth.make_window(cfg, n_landmarks=n_landmarks, seed=777000 + n_win)
        O.fill_preint(ocfg, w)
        t0 = time.perf_counter()
        sm = O.solve_window(ocfg, w, opts)
        t_total += time.perf_counter() - t0
        n_it += sm.iterations
        n_win += 1
    return {"value": n_it / t_total, "unit": "GN iters/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic config-2 windows x %d iterations, oracle/liboracle.so (g++ -O3), 1 thread, %.1f s" % (n_win, ITERS, t_total)}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=4096, help="independent windows per GPU")
    ap.add_argument("--landmarks", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-window-latency", action="store_true", help="also time a batch of one window (default at N = 1)")
    ap.add_argument("--no-single-window", action="store_true", help="skip the one-window timing (rocprofv3 runs: keeps per-kernel averages pure)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; VILO_BENCH_BACKEND=gloo lets the multi-rank path be exercised on a single-GPU box
        dist.init_process_group(os.environ.get("VILO_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)

    import numpy as np
    from cerberus_amd import api, synth
    cfg = synth.default_config()
    ctx = api.Context(cfg, device=local_rank)
    W = args.windows
    t0 = time.perf_counter()
    windows = [synth.make_window(cfg, n_landmarks=args.landmarks, seed=20260925 + rank * W + i) for i in range(W)]
    ctx.preintegrate_windows(windows)   # K1 on the GPU: contact preintegration of all 10 * W intervals
    batch = api.Batch(ctx, windows)
    setup_s = time.perf_counter() - t0
    opts = api.default_solve_opts(fixed_iterations=True, max_num_iterations=ITERS)
    lib = api.lib()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.reset()
        batch.solve(opts)
    lib.vilo_set_profiling(ctx.h, 1)
    barrier()
    t0 = time.perf_counter()
    gpu_ms = 0.0
    for _ in range(args.steps):
        batch.reset()
        gpu_ms += batch.solve(opts)   # returns after the stream has drained (HIP event)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel GPU time over the timed region (HIP events on the solver's stream)
    ms = (C.c_double * 16)()
    launches = (C.c_longlong * 16)()
    nk = lib.vilo_get_kernel_times(ctx.h, ms, launches, 16)
    lib.vilo_kernel_name.restype = C.c_char_p
    kern = {lib.vilo_kernel_name(i).decode(): {"ms_total": ms[i], "launches": int(launches[i]),
                                               "avg_ms": ms[i] / max(1, launches[i])} for i in range(nk)}
    summ = batch.download()
    iters_done = sum(s.iterations for s in summ)
    assert iters_done == W * ITERS, "every window must run the fixed iteration count"
    final_cost = float(np.mean([s.final_cost for s in summ]))

    if rank == 0:
        unit_work = world * W * ITERS * args.steps        # window-iterations of the whole job
        value = unit_work / elapsed
        sum_k = int(windows[0].n_obs)
        b_alg = algorithmic_bytes(sum_k, args.landmarks)
        dom = max((k for k in kern if kern[k]["launches"] > 0), key=lambda k: kern[k]["ms_total"])
        dom_avg_s = kern[dom]["avg_ms"] * 1e-3
        achieved = b_alg * W / dom_avg_s / 1e9             # one launch of the dominant kernel covers W window-iterations
        iter_ms = sum(v["ms_total"] for v in kern.values()) / (args.steps * ITERS)
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile_gpu.sh: separate
        # FETCH_SIZE / WRITE_SIZE runs, calibrated on a known 1 GiB copy); only valid for the profiled configuration
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "round1_pmc_v8.json")   # profiled at 4096 windows per dispatch; traffic is per window
        if os.path.exists(pmc_file) and args.landmarks == 200:
            try:
                pmc = json.load(open(pmc_file))
                traffic = pmc["hbm_bytes_per_dispatch"].get(dom)
                if traffic is not None:
                    traffic = traffic * W / pmc.get("windows_per_dispatch", 4096)
            except Exception:
                traffic = None
        out = {
            "metric": "GN iters/sec, 10-KF x 200-landmark VILO window; 1/2/4/8-GPU batch throughput",
            "value": value, "unit": "GN window-iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]" if args.landmarks == 200 else "BASELINE configs[2]-sized") + ": synthetic 10-KF x %d-landmark window, 4-leg contact preintegration (500 Hz), "
                                   "%d independent windows per GPU, %d fixed dogleg iterations per step" % (args.landmarks, W, ITERS),
                       "windows_per_gpu": W, "iterations_per_step": ITERS, "observations_per_window": sum_k,
                       "parallelism": "independent windows sharded over ranks, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, calibrated; measured at 4096 windows per dispatch, scaled to this batch)" % os.path.basename(pmc_file)) if traffic else None, "kernel": dom, "kernel_avg_ms": kern[dom]["avg_ms"],
                         "rocprof_summary": "profiles/round1_rocprof_summary_v8.txt (rocprofv3 --kernel-trace --stats of this command, tools/profile_gpu.sh)",
                         "algorithmic_bytes_per_window_iteration": b_alg,
                         "whole_iteration_gbps": b_alg * W / (iter_ms * 1e-3) / 1e9,
                         # the same kernel against the FP64 matrix-core ceiling (78.6 TFLOP/s = half the 157.3 TF f32 MFMA rate
                         # of MI355X_MICROARCH.md): SURVEY 8(d) prices the path in bytes, but at ~40 flop/B the FP64 units bound it
                         "fp64": {"bound": "mfma", "achieved": algorithmic_flops_build_solve(args.landmarks) * W / dom_avg_s / 1e12 if dom == "k_build_solve" else None,
                                  "peak": 78.6, "unit": "TFLOP/s",
                                  "frac": algorithmic_flops_build_solve(args.landmarks) * W / dom_avg_s / 1e12 / 78.6 if dom == "k_build_solve" else None,
                                  "algorithmic_flops_per_window": algorithmic_flops_build_solve(args.landmarks),
                                  # 13.7 Mflop per window-iteration at config 2 for the WHOLE iteration (all kernels)
                                  "whole_iteration_tflops": (13.7e6 * W / (iter_ms * 1e-3) / 1e12) if args.landmarks == 200 else None,
                                  "whole_iteration_frac": (13.7e6 * W / (iter_ms * 1e-3) / 1e12 / 78.6) if args.landmarks == 200 else None}},
            "kernels": kern,
            "gpu_ms_per_step": gpu_ms / args.steps, "setup_s": setup_s, "mean_final_cost": final_cost,
        }
        if (args.single_window_latency or world == 1) and not args.no_single_window:   # absolute rate of ONE window on one GPU
            b1 = api.Batch(ctx, windows[:1])
            lib.vilo_set_profiling(ctx.h, 0)
            for _ in range(3):
                b1.reset(); b1.solve(opts)
            t1 = time.perf_counter()
            for _ in range(20):
                b1.reset(); b1.solve(opts)
            out["single_window_iters_per_s"] = 20 * ITERS / (time.perf_counter() - t1)
            b1.close()
        if world == 1 and not args.no_single_window:
            # Two resident batches of the same size solved concurrently, each on its own context = HIP stream, one host thread each:
            # the small kernels and the tails of the large ones of one batch run in the gaps of the other. Reported beside `value`,
            # not as `value`: per-launch kernel durations (the roofline block above) are only meaningful without a co-runner.
            try:
                import threading
                lib.vilo_set_profiling(ctx.h, 0)
                ctx2 = api.Context(cfg, device=local_rank)
                windows2 = [synth.make_window(cfg, n_landmarks=args.landmarks, seed=30260925 + i) for i in range(W)]
                ctx2.preintegrate_windows(windows2)
                batch2 = api.Batch(ctx2, windows2)

                def run(b, n):
                    for _ in range(n):
                        b.reset(); b.solve(opts)
                for b in (batch, batch2):
                    run(b, 2)
                t2 = time.perf_counter()
                th = [threading.Thread(target=run, args=(b, args.steps)) for b in (batch, batch2)]
                [t.start() for t in th]; [t.join() for t in th]
                dt2 = time.perf_counter() - t2
                out["two_streams"] = {"value": 2 * W * ITERS * args.steps / dt2, "unit": "GN window-iterations/s", "windows_per_gpu": 2 * W,
                                      "note": "two batches of %d windows on two HIP streams, same kernels; not the headline value" % W}
                batch2.close(); ctx2.close()
            except Exception as e:
                out["two_streams"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.landmarks)
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.landmarks)
            except Exception as e:   # the single-thread number above is the contract; this one is extra information
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    batch.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__" and len(sys.argv) >= 2 and sys.argv[1] == "--cpu-all-cores":
    cpu_all_cores_main(int(sys.argv[2]), float(sys.argv[3]))
    sys.exit(0)

if __name__ == "__main__":
    main()
