"""Replay a synthetic sensor stream frame by frame through the sliding-window manager on the GPU and print, per image, the
error of the newest pose / velocity against ground truth, the solver summary and the time per image.
    python tools/run_sequence.py [--images 60] [--robots 1] [--no-leg] [--dump DIR] [--csv FILE]
--csv writes robot 0's trajectory in the 20-column layout of the reference's VILO_RESULT_PATH file (src/main.cpp:156-196):
time [ns], robot position (3), velocity (3), six Kalman-filter columns (no KF here: zeros), the mocap position (here: the synthetic
ground truth), Rho1..Rho4 — so the reference's evaluation scripts read it unchanged."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberus_amd import api, sequence, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=60)
    ap.add_argument("--robots", type=int, default=1)
    ap.add_argument("--no-leg", action="store_true")
    ap.add_argument("--dump", default=None)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--csv", default=None)
    a = ap.parse_args()
    cfg = synth.default_config()
    ctx = api.Context(cfg, 0)
    streams = [sequence.Stream(cfg, seed=100 + r, t0=0.37 * r) for r in range(a.robots)]
    if a.dump:
        os.makedirs(a.dump, exist_ok=True)
    robots = [sequence.SlidingWindow(ctx, cfg, use_leg=0 if a.no_leg else 1, dump_dir=a.dump if r == 0 else None) for r in range(a.robots)]
    pool = api.PreintStreams(ctx, 11 * a.robots, imu_only=a.no_leg)
    priors = api.PriorPool(ctx, 2 * a.robots)
    for r, (s, w) in enumerate(zip(streams, robots)):
        w.set_extrinsics(*s.extrinsics())
        if pool:
            w.attach_streams(pool, 11 * r)
            w.attach_prior_pool(priors, 2 * r)
    t_img = []
    csv = open(a.csv, "w") if a.csv else None
    for k in range(a.images):
        frames = [s.next() for s in streams]
        for w, f in zip(robots, frames):
            sequence.feed(w, f, k == 0)
        t0 = time.perf_counter()
        sequence.process_images(ctx, robots, frames)
        t_img.append(time.perf_counter() - t0)
        st = robots[0].state()
        if csv and st["n_optimizations"] > 0:
            j, tr = api.T.F - 2, frames[0]["truth"]      # newest frame after the slide; p_br = 0, R_br = I (estimator.cpp:140-141)
            cols = ["%.0f" % (frames[0]["header"] * 1e9)] + ["%.5f" % v for v in list(st["Ps"][j]) + list(st["Vs"][j]) + [0.0] * 6 + list(tr[0:3]) + list(st["Rho"][j])]
            csv.write(",".join(cols) + ",\n")
        if st["n_optimizations"] == 0 or a.quiet:
            continue
        tr = frames[0]["truth"]
        j = api.T.F - 2   # newest frame after the slide
        sm = robots[0].summary()
        print(f"img {k:3d} flag {st['marginalization_flag']} feats {st['feature_count']:3d} prior_n {st['prior_n']:2d} it {sm.iterations:2d} "
              f"cost {sm.initial_cost:10.3f}->{sm.final_cost:10.3f}  |dp| {np.linalg.norm(st['Ps'][j] - tr[0:3]):.4f} m  "
              f"|dv| {np.linalg.norm(st['Vs'][j] - tr[7:10]):.4f}  |dba| {np.linalg.norm(st['Bas'][j] - tr[10:13]):.4f}  "
              f"|dbg| {np.linalg.norm(st['Bgs'][j] - tr[13:16]):.5f}  |drho| {np.linalg.norm(st['Rho'][j] - tr[16:20]):.5f}  {1e3 * t_img[-1]:.1f} ms")
    if csv:
        csv.close()
    steady = t_img[12:]
    if steady:
        print(f"{a.robots} robot(s): {1e3 * np.mean(steady):.2f} ms per image step (host bookkeeping + solve + marginalise), "
              f"{a.robots / np.mean(steady):.1f} robot-images/s")


if __name__ == "__main__":
    main()
