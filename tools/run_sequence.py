"""Replay a synthetic sensor stream frame by frame through the sliding-window manager on the GPU and print, per image, the
error of the newest pose / velocity against ground truth, the solver summary and the time per image.
    python tools/run_sequence.py [--images 60] [--robots 1] [--no-leg] [--dump DIR] [--csv FILE]
    python tools/run_sequence.py --write-bag FILE [--images 60]     the synthetic stream as a ROS bag with the reference's topics
    python tools/run_sequence.py --bag FILE [--contact-sensor-type 1] [--csv FILE]     replay a bag (cerberus_amd/rosbag.py: the reference's node)
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
    ap.add_argument("--write-bag", default=None, help="write the synthetic stream of robot 0 (IMU, JointState, feature clouds) to this bag and stop")
    ap.add_argument("--bag", default=None, help="replay this bag instead of a synthetic stream (one robot)")
    ap.add_argument("--compression", default="none", choices=("none", "bz2", "lz4"), help="--write-bag: chunk compression (rosbag record --bz2 / --lz4)")
    ap.add_argument("--push-every", type=int, default=0, help="push the IMU / leg samples to the device-resident preintegration objects every N messages, "
                    "as they arrive, instead of in the image step (0: in the image step); same estimates, shorter image step")
    ap.add_argument("--prior-form", default="factor", choices=("factor", "eigen"), help="what a marginalisation leaves as the prior's J0: the certified pivoted "
                    "Cholesky factor (same information, a quarter of the time) or sqrt(S) V^T as the reference writes it (vilo_set_prior_form)")
    ap.add_argument("--contact-sensor-type", type=int, default=1, help="--bag: 0 / 1 the planner's flags (0: in place of the absent Kalman filter), 2 foot forces")
    a = ap.parse_args()
    cfg = synth.default_config()
    if a.write_bag:
        from cerberus_amd import rosbag
        stream = sequence.Stream(cfg, seed=100)
        frames = [stream.next() for _ in range(a.images)]
        msgs = rosbag.write_stream_bag(a.write_bag, frames, frames[0]["header"] - len(frames[0]["samples"]) / 500.0, compression=a.compression)
        print("%s: %d messages of %d images (%d bytes)" % (a.write_bag, len(msgs), a.images, os.path.getsize(a.write_bag)))
        return
    ctx = api.Context(cfg, 0)
    ctx.set_prior_form(a.prior_form)
    if a.bag:
        from cerberus_amd import rosbag
        sw = sequence.SlidingWindow(ctx, cfg, use_leg=0 if a.no_leg else 1)
        mp = sequence.MeasurementProcessor(sw)
        csv = open(a.csv, "w") if a.csv else None
        t_last = [time.perf_counter()]

        def on_image(k, t):
            st = sw.state()
            j = api.T.F - 2 if st["n_optimizations"] > 0 else max(st["frame_count"] - 1, 0)
            now = time.perf_counter()
            if not a.quiet:
                print("img %3d t %.6f frame_count %2d feats %3d solves %3d  p = %s  %.1f ms" % (k, t, st["frame_count"], st["feature_count"], st["n_optimizations"],
                                                                                             np.array2string(st["Ps"][j], precision=4), 1e3 * (now - t_last[0])))
            t_last[0] = now
            if csv and st["n_optimizations"] > 0:
                cols = ["%.0f" % (t * 1e9)] + ["%.5f" % v for v in list(st["Ps"][j]) + list(st["Vs"][j]) + [0.0] * 9 + list(st["Rho"][j])]
                csv.write(",".join(cols) + ",\n")

        cnt = rosbag.replay(rosbag.BagReader(a.bag), mp, contact_sensor_type=a.contact_sensor_type, on_image=on_image)
        if csv:
            csv.close()
        print("%s: %s" % (a.bag, cnt))
        return
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
        if a.push_every > 0:
            # the samples reach the device as they arrive (SlidingWindow::pushSamples every --push-every messages), not in the image step
            for w, f in zip(robots, frames):
                sequence.feed(w, dict(f, samples=f["samples"][:0]), k == 0)
            n_msg = max(len(f["samples"]) for f in frames)
            for m0 in range(0, n_msg, a.push_every):
                for w, f in zip(robots, frames):
                    part = f["samples"][m0:m0 + a.push_every]
                    if len(part):
                        w.process_samples(part)
                sequence.push_samples(ctx, robots)
        else:
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
