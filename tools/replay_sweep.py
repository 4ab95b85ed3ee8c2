"""The bag replay of `bench.py --config 5` (synthetic Go1-parameter stream -> ROS bag -> bag reader -> MeasurementProcessor -> SlidingWindow on
the GPU; every solved window dumped and solved again by the oracle) over several stream seeds and both prior forms: how far the GPU's final
states are from the oracle's over a whole sequence, beyond the one seed the bench and the suite replay. Run on the GPU box:
python tools/replay_sweep.py [images] [seed ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seeds = [int(a) for a in sys.argv[2:]] or [505, 11, 2026, 77777, 31337]
    for form in ("factor", "eigen"):
        for seed in seeds:
            r = bench.replay_block(0, n_images=n, cpu_budget_s=120.0, seed=seed, prior_form=form)
            c = r["cpu_baseline"]
            print("seed %6d  prior form %-6s  %3d images  %.0f images/s (%.2f ms per image; host-carried %.2f)  oracle on the dumped windows: %s  largest state difference GPU vs oracle %.1e  "
                  "resident vs host-carried positions %.1e m  rho error %.4f -> %.4f m" % (
                      seed, form, r["images"], r["value"], r["ms_per_image"]["estimator"], r["host_carried_replay"]["ms_per_image_estimator"],
                      c["sample"].split(":")[0], c["max_state_difference_gpu_vs_oracle"], r["host_carried_replay"]["largest_position_difference_to_the_resident_replay_m"],
                      r["rho_error_m"]["at_start"], r["rho_error_m"]["final"]), flush=True)


if __name__ == "__main__":
    main()
