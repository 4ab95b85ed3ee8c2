"""Copy the summaries of tools/profile_gpu.sh / tools/profile_sq.sh (run on the GPU box, merged back under gpurun_out/) into
profiles/<round>_* (bench.PROFILE_ROUND) — the files bench.py's roofline block and DESIGN.md cite.
Usage: python tools/refresh_profiles.py <prof_tag> <sq_tag> [bench.json] [bench_config3.json]
       python tools/refresh_profiles.py --config3 <prof_tag>     (PROFILE_ARGS="--config 3 --windows 1024" bash tools/profile_gpu.sh <prof_tag>)"""
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import subprocess  # noqa: E402

from bench import PROFILE_ROUND as RND, kernels_sha16  # noqa: E402


def _commit():
    try:
        return subprocess.check_output(["git", "-C", R, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        return None


def trace_and_pmc(tag, suffix, windows, what):
    lines = open(os.path.join(R, "gpurun_out", "prof_" + tag, "summary.txt")).read().rstrip("\n").split("\n")
    js = json.loads(lines[-1])
    js["windows_per_dispatch"] = windows
    # which kernels the counters belong to: bench.py compares this fingerprint with the tree it runs from (roofline.traffic_source.stale)
    js["kernels_sha16"] = kernels_sha16()
    js["commit"] = _commit()
    js["note"] = ("rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh (bench.py --steps 2 --warmup 1 %s), "
                  "calibrated on k_calib_copy; %s" % (what, RND))
    open(os.path.join(R, "profiles", "%s_rocprof_summary%s.txt" % (RND, suffix)), "w").write("\n".join(lines[:-1]) + "\n")
    json.dump(js, open(os.path.join(R, "profiles", "%s_pmc%s.json" % (RND, suffix)), "w"))
    return js


if sys.argv[1] == "--config3":
    js = trace_and_pmc(sys.argv[2], "_config3", 1024, "--config 3 --windows 1024")
    for k, v in sorted(js["kernel_trace"].items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
        print("%-24s %9.1f us  %8.1f KB/window" % (k, v["avg_us"], js["hbm_bytes_per_dispatch"].get(k, 0.0) / 1024 / 1e3))
    sys.exit(0)

prof, sq = sys.argv[1], sys.argv[2]
PROF_WINDOWS = int(os.environ.get("PROF_WINDOWS", "4096"))   # windows per dispatch of the profiled runs (tools/profile_gpu.sh / profile_sq.sh)
js = trace_and_pmc(prof, "", PROF_WINDOWS, "--windows %d" % PROF_WINDOWS)
lines = open(os.path.join(R, "gpurun_out", "prof_" + sq, "summary.txt")).read().rstrip("\n").split("\n")
res = json.loads(lines[-1])
open(os.path.join(R, "profiles", RND + "_sq_counters.txt"), "w").write("\n".join(lines[:-1]) + "\n")
json.dump({"note": "rocprofv3 --pmc SQ passes of tools/profile_sq.sh (bench.py --steps 1 --warmup 1 --windows %d): per-dispatch" % PROF_WINDOWS + " means summed over the "
                   "chip; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles; " + RND, "kernels_sha16": kernels_sha16(), "commit": _commit(),
           "windows_per_dispatch": PROF_WINDOWS, "kernels": res}, open(os.path.join(R, "profiles", RND + "_mfma.json"), "w"))
if len(sys.argv) > 3:
    shutil.copy(sys.argv[3], os.path.join(R, "profiles", RND + "_bench_final.json"))
if len(sys.argv) > 4:
    shutil.copy(sys.argv[4], os.path.join(R, "profiles", RND + "_bench_config3.json"))
it = ("k_visual_linearize", "k_imu_raw", "k_imu_linearize", "k_assemble", "k_assemble_bias", "k_chain", "k_solve_mid", "k_backsub", "k_solve_wave")
tot = 0.0
for k in it:
    b, us = js["hbm_bytes_per_dispatch"][k], js["kernel_trace"][k]["avg_us"]
    tot += b
    c = res.get(k, {})
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if "GRBM_GUI_ACTIVE" in c else float("nan")
    print("%-20s %8.1f us  %7.1f KB/window  %5.2f TB/s  mfma busy %4.1f %%  active %4.1f %%  wait %4.1f %%" %
          (k, us, b / PROF_WINDOWS / 1e3, b / us / 1e6, 100 * busy, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c.get("SQ_WAVE_CYCLES", 1), 100 * c.get("SQ_WAIT_ANY", 0) / c.get("SQ_WAVE_CYCLES", 1)))
print("iteration: %.1f KB per window-iteration = %.2f x 299 088 B; kernels %.1f us" % (tot / PROF_WINDOWS / 1e3, tot / PROF_WINDOWS / 299088, sum(js["kernel_trace"][k]["avg_us"] for k in it)))
