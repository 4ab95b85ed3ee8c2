"""Condense the rocprofv3 (sqlite) output of tools/profile_gpu.sh into the numbers bench.py's roofline block cites."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
KNOWN = ("k_repropagate", "k_lin_small_c", "k_visual_linearize", "k_imu_linearize", "k_assemble_bias", "k_assemble_s", "k_assemble", "k_solve_wave", "k_solve_mid", "k_solve_mw8", "k_solve_mw", "k_chain", "k_backsub", "k_imu_raw", "k_visual_cost_walk", "k_visual_cost", "k_imu_cost", "k_accept", "k_init_state",
         "k_preint_imu_leg", "k_prepare_preint", "k_calib_copy", "k_marginalize")


def short(name):
    for k in KNOWN:
        if k in name:
            return k
    return name[:40]


def db(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


summary = {}
con = db("trace/**/*.db")
print("== rocprofv3 --kernel-trace --stats: python bench.py --steps 2 --warmup 1 (windows per GPU = grid / 64 of the k_solve_wave dispatch below)")
if con:
    ks = {}
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-22s calls %5d  avg %10.1f us  total %10.3f ms  %5.1f%%" % (short(name), calls, avg, total / 1e3, pct))
        ks[short(name)] = {"calls": calls, "avg_us": avg, "total_ms": total / 1e3}
    summary["kernel_trace"] = ks
    row = con.execute("select lds_size, vgpr_count, accum_vgpr_count, sgpr_count, workgroup_x, grid_x from kernels where name like '%k_solve_wave%' limit 1").fetchone()
    if row:
        print("k_solve_wave dispatch: lds %d B, vgpr %d, agpr %d, sgpr %d, workgroup %d, grid %d" % row)
cal, per = {}, {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    con = db("pmc_%s/**/*.db" % cname)
    print("== rocprofv3 --pmc %s --kernel-trace (counter unit: KiB per rocprofv3's definition)" % cname)
    if not con:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for name, val in con.execute("select kernel_name, value from counters_collection where counter_name = ?", (cname,)):
        k = short(name)
        acc[k][0] += val
        acc[k][1] += 1
    per[cname] = {k: v / n for k, (v, n) in acc.items()}
    for k, (v, n) in sorted(acc.items()):
        print("%-22s dispatches %5d  %s/dispatch %14.1f KiB" % (k, n, cname, v / n))
    if "k_calib_copy" in acc:
        v, n = acc["k_calib_copy"]
        cal[cname] = ((1 << 27) * 8.0 / 1024.0) / (v / n)   # true KiB per counted KiB (1 GiB streamed per dispatch, 8 B per lane)
        print("calibration (k_calib_copy, 1 GiB per dispatch): %.4f true bytes per counted byte" % cal[cname])
summary["calibration"] = cal
if cal:
    traffic = {}
    for k in set(list(per.get("FETCH_SIZE", {})) + list(per.get("WRITE_SIZE", {}))):
        traffic[k] = 1024.0 * (per.get("FETCH_SIZE", {}).get(k, 0.0) * cal.get("FETCH_SIZE", 1.0) + per.get("WRITE_SIZE", {}).get(k, 0.0) * cal.get("WRITE_SIZE", 1.0))
    summary["hbm_bytes_per_dispatch"] = traffic
    print("== calibrated HBM bytes per dispatch (read + write)")
    for k, v in sorted(traffic.items()):
        print("%-22s %14.0f B" % (k, v))
print(json.dumps(summary))
