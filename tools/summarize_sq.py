"""Per-kernel means of every counter found in the rocprofv3 databases under <dir>/pass*/ (tools/profile_sq.sh)."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
KNOWN = ("k_repropagate", "k_visual_linearize", "k_visual_cost_walk", "k_visual_cost", "k_visual", "k_imu_linearize", "k_imu_raw", "k_imu_cost", "k_imu", "k_assemble_bias", "k_assemble_s", "k_assemble", "k_solve_wave", "k_solve_mid", "k_solve_mw8", "k_chain", "k_backsub", "k_accept",
         "k_init_state", "k_preint_imu_leg", "k_prepare_preint", "k_marginalize")


def short(name):
    for k in KNOWN:
        if k in name:
            return k
    return name[:40]


res = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "pass*"))):
    if not os.path.isdir(d):
        continue
    for dbf in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(dbf)
        try:
            rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
        except sqlite3.Error as e:
            print("skip", dbf, e)
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for name, cname, val in rows:
            a = acc[(short(name), cname)]
            a[0] += val
            a[1] += 1
        for (k, cname), (v, n) in acc.items():
            res[k][cname] = v / n
for k in sorted(res):
    c = res[k]
    print(k)
    for cname in sorted(c):
        print("   %-32s %18.1f" % (cname, c[cname]))
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for nm in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if nm in c:
                print("   %-32s %17.1f%% of SQ_WAVE_CYCLES" % (nm, 100.0 * c[nm] / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
        print("   MFMA busy / SQ busy cycles       %17.2f%%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]))
print(json.dumps(res))
