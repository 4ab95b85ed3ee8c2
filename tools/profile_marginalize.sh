#!/bin/bash
# rocprofv3 kernel trace of the marginalisation half (tools/time_marginalize.py, 256 windows, both flags); run through gpurun.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_marg
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/time_marginalize.py 256 > $OUT/run.log 2>&1
python - <<PY > $OUT/summary.txt
import glob, sqlite3
con = sqlite3.connect(glob.glob("$OUT/trace/**/*.db", recursive=True)[0])
print("== rocprofv3 --kernel-trace --stats: python tools/time_marginalize.py 256   (2 x MARGIN_OLD + 2 x MARGIN_SECOND_NEW calls of 256 windows)")
for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-60s calls %5d  avg %10.1f us  total %10.3f ms  %5.1f%%" % (name[:60], calls, avg, total / 1e3, pct))
row = con.execute("select lds_size, vgpr_count, accum_vgpr_count, sgpr_count, workgroup_x, grid_x from kernels where name like '%k_marginalize_lds%' limit 1").fetchone()
if row: print("k_marginalize_lds dispatch: lds %d B, vgpr %d, agpr %d, sgpr %d, workgroup %d, grid %d" % row)
PY
cat $OUT/run.log | tail -3
cat $OUT/summary.txt
