#!/bin/bash
# rocprofv3 evidence for bench.py's roofline block (run on the GPU box through gpurun, from the repo root):
#   pass 1: --kernel-trace --stats              -> per-kernel average duration (must agree with bench.py's HIP events)
#   pass 2: --pmc FETCH_SIZE  (+ kernel trace)  -> HBM read bytes per dispatch   (own pass: TCC has 4 slots, FETCH_SIZE takes 3)
#   pass 3: --pmc WRITE_SIZE  (+ kernel trace)  -> HBM write bytes per dispatch
# Each PMC pass also runs tools/calib_and_bench.py (k_calib_copy: 1 GiB read + 1 GiB written at 8 B per lane, 4 dispatches) to calibrate the counters (MI355X_MICROARCH.md §HBM).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# PROFILE_ARGS replaces the default workload, e.g. PROFILE_ARGS="--config 3 --windows 1024" for BASELINE configs[2]
ARGS="--steps 2 --warmup 1 ${PROFILE_ARGS:---windows 4096} --no-cpu-baseline --no-single-window --no-strong"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc -- python $R/tools/calib_and_bench.py $ARGS > $OUT/bench_pmc_$C.log 2>&1
done
python $R/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
