#!/bin/bash
# rocprofv3 SQ-counter passes for the solve loop (run on the GPU box through gpurun, from the repo root):
# wave occupancy / issue / wait breakdown, FP64-MFMA busy cycles, LDS bank conflicts, per kernel.
# Each pass is its own rocprofv3 run with --kernel-trace only (gpurun refuses PMC + other trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-sq}
WIN=${2:-4096}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# a third argument replaces the workload, e.g. "--config 3 --windows 1024" for BASELINE configs[2]
ARGS="--steps 1 --warmup 1 ${3:---windows $WIN} --no-cpu-baseline --no-single-window --no-strong --no-replay --no-config3"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
while read -r LINE; do
  [ -z "$LINE" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $LINE --kernel-trace -d $OUT/pass$i -o pmc -- python $R/bench.py $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i ($LINE): rc $?"
done <<PASSES
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS
PASSES
python $R/tools/summarize_sq.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
