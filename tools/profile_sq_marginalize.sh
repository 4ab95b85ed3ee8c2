#!/bin/bash
# rocprofv3 SQ-counter passes for the marginalisation half (tools/time_marginalize.py 256; run through gpurun from the repo root):
# wave issue / wait breakdown, VALU / LDS / MFMA instruction counts of k_marginalize_lds. Each pass is its own run with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_sq_marg
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
while read -r LINE; do
  [ -z "$LINE" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $LINE --kernel-trace -d $OUT/pass$i -o pmc -- python $R/tools/time_marginalize.py 256 > $OUT/pass$i.log 2>&1
  echo "pass $i ($LINE): rc $?"
done <<PASSES
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS
PASSES
python $R/tools/summarize_sq.py $OUT > $OUT/summary.txt 2>&1
sed -n "/^k_marginalize/,/^k_[a-ln-z]/p" $OUT/summary.txt | head -40
