#!/bin/bash
# mid-size batches (257 .. 1024 windows per GPU): which assembly / solver form — tools/r5_mid.sh on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for W in ${MID_WINDOWS:-320 512 768 1024}; do
  for V in "a:" "b:VILO_MW_MAX_WINDOWS=256 VILO_FUSE_ACCEPT_MAX_WINDOWS=256 VILO_ASM_FULL_MIN_WINDOWS=257" "c:VILO_MW_MAX_WINDOWS=256 VILO_FUSE_ACCEPT_MAX_WINDOWS=256 VILO_ASM_FULL_MIN_WINDOWS=257 VILO_SPLIT_MIN_WINDOWS=257"; do
    N=${V%%:*}; E=${V#*:}
    echo -n "W=$W variant $N: "
    env $E timeout 200 python bench.py --steps 20 --warmup 3 --windows $W --no-cpu-baseline --no-single-window --no-strong --no-config3 2>/dev/null < /dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
    print('value %.0f  ' % d['value'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if v['launches'] and k in ('k_assemble','k_assemble_bias','k_accept','k_solve_wave','k_chain','k_solve_mid','k_backsub')})
except Exception as e:
    print('FAILED', e)"
  done
done
