#!/bin/bash
# A/B timing of library variants on the GPU box (through gpurun, from the repo root): tools/ab.sh <lib.so|default>[@VAR=V[,VAR=V...]] ...   [AB_ARGS="..."]
# prints the bench value and the per-kernel table (HIP events) per variant; @VAR=V sets environment switches of the library for that run
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for SPEC in "$@"; do
  L=${SPEC%%@*}; E=""
  if [ "$L" != "$SPEC" ]; then E=$(echo "${SPEC#*@}" | tr ',' ' '); fi
  if [ "$L" = default ]; then unset VILO_GPU_LIB; else export VILO_GPU_LIB=$R/$L; fi
  echo "== $L $E ${AB_ENV:-}"
  env ${AB_ENV:-} $E timeout 200 python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --no-single-window --no-strong --no-config3 ${AB_ARGS:-} 2>gpurun_out/ab_last.err < /dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
    print('value %.0f  ms_per_step %.3f' % (d['value'], d['ms_per_step']))
    print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if v['launches']})
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/ab_last.err').read()[-1500:])"
done
