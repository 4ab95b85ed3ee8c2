#!/bin/bash
# A/B timing of library variants on the GPU box (through gpurun, from the repo root): tools/ab.sh <lib.so|default> ...   [AB_ARGS="..."]
# prints the bench value and the per-kernel table (HIP events) per variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for L in "$@"; do
  if [ "$L" = default ]; then unset VILO_GPU_LIB; else export VILO_GPU_LIB=$R/$L; fi
  echo "== $L ${AB_ENV:-}"
  env ${AB_ENV:-} timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-window --no-strong --no-config3 ${AB_ARGS:-} 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('value %.0f  ms_per_step %.3f' % (d['value'], d['ms_per_step']))
print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if v['launches']})"
done
