#!/bin/bash
# Round-5 baseline pass on the GPU box (through gpurun, from the repo root): phase clocks of the full batch (profiling build) and the
# bench's value / kernel table for several windows-per-GPU and stream counts. Output: gpurun_out/r5_sweep/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r5_sweep; mkdir -p $O
if [ -f cerberus_amd/lib/libvilo_gpu_prof.so ]; then
  VILO_GPU_LIB=$R/cerberus_amd/lib/libvilo_gpu_prof.so timeout 300 python tools/phase_clocks_r3.py 4096 > $O/phase_4096.txt 2>&1
  cat $O/phase_4096.txt
fi
for W in ${SWEEP_WINDOWS:-4096 3072 6144 8192}; do
  echo "== windows $W"
  timeout 400 python bench.py --steps 10 --warmup 3 --windows $W --streams ${SWEEP_STREAMS:-2} --no-cpu-baseline --no-single-window --no-strong --no-config3 2>$O/bench_$W.err > $O/bench_$W.json
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$W.json').read().strip().split(chr(10))[-1])
    print('value %.0f  ms_per_step %.3f two_streams %s' % (d['value'], d['ms_per_step'], (d.get('two_streams') or {}).get('value')))
    print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if v['launches']})
except Exception as e:
    print('failed', e)
PY
done
