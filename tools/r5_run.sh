#!/bin/bash
# Round-5 GPU pass: [PHASE=1: phase clocks of the profiling build at 4096 windows] [RUN_TESTS=1: -m gpu suite] [MICRO=1: tools/micro checks] then tools/ab.sh over "$@"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r5_run; mkdir -p $O
if [ -n "${MICRO:-}" ]; then
  for x in _tmpbin/dpp_reduce_check; do [ -x $x ] && timeout 60 $x 2>&1 | tee -a $O/micro.txt; done
fi
if [ -n "${RUN_TESTS:-}" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:-} 2>&1 | tail -15 | tee $O/tests.txt
fi
if [ -n "${PHASE:-}" ]; then
  VILO_GPU_LIB=$R/cerberus_amd/lib/libvilo_gpu_prof.so timeout 300 python tools/phase_clocks_r3.py ${PHASE_W:-4096} 2>&1 | tee $O/phase.txt
fi
bash tools/ab.sh "$@" 2>&1 | tee $O/ab.txt
