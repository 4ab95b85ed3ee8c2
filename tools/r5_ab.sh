#!/bin/bash
# Round-5 A/B pass on the GPU box: [RUN_TESTS=1: the -m gpu suite with the default library first], then tools/ab.sh over the libraries
# named on the command line ("default" = cerberus_amd/lib/libvilo_gpu.so). Output under gpurun_out/r5_ab/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r5_ab; mkdir -p $O
if [ -n "${RUN_TESTS:-}" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:-} 2>&1 | tail -15 | tee $O/tests.txt
fi
bash tools/ab.sh "$@" 2>&1 | tee $O/ab.txt
