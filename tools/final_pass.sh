#!/bin/bash
# The end-of-round pass on the GPU box (gpurun -- 'bash tools/final_pass.sh'): the whole -m gpu suite, the default and config-3 bench
# lines, and the rocprofv3 passes whose summaries tools/refresh_profiles.py copies into profiles/ (kernel trace + FETCH_SIZE / WRITE_SIZE
# at the default line's 32768 windows per launch, the SQ counters, config 3 at 1024 windows, 128 windows, one window). ~13 minutes of box time.
#   afterwards, here:  PROF_WINDOWS=32768 python tools/refresh_profiles.py r6 r6sq gpurun_out/final/bench_default.json gpurun_out/final/bench_c3.json
#                      python tools/refresh_profiles.py --config3 r6c3
#                      (the w128 / w1 summaries: gpurun_out/prof_r6w128/summary.txt, prof_r6w1/summary.txt without their last line)
#   then commit and run tools/final_bench.sh for the bench lines whose counters belong to the committed kernels (traffic_source.stale false)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/final; mkdir -p $O
W=${PROF_WINDOWS:-32768}
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tee $O/pytest.log | tail -3
SECONDS=0
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? in $SECONDS s"
python bench.py --config 3 --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python bench.py --config 5 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
PROFILE_ARGS="--windows $W" bash tools/profile_gpu.sh r6 > gpurun_out/r6_prof.log 2>&1; tail -3 gpurun_out/r6_prof.log | cut -c1-300
bash tools/profile_sq.sh r6sq $W > gpurun_out/r6_sq.log 2>&1; grep -c "pass" gpurun_out/r6_sq.log
bash tools/profile_sq.sh r6sqc3 1024 "--config 3 --windows 1024" > gpurun_out/r6_sq_c3.log 2>&1; grep -c "pass" gpurun_out/r6_sq_c3.log
PROFILE_ARGS="--config 5 --images 60" bash tools/profile_gpu.sh r6replay > gpurun_out/r6_prof_replay.log 2>&1; tail -2 gpurun_out/r6_prof_replay.log | cut -c1-200
PROFILE_ARGS="--config 3 --windows 1024" bash tools/profile_gpu.sh r6c3 > gpurun_out/r6_prof_c3.log 2>&1; tail -2 gpurun_out/r6_prof_c3.log | cut -c1-200
PROFILE_ARGS="--windows 128" bash tools/profile_gpu.sh r6w128 > gpurun_out/r6_prof_w128.log 2>&1; tail -2 gpurun_out/r6_prof_w128.log | cut -c1-200
PROFILE_ARGS="--windows 1" bash tools/profile_gpu.sh r6w1 > gpurun_out/r6_prof_w1.log 2>&1; tail -2 gpurun_out/r6_prof_w1.log | cut -c1-200
echo total $SECONDS s
