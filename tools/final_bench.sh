#!/bin/bash
# The two bench lines profiles/ keeps (default and --config 3), run after the profiles of the same kernel sources are committed so that
# roofline.traffic_source.stale is false: gpurun -- 'bash tools/final_bench.sh', then copy gpurun_out/final_bench/*.json to
# profiles/<round>_bench_final.json / <round>_bench_config3.json.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/final_bench; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python bench.py --config 3 --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['roofline']['traffic_source']['stale'], d['single_window_iters_per_s'], {k:v['value'] for k,v in d['small_batches'].items()}, d['config3']['value'])
c=json.load(open('$O/bench_c3.json')); print(c['value'], c['roofline']['traffic_source']['stale'], c['roofline']['kernel_avg_ms'])"
