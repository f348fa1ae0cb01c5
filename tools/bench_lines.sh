#!/bin/bash
# The bench lines of every config (no tests, no profiler passes).  usage (via gpurun): bash tools/bench_lines.sh <tag>; then
# for c in 2a 2b 1 3 5 4: tail -1 gpurun_out/bench_${c}_<tag>.log > profiles/<tag>_bench_cfg$c.json
tag=${1:-lines}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/ab_variant.py 2a box_check 4 2>&1 | grep "^AB"
timeout 900 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-200
for c in 2b 1; do timeout 400 python bench.py --config $c > gpurun_out/bench_${c}_$tag.log 2>&1; tail -1 gpurun_out/bench_${c}_$tag.log | cut -c1-160; done
timeout 400 python bench.py --config 3 --device-scaling > gpurun_out/bench_3_$tag.log 2>&1; tail -1 gpurun_out/bench_3_$tag.log | cut -c1-160
timeout 400 python bench.py --config 5 --steps 10 --warmup 2 --device-scaling > gpurun_out/bench_5_$tag.log 2>&1; tail -1 gpurun_out/bench_5_$tag.log | cut -c1-160
timeout 900 python bench.py --config 4 --warmup 4 > gpurun_out/bench_4_$tag.log 2>&1; tail -1 gpurun_out/bench_4_$tag.log | cut -c1-300
