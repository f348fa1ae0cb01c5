#!/bin/bash
# One GPU round: parity tests, bench lines, microbench, rocprofv3 kernel trace.  Usage (via gpurun):
#   bash tools/gpu_round.sh <tag> [quick]
tag=${1:-x}; mode=${2:-full}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log)
tail -4 gpurun_out/pytest_gpu_$tag.log
if [ "$mode" = "full" ]; then
  timeout 600 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log
else
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log
fi
timeout 300 python bench.py --config 2b --no-cpu-baseline > gpurun_out/bench_2b_$tag.log 2>&1; tail -1 gpurun_out/bench_2b_$tag.log
if [ -x tools/ubench ]; then timeout 120 tools/ubench > gpurun_out/ubench_$tag.log 2>&1; cat gpurun_out/ubench_$tag.log; fi
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof_$tag/*results.db gpurun_out/prof_$tag/*/*results.db 2>/dev/null | head -1) > gpurun_out/prof_summary_$tag.txt 2>&1
head -12 gpurun_out/prof_summary_$tag.txt
# keep the merged output small
find gpurun_out/prof_$tag -name "*.db" -size +40M -delete
