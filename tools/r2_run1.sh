#!/bin/bash
# round 2, GPU run 1: new parity tests + bench with the parity block + ticket A/B on cfg 2b
tag=${1:-r2a}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log)
tail -25 gpurun_out/pytest_gpu_$tag.log
timeout 600 python bench.py > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-1500
timeout 300 python bench.py --config 2b > gpurun_out/bench_2b_$tag.log 2>&1; tail -1 gpurun_out/bench_2b_$tag.log | cut -c1-1200
HIPKKT_SEG_TICKET=0 timeout 300 python bench.py --config 2b --no-cpu-baseline > gpurun_out/bench_2b_${tag}_noticket.log 2>&1; tail -1 gpurun_out/bench_2b_${tag}_noticket.log | cut -c1-400
