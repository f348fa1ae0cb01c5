#!/bin/bash
tag=${1:-r5p}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_2a_$tag
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_2a_$tag -o p -- python bench.py --config 2a --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_2a_$tag.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof_2a_$tag/*results.db | head -1) > gpurun_out/prof_summary_2a_$tag.txt 2>&1
head -40 gpurun_out/prof_summary_2a_$tag.txt
find gpurun_out -name "*.db" -delete
