#!/bin/bash
tag=${1:-r04r}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf gpurun_out/prof_ab5_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab5_$tag -o p -- python tools/ab_variant.py 5 prof 10 > gpurun_out/prof_ab5_$tag.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof_ab5_$tag/*results.db | head -1) > gpurun_out/prof_summary_ab5_$tag.txt 2>&1
head -22 gpurun_out/prof_summary_ab5_$tag.txt | cut -c1-150; grep "^AB" gpurun_out/prof_ab5_$tag.log
find gpurun_out -name "*.db" -size +30M -delete
