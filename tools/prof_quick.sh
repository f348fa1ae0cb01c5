#!/bin/bash
# kernel-time summary of a short bench run (via gpurun): bash tools/prof_quick.sh <tag> [bench args]
tag=${1:-q}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
python tools/prof_summary.py $(ls gpurun_out/prof_$tag/*results.db gpurun_out/prof_$tag/*/*results.db 2>/dev/null | head -1) > gpurun_out/prof_summary_$tag.txt 2>&1
head -16 gpurun_out/prof_summary_$tag.txt
find gpurun_out/prof_$tag -name "*.db" -size +40M -delete
