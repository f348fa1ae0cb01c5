#!/bin/bash
# rocprofv3 kernel trace of one bench config: bash tools/prof_cfg.sh <cfg> <tag> [extra bench args]
cfg=$1; tag=$2; shift 2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_${cfg}_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${cfg}_$tag -o p -- python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_${cfg}_$tag.log 2>&1
tail -1 gpurun_out/prof_${cfg}_$tag.log | cut -c1-300
python tools/prof_summary.py $(ls gpurun_out/prof_${cfg}_$tag/*results.db gpurun_out/prof_${cfg}_$tag/*/*results.db 2>/dev/null | head -1) > gpurun_out/prof_summary_${cfg}_$tag.txt 2>&1
head -40 gpurun_out/prof_summary_${cfg}_$tag.txt
find gpurun_out/prof_${cfg}_$tag -name "*.db" -size +40M -delete
