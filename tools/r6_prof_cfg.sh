#!/bin/bash
# Round-6 developer loop: kernel statistics of one bench config under rocprofv3.  usage (via gpurun): bash tools/r6_prof_cfg.sh <tag> <cfg> [pattern]
tag=$1; cfg=$2; pat=${3:-seg|front_|narrow|level}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o p -- python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
db=$(ls gpurun_out/prof_$tag/*results.db | head -1)
python tools/prof_summary.py $db > gpurun_out/${tag}_kernel_stats_$cfg.txt 2>&1
find gpurun_out -name "*.db" -delete
grep -E "$pat" gpurun_out/${tag}_kernel_stats_$cfg.txt | cut -c1-140
