#!/bin/bash
tag=${1:-r04f}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 2a 3; do timeout 300 python tools/chk_stream.py $c $tag 2>&1 | grep "^CHK\|rror" >> gpurun_out/chk_$tag.txt; done
for c in 2a 3; do
  HIPKKT_FB_STREAM=0 timeout 300 python tools/ab_variant.py $c stream0 8 2>&1 | grep "^AB" >> gpurun_out/chk_$tag.txt
  timeout 300 python tools/ab_variant.py $c stream1 8 2>&1 | grep "^AB" >> gpurun_out/chk_$tag.txt
  HIPKKT_FB_EXTRA_PEN2=1000 timeout 300 python tools/ab_variant.py $c stream1_pw1 8 2>&1 | grep "^AB" >> gpurun_out/chk_$tag.txt
done
cat gpurun_out/chk_$tag.txt
timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_$tag.txt 2>&1; grep -A8 "batch 8" gpurun_out/fbtrace_$tag.txt; grep chain gpurun_out/fbtrace_$tag.txt
