#!/bin/bash
tag=${1:-r04g}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "streamed or (matches_oracle and (cfg2a or cfg3 or cfg2b))" 2>&1 | tail -4)
for c in 2a 3; do
  HIPKKT_FB_STREAM=0 timeout 300 python tools/ab_variant.py $c stream0 8 2>&1 | grep "^AB"
  timeout 300 python tools/ab_variant.py $c stream1 8 2>&1 | grep "^AB"
done
timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_$tag.txt 2>&1; grep -B1 -A10 "wg 0 pivot loop" gpurun_out/fbtrace_$tag.txt | head -28; grep chain gpurun_out/fbtrace_$tag.txt
