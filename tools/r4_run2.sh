#!/bin/bash
tag=${1:-r04b}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "streamed or (matches_oracle and cfg2a)" > gpurun_out/pytest_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_$tag.log); tail -3 gpurun_out/pytest_$tag.log
for c in 2a 3; do
  HIPKKT_FB_STREAM=0 timeout 300 python tools/ab_variant.py $c stream0 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  timeout 300 python tools/ab_variant.py $c stream1 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  HIPKKT_FB_EXTRA=0 timeout 300 python tools/ab_variant.py $c stream1_noextra 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  HIPKKT_FB_STREAM=0 HIPKKT_FB_EXTRA=0 timeout 300 python tools/ab_variant.py $c stream0_noextra 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  HIPKKT_FB_EXTRA_PEN2=1000 timeout 300 python tools/ab_variant.py $c stream1_pw1only 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
done
cat gpurun_out/ab_$tag.txt
timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_stream1_$tag.txt 2>&1; grep -A8 "batch 8" gpurun_out/fbtrace_stream1_$tag.txt; grep chain gpurun_out/fbtrace_stream1_$tag.txt
