#!/bin/bash
# round 4, first GPU visit: streamed pivot chain -- parity (quick subset), A/B against the round-3 chain, wall-clock stamps, bench line
tag=${1:-r04a}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "streamed or (matches_oracle and cfg2a) or (matches_oracle and cfg3) or extra_tiles" > gpurun_out/pytest_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_$tag.log); tail -5 gpurun_out/pytest_$tag.log
(timeout 900 python -m pytest tests/test_gpu_kkt.py -q -x -k "developer_switches or timeout or front or hs_update or refine" > gpurun_out/pytest_kkt_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_kkt_$tag.log); tail -3 gpurun_out/pytest_kkt_$tag.log
for c in 2a 3; do
  HIPKKT_FB_STREAM=0 timeout 300 python tools/ab_variant.py $c stream0 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  timeout 300 python tools/ab_variant.py $c stream1 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  HIPKKT_FB_EXTRA=0 timeout 300 python tools/ab_variant.py $c stream1_noextra 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  HIPKKT_FB_EXTRA_PEN2=60 timeout 300 python tools/ab_variant.py $c stream1_pen2_60 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
  HIPKKT_FB_EXTRA_PEN2=1000 timeout 300 python tools/ab_variant.py $c stream1_pw1only 8 2>&1 | grep "^AB" >> gpurun_out/ab_$tag.txt
done
cat gpurun_out/ab_$tag.txt
timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_stream1_$tag.txt 2>&1; tail -30 gpurun_out/fbtrace_stream1_$tag.txt
HIPKKT_FB_STREAM=0 timeout 300 python tools/fb_trace.py > gpurun_out/fbtrace_stream0_$tag.txt 2>&1; grep "chain" gpurun_out/fbtrace_stream0_$tag.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-600
