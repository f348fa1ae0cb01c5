#!/bin/bash
# Round-3 A/B run (via gpurun): GPU parity suite on the new defaults, then factor / solve device times of cfg 2a (and 3) with
# the round-3 switches off one at a time.   usage: bash tools/r3_ab.sh <tag> [cfgs...]
tag=${1:-r03b}; shift
cfgs=${@:-2a}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log)
tail -5 gpurun_out/pytest_gpu_$tag.log
out=gpurun_out/ab_$tag.txt; : > $out
for c in $cfgs; do
  HIPKKT_PIVOT_MODE=0 HIPKKT_XCD_ORDER=0 HIPKKT_FORK_GATHER=0 HIPKKT_SUPERHOP=0 timeout 300 python tools/ab_variant.py $c all_off 2>&1 | grep -E "^AB|rror|timed out" >> $out
  timeout 300 python tools/ab_variant.py $c all_on 2>&1 | grep -E "^AB|rror|timed out" >> $out
  HIPKKT_PIVOT_MODE=0 timeout 300 python tools/ab_variant.py $c pivot_off 2>&1 | grep -E "^AB|rror|timed out" >> $out
  HIPKKT_XCD_ORDER=0 timeout 300 python tools/ab_variant.py $c xcd_off 2>&1 | grep -E "^AB|rror|timed out" >> $out
  HIPKKT_FORK_GATHER=0 timeout 300 python tools/ab_variant.py $c forkgather_off 2>&1 | grep -E "^AB|rror|timed out" >> $out
  HIPKKT_SUPERHOP=0 timeout 300 python tools/ab_variant.py $c superhop_off 2>&1 | grep -E "^AB|rror|timed out" >> $out
done
cat $out
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/bench_2a_$tag.log 2>&1; tail -1 gpurun_out/bench_2a_$tag.log | cut -c1-900
