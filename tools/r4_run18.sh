#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "(test_full_size_matches_oracle or solve_properties) and (cfg5 or cfg3 or cfg2a)" 2>&1 | tail -3)
for c in 5 3 2a 2b 1; do
  HIPKKT_SPLIT_K=0 timeout 300 python tools/ab_variant.py $c base 6 2>&1 | grep "^AB"
  HIPKKT_VERBOSE=1 timeout 300 python tools/ab_variant.py $c split+pad 6 2>&1 | grep "^AB\|split-K"
done
