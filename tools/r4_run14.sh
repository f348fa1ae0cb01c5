#!/bin/bash
tag=${1:-r04q}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "(matches_oracle or solve_properties) and (cfg5 or cfg3)" 2>&1 | tail -3)
(timeout 600 python -m pytest tests/test_gpu_kkt.py -q -x 2>&1 | tail -2)
for c in 5 3 2a; do
  HIPKKT_SPLIT_K=0 timeout 300 python tools/ab_variant.py $c split0 6 2>&1 | grep "^AB"
  timeout 300 python tools/ab_variant.py $c split1 6 2>&1 | grep "^AB"
done
