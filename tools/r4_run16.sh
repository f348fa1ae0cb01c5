#!/bin/bash
tag=${1:-r04s}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "(test_full_size_matches_oracle or solve_properties) and cfg5" 2>&1 | tail -3)
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "twin_factorisation" 2>&1 | tail -3)
for v in 0 0.25 0.1; do
  HIPKKT_PAD_HELD=$v timeout 300 python tools/ab_variant.py 5 pad$v 6 2>&1 | grep "^AB"
done
HIPKKT_SPLIT_K=0 timeout 300 python tools/ab_variant.py 5 pad0.25_nosplit 6 2>&1 | grep "^AB"
