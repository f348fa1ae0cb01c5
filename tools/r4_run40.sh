#!/bin/bash
mkdir -p gpurun_out
for c in 2a 3 1; do python tools/ab_variant.py $c ownfirst 6 | grep "^AB"; done
HIPKKT_FB_STREAM=0 python tools/ab_variant.py 2a nostream 6 | grep "^AB"
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "streamed_pivot or extra_tiles or (test_full_size_matches_oracle and cfg2a) or full_tile" > gpurun_out/r40_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r40_pytest.log); tail -4 gpurun_out/r40_pytest.log
timeout 300 python tools/fb_trace.py 2>&1 | head -12
