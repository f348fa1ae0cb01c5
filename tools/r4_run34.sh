#!/bin/bash
export HIPKKT_PLAN_CACHE=0
for c in 5 2a 3 1 2b; do python tools/ab_variant.py $c thin 4 | grep "^AB"; done
timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r34.log 2>&1; echo cfg4 $(tail -1 gpurun_out/r34.log | cut -c80-110)
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "(test_full_size_matches_oracle and cfg5) or split_k or extra_tiles" > gpurun_out/r34_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r34_pytest.log); tail -4 gpurun_out/r34_pytest.log
