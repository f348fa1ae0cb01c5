#!/bin/bash
# The cause lines of the parity tests (printed with capsys disabled; pytest-xdist workers do not forward them, so this runs serially).
# usage (via gpurun): bash tools/parity_causes.sh <tag>
tag=${1:-final}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kkt.py -m gpu -q -s -k "(batch_config_matches_oracle and (179 or 208 or 324 or 126 or 100)) or twin_factorisation_matches or own_ordering or (full_size_ipm and cfg3)" > gpurun_out/parity_run_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/parity_run_$tag.log)
grep -h "batch-parity\|order-parity\|full-size-ipm\|twin-parity" gpurun_out/parity_run_$tag.log > gpurun_out/parity_causes_$tag.txt
tail -3 gpurun_out/parity_run_$tag.log; wc -l gpurun_out/parity_causes_$tag.txt
