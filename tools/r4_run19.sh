#!/bin/bash
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "split_k or (ipm_matches_oracle and cfg5)" > gpurun_out/r19_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r19_pytest.log); tail -40 gpurun_out/r19_pytest.log
timeout 600 python bench.py --config 4 --warmup 4 > gpurun_out/r19_bench4.log 2>&1; tail -1 gpurun_out/r19_bench4.log | cut -c1-600
