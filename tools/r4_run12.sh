#!/bin/bash
tag=${1:-r04n}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -s -k "twin_factorisation or (batch_config and 324)" > gpurun_out/pytest_new_$tag.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_new_$tag.log); grep -h "twin-parity\|batch-parity\|passed\|failed\|rc=\|SKIP\|skipped\|^E " gpurun_out/pytest_new_$tag.log | cut -c1-700
