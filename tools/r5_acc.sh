#!/bin/bash
tag=${1:-r5acc}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu -k "developer_switches" > gpurun_out/${tag}_pytest_sw.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_sw.txt); tail -3 gpurun_out/${tag}_pytest_sw.txt
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu -k "test_batch_config_matches_oracle and (324 or 179 or 208 or 100 or 150)" > gpurun_out/${tag}_pytest_324.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_324.txt); grep -E "batch-parity|passed|failed|rc=" gpurun_out/${tag}_pytest_324.txt | cut -c1-400
for c in 2a 3 5 1; do
timeout 400 python bench.py --config $c --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg $c', d['value'], 'factor', d['kkt_factor_ms'], 'solve', d['kkt_solve_ms_per_call'], d.get('refined_block_solves'))"
done 2>&1 | cut -c1-300
