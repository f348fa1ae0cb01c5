#!/bin/bash
# Round-6 developer loop for k_front_block2: parity of the two forms of the kernel + full-size oracle check on cfg 2a, stamps, bench line.
tag=${1:-r6fb}; tests=${2:-yes}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ "$tests" = "yes" ]; then
(timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "second_form or (full_size_matches_oracle and cfg2a) or (solve_properties and cfg2a)" > gpurun_out/${tag}_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest.txt); tail -2 gpurun_out/${tag}_pytest.txt
fi
timeout 300 python tools/fb2_trace.py > gpurun_out/${tag}_fb2trace.txt 2>&1; grep -A22 "batch 8" gpurun_out/${tag}_fb2trace.txt | grep -E "chain|first record|total|wg [0-4] start" | cut -c1-250; tail -1 gpurun_out/${tag}_fb2trace.txt
timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); fb = d['roofline']['kernels']['front_block']
print('bench', d['value'], 'factor', d['kkt_factor_ms'], 'fb us/launch', round(1e3 * fb['ms_per_refactor'] / 17, 1), 'frac', d['roofline']['frac'])"
