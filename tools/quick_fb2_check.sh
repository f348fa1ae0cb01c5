#!/bin/bash
tag=${1:-r5c3}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
./tools/bin/ubench_pivot 2>&1 | grep -i "reciprocal error" | tee gpurun_out/${tag}_rcpacc.txt
(timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu > gpurun_out/${tag}_pytest_kkt.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_kkt.txt); tail -2 gpurun_out/${tag}_pytest_kkt.txt
(timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "second_form" > gpurun_out/${tag}_pytest_v2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_v2.txt); tail -2 gpurun_out/${tag}_pytest_v2.txt
timeout 300 python tools/fb2_trace.py > gpurun_out/${tag}_fb2trace.txt 2>&1; grep -A20 "batch 8" gpurun_out/${tag}_fb2trace.txt | grep -E "chain|first record|total" | cut -c1-220; tail -1 gpurun_out/${tag}_fb2trace.txt
for e in "HIPKKT_VERBOSE=0" "HIPKKT_FB_EXTRA_PW=1,10,28"; do
env $e timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); fb = d['roofline']['kernels']['front_block']
print('$e', d['value'], 'factor', d['kkt_factor_ms'], 'fb us/launch', round(1e3 * fb['ms_per_refactor'] / 17, 1), 'frac', d['roofline']['frac'])"
done
