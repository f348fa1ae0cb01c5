#!/bin/bash
# round 5: first run of front_block2.hip -- small-front parity tests, full-size comparison with the first form, stamps, bench
tag=${1:-r5b}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu > gpurun_out/${tag}_pytest_kkt.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_kkt.txt); tail -15 gpurun_out/${tag}_pytest_kkt.txt
(timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "second_form or streamed_pivot" > gpurun_out/${tag}_pytest_v2.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_v2.txt); tail -15 gpurun_out/${tag}_pytest_v2.txt
timeout 300 python tools/fb2_trace.py > gpurun_out/${tag}_fb2trace.txt 2>&1; tail -25 gpurun_out/${tag}_fb2trace.txt | cut -c1-260
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_2a.log 2>&1; tail -1 gpurun_out/${tag}_bench_2a.log | cut -c1-300
HIPKKT_FB_V2=0 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_2a_v1.log 2>&1; tail -1 gpurun_out/${tag}_bench_2a_v1.log | cut -c1-300
