#!/bin/bash
# round 5, first GPU call: box diagnosis, microbenchmarks, the suite the way the driver runs it (ONE process), baseline bench
tag=${1:-r5a}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{ rocm-smi --showclocks --showperflevel --showcomputepartition --showmemorypartition 2>&1 | head -60; } > gpurun_out/${tag}_rocm_smi.txt
python -c "
import clarabel_jl_amd
from clarabel_jl_amd import hipkkt
for i in range(3): print(hipkkt.box_probe(0))
" > gpurun_out/${tag}_box_probe.txt 2>&1
tail -1 gpurun_out/${tag}_box_probe.txt
./tools/bin/ubench_pivot > gpurun_out/${tag}_ubench_pivot.txt 2>&1; cat gpurun_out/${tag}_ubench_pivot.txt
./tools/bin/ubench_handoff > gpurun_out/${tag}_ubench_handoff.txt 2>&1; tail -1 gpurun_out/${tag}_ubench_handoff.txt
timeout 300 python tools/fb_trace.py > gpurun_out/${tag}_fbtrace.txt 2>&1; head -3 gpurun_out/${tag}_fbtrace.txt | cut -c1-200
timeout 600 python bench.py > gpurun_out/${tag}_bench_2a.log 2>&1; tail -1 gpurun_out/${tag}_bench_2a.log | cut -c1-400
(timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_pytest_gpu_single_process.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu_single_process.txt); tail -4 gpurun_out/${tag}_pytest_gpu_single_process.txt
{ rocm-smi --showclocks 2>&1 | head -40; } >> gpurun_out/${tag}_rocm_smi.txt
python -c "
import clarabel_jl_amd
from clarabel_jl_amd import hipkkt
print(hipkkt.box_probe(0))
" >> gpurun_out/${tag}_box_probe.txt 2>&1
