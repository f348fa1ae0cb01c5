#!/bin/bash
mkdir -p gpurun_out
for v in 1 0; do
HIPKKT_X_SIDE=$v timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r28_$v.log 2>&1; echo side=$v $(tail -1 gpurun_out/r28_$v.log | cut -c80-110)
done
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r28_q8.log 2>&1; echo q8 $(tail -1 gpurun_out/r28_q8.log | cut -c80-110)
GPU_MAX_HW_QUEUES=2 timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r28_q2.log 2>&1; echo q2 $(tail -1 gpurun_out/r28_q2.log | cut -c80-110)
