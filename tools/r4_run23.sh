#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do
HIPKKT_X_SIDE=$v timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r23_$v.log 2>&1; echo side=$v $(tail -1 gpurun_out/r23_$v.log | cut -c80-110)
done
