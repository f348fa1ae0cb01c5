#!/bin/bash
mkdir -p gpurun_out
run() { HIPKKT_X_MAIN=$1 HIPKKT_X_EXTRA=$2 timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r30.log 2>&1; echo main=$1 extra=$2 $(tail -1 gpurun_out/r30.log | cut -c80-110); }
run 0 1
run 0 11
run 0 2
run 0 0
run 2 1
run 2 2
run 2 0
run 0 12
run 0 1
