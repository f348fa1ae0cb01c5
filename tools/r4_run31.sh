#!/bin/bash
mkdir -p gpurun_out
run() { timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline --workers $1 --in-flight $2 > gpurun_out/r31.log 2>&1; echo workers=$1 inflight=$2 $(tail -1 gpurun_out/r31.log | cut -c80-110); }
run 6 1
run 8 1
run 10 1
run 12 1
run 16 1
run 4 2
run 6 2
run 8 2
