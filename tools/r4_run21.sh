#!/bin/bash
mkdir -p gpurun_out
(cd _r3 && timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > ../gpurun_out/r21_r3.log 2>&1); tail -1 gpurun_out/r21_r3.log | cut -c1-330
timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r21_r4.log 2>&1; tail -1 gpurun_out/r21_r4.log | cut -c1-330
(cd _r3 && timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > ../gpurun_out/r21_r3b.log 2>&1); tail -1 gpurun_out/r21_r3b.log | cut -c1-330
HIPKKT_FB_STREAM=0 timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r21_r4b.log 2>&1; tail -1 gpurun_out/r21_r4b.log | cut -c1-330
nproc; lscpu | grep -i "numa\|model name" | head
