#!/bin/bash
mkdir -p gpurun_out
for v in 2 1 0; do
HIPKKT_X_MAIN=$v timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r29_$v.log 2>&1; echo main=$v $(tail -1 gpurun_out/r29_$v.log | cut -c80-110)
done
HIPKKT_X_MAIN=2 python tools/ab_variant.py 2a main2 6 | grep "^AB"
HIPKKT_X_MAIN=0 python tools/ab_variant.py 2a main0 6 | grep "^AB"
HIPKKT_X_SIDE=1 python tools/ab_variant.py 2a side1 6 | grep "^AB"
