#!/bin/bash
mkdir -p gpurun_out
for d in _r3 _b_ab9ec64 _b_66d2be3 _b_94db3fa _b_02e1065; do
(cd $d && timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > ../gpurun_out/r22_$d.log 2>&1); echo $d $(tail -1 gpurun_out/r22_$d.log | cut -c80-110)
done
