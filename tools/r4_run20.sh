#!/bin/bash
mkdir -p gpurun_out
python tools/cfg4_probe.py default > gpurun_out/r20_a.log 2>&1
HIPKKT_FB_STREAM=0 python tools/cfg4_probe.py nostream > gpurun_out/r20_b.log 2>&1
HIPKKT_SPLIT_K=0 python tools/cfg4_probe.py nosplit > gpurun_out/r20_c.log 2>&1
HIPKKT_VERBOSE=1 python tools/cfg4_probe.py verbose 100 101 > gpurun_out/r20_d.log 2>&1
grep -h CFG4 gpurun_out/r20_a.log gpurun_out/r20_b.log gpurun_out/r20_c.log
tail -60 gpurun_out/r20_d.log
