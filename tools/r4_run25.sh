#!/bin/bash
mkdir -p gpurun_out
for p in 1 6; do
(cd _r3 && python tools/cfg4_probe.py r3 $p 8 2>&1 | grep CFG4)
python tools/cfg4_probe.py r4 $p 8 2>&1 | grep CFG4
done
