#!/bin/bash
export OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 NUMEXPR_NUM_THREADS=1
for p in 6; do
(cd _r3 && python tools/cfg4_probe.py r3 $p 16 2>&1 | grep CFG4)
python tools/cfg4_probe.py r4 $p 16 2>&1 | grep CFG4
(cd _r3 && taskset -c 0-63 python tools/cfg4_probe.py r3pin $p 16 2>&1 | grep CFG4)
taskset -c 0-63 python tools/cfg4_probe.py r4pin $p 16 2>&1 | grep CFG4
done
