#!/bin/bash
mkdir -p gpurun_out
(cd _r3 && timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > ../gpurun_out/r27_r3.log 2>&1)
timeout 300 python bench_tt.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r27_r4.log 2>&1
for f in r3 r4; do python - gpurun_out/r27_$f.log <<'PY'
import re,sys
t=open(sys.argv[1]).read()
rows=re.findall(r"TT (\d+) gen ([0-9.]+) construct ([0-9.]+) solve ([0-9.]+) post ([0-9.]+) its (\d+)",t)
import numpy as np
a=np.array([[float(x) for x in r] for r in rows])
print(sys.argv[1], "n",len(a),"sum ms: gen %.0f construct %.0f solve %.0f post %.0f its %d"%tuple(a[:,1:].sum(0)), t.strip().splitlines()[-1][80:110])
PY
done
