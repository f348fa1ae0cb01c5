#!/bin/bash
mkdir -p gpurun_out
(cd _r3 && HIPKKT_VERBOSE=1 timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > ../gpurun_out/r24_r3.log 2>&1)
HIPKKT_VERBOSE=1 timeout 300 python bench.py --config 4 --warmup 4 --no-cpu-baseline > gpurun_out/r24_r4.log 2>&1
for f in r3 r4; do echo $f; python - gpurun_out/r24_$f.log <<'PY'
import re,sys
t=open(sys.argv[1]).read()
sym=[float(x) for x in re.findall(r"symbolic ([0-9.]+) ms",t)]
dev=[float(x) for x in re.findall(r"device set-up ([0-9.]+) ms",t)]
ro=[float(x) for x in re.findall(r"runtime objects ([0-9.]+) ms",t)]
de=[float(x) for x in re.findall(r"destroy ([0-9.]+) ms",t)]
print("n",len(sym),"symbolic",sum(sym),"device set-up",sum(dev),"runtime objects",sum(ro),"destroy",sum(de))
print(t.strip().splitlines()[-1][80:110])
PY
done
grep -v "^hipkkt: N\|destroy\|front batch" gpurun_out/r24_r4.log | head -20
