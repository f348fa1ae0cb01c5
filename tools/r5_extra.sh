#!/bin/bash
tag=${1:-r5x}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { n=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); fb = d['roofline']['kernels']['front_block']; du = d['roofline']['kernels']['dense_update']
print('$n', d['value'], 'factor', d['kkt_factor_ms'], 'fb us/launch', round(1e3 * fb['ms_per_refactor'] / 17, 1), 'dense', du.get('ms_per_refactor'), du.get('launches_per_refactor'))
" >> gpurun_out/${tag}_extra.txt 2>&1; }
run default HIPKKT_VERBOSE=0
run noextra HIPKKT_FB_EXTRA=0
run pw1 HIPKKT_FB_EXTRA_PW=1,10,28
run pw2_pen20_50 HIPKKT_FB_EXTRA_PW=2,20,50
run pw2_pen30_70 HIPKKT_FB_EXTRA_PW=2,30,70
run pw2_pen5_15 HIPKKT_FB_EXTRA_PW=2,5,15
cat gpurun_out/${tag}_extra.txt
