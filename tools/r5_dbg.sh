#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0 16 18 24 26; do
echo "== HIPKKT_DEBUG_FLAGS=$f"; HIPKKT_DEBUG_FLAGS=$f timeout 100 python tools/fb2_trace.py 2>&1 | grep -A12 "batch 8" | grep -E "block|total" | cut -c1-120
done > gpurun_out/r5_dbg.txt 2>&1
cat gpurun_out/r5_dbg.txt
