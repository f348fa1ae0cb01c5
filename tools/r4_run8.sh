#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/clarabel.jl_amd
for v in default d1 d2 d4 d7; do
  lib=$L/libvariant_$v.so; [ $v = default ] && lib=$L/libclarabel_hipkkt.so
  echo "== $v"; CLARABEL_HIPKKT_LIB=$lib HIPKKT_FB_STREAM=0 timeout 300 python tools/fb_trace.py 2>&1 | grep -A10 "wg 0 pivot loop" | head -11
done
