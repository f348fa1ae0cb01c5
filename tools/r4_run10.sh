#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/clarabel.jl_amd
for v in default v1 v2 v4; do
  lib=$L/libvariant_$v.so; [ $v = default ] && lib=$L/libclarabel_hipkkt.so
  echo "== $v"; CLARABEL_HIPKKT_LIB=$lib timeout 300 python tools/fb_trace.py > gpurun_out/fbt_$v.txt 2>&1; grep "factorisations ok\|factor ms" gpurun_out/fbt_$v.txt | cut -c1-120; grep -A3 "batch 8" gpurun_out/fbt_$v.txt | cut -c1-260; grep chain gpurun_out/fbt_$v.txt | tail -3
done
