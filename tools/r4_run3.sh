#!/bin/bash
tag=${1:-r04c}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/clarabel.jl_amd
for v in default nop ns; do
  lib=$L/libvariant_$v.so; [ $v = default ] && lib=$L/libclarabel_hipkkt.so
  CLARABEL_HIPKKT_LIB=$lib timeout 300 python tools/chk_stream.py 2a $v 2>&1 | grep "^CHK\|rror" >> gpurun_out/chk_$tag.txt
  CLARABEL_HIPKKT_LIB=$lib timeout 300 python tools/ab_variant.py 2a $v 8 2>&1 | grep "^AB" >> gpurun_out/chk_$tag.txt
done
cat gpurun_out/chk_$tag.txt
for v in default nop ns; do
  lib=$L/libvariant_$v.so; [ $v = default ] && lib=$L/libclarabel_hipkkt.so
  CLARABEL_HIPKKT_LIB=$lib timeout 300 python tools/fb_trace.py 2>&1 | grep "chain" | sed "s/^/$v /"
done
CLARABEL_HIPKKT_LIB=$L/libclarabel_hipkkt.so timeout 300 python tools/fb_trace.py 2>&1 | grep -A8 "batch 8"
