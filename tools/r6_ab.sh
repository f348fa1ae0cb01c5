#!/bin/bash
# Round-6 developer loop: A/B of debug switches on the TESTING build (same sources as the product): refactorisation time of a config per setting.
# usage (via gpurun): bash tools/r6_ab.sh <tag> <cfg> "<ENV1=.. ENV2=..>" "<ENV..>" ...   (an empty string = defaults)
tag=$1; cfg=$2; shift 2
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export CLARABEL_HIPKKT_TESTING=1
for e in "$@"; do
  echo "== [$e]" | tee -a gpurun_out/${tag}_ab.txt
  env $e timeout 300 python tools/ab_variant.py $cfg "x" 6 2>&1 | tail -2 | tee -a gpurun_out/${tag}_ab.txt
done
