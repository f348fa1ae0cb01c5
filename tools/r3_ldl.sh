#!/bin/bash
# plain-LDL-solve timing of a config under variants (via gpurun): VARIANTS="label:ENV=v,..." bash tools/r3_ldl.sh <tag> <cfg...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; out=gpurun_out/ldl_$tag.txt; : > $out
for c in "$@"; do
  for v in $VARIANTS; do
    label=${v%%:*}; envs=${v#*:}
    env $(echo $envs | tr ',' ' ') timeout 300 python tools/ab_ldl.py $c $label 2>&1 | grep -E "^LDL|rror|timed out" >> $out
  done
done
cat $out
