#!/bin/bash
export HIPKKT_PLAN_CACHE=0
for nt in 0 8 16 64; do
for c in 5 2a 3 1; do HIPKKT_X_DENSE_NT=$nt python tools/ab_variant.py $c nt$nt 4 | grep "^AB"; done
done
