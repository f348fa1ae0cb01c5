#!/bin/bash
export HIPKKT_PLAN_CACHE=0
for nd in 128 256 400; do
for c in 5 2a 3 1; do HIPKKT_X_THIN_ND=$nd python tools/ab_variant.py $c nd$nd 4 | grep "^AB" | cut -c1-90; done
done
