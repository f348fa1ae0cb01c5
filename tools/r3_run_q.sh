cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VARIANTS="default: off:HIPKKT_FB_EXTRA=0" SKIP_TESTS=1 bash tools/r3_ab.sh r03q 2a 3 1 2>&1 | grep -E "^AB|BENCH"
