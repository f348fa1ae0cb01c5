cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m "gpu and not slow" -k "extra_tiles or (matches_oracle and cfg2a)" 2>&1 | tail -2
VARIANTS="default: pw1:HIPKKT_FB_EXTRA_PER_WAVE=1 off:HIPKKT_FB_EXTRA=0" SKIP_TESTS=1 bash tools/r3_ab.sh r03q 2a 3 2>&1 | grep -E "^AB|BENCH"
