#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/clarabel.jl_amd
for v in default ns; do
  lib=$L/libvariant_$v.so; [ $v = default ] && lib=$L/libclarabel_hipkkt.so
  CLARABEL_HIPKKT_LIB=$lib timeout 300 python tools/dump_stream.py dump_$v 2>&1 | tail -2
done
python - <<'PY'
import numpy as np
a = np.load("gpurun_out/dump_default.npz"); b = np.load("gpurun_out/dump_ns.npz")
sa, sb = a["stream"], b["stream"]
print("stream words", sa.size, "differ", int(np.sum(sa != sb)))
d = np.nonzero(sa != sb)[0]
if d.size:
    rec = d // 528
    print("first differing words", d[:10], "records (batch*40 + panel*8 + block)", np.unique(rec)[:20])
    q = d[0]
    print("first diff: batch", q // (5 * 8 * 528), "panel", (q // (8 * 528)) % 5, "block", (q // 528) % 8, "word", q % 528, sa[q:q+4].view(np.float64), sb[q:q+4].view(np.float64))
    w = d % 528
    print("word histogram of differing entries (kk = w // 64, col = w % 64):", np.unique(w // 64, return_counts=True))
da, db = a["D"], b["D"]
k = np.nonzero(da != db)[0]
print("D differ at", k.size, "first", k[:5], "N", da.size)
PY
