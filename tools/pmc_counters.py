#!/usr/bin/env python
"""Per-kernel sums of every counter in a rocprofv3 --pmc results.db (any number of counters per pass), with the derived
matrix-core figures when the SQ counters are there:
    mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES-like denominator)   (both summed over the launch)
    f64_mfma_ops  = SQ_INSTS_VALU_MFMA_MOPS_F64 (x 512 flop-units, see --help of rocprofv3 -L)
usage: pmc_counters.py <results.db> [<results.db> ...]   (databases of separate passes are merged by kernel name)"""
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    q = ("select s.kernel_name, i.name, count(*), sum(p.value), avg(d.end - d.start) from rocpd_pmc_event p "
         "join rocpd_info_pmc i on p.pmc_id = i.id join rocpd_kernel_dispatch d on p.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, i.name")
    out = defaultdict(dict)
    for kn, cn, n, v, a in cur.execute(q):
        out[kn][cn] = (n, v, a)
    return out


def main(paths):
    merged = defaultdict(dict)
    for p in paths:
        for kn, d in per_kernel(p).items():
            merged[kn].update(d)
    names = sorted({c for d in merged.values() for c in d})
    print("# sources: " + " ".join(paths))
    print("# per-kernel SUM over the launches in the trace; calls and avg_us from the first counter's pass")
    hdr = f"{'kernel':<46} {'calls':>6} {'avg_us':>9}" + "".join(f" {c[:26]:>26}" for c in names)
    derived = []
    if "SQ_VALU_MFMA_BUSY_CYCLES" in names:
        for den in ("SQ_BUSY_CU_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if den in names:
                derived.append(("mfma_busy/" + den, "SQ_VALU_MFMA_BUSY_CYCLES", den))
    hdr += "".join(f" {d[0][:30]:>30}" for d in derived)
    print(hdr)
    order = sorted(merged, key=lambda k: -max((v[1] or 0) for v in merged[k].values()))
    for kn in order:
        d = merged[kn]
        first = next(iter(d.values()))
        name = kn.replace("_ZN6hipkkt", "").replace(".kd", "")[:46]
        line = f"{name:<46} {first[0]:>6d} {first[2] / 1e3:>9.1f}"
        for c in names:
            line += f" {d[c][1]:>26.6g}" if c in d else f" {'-':>26}"
        for _, num, den in derived:
            if num in d and den in d and d[den][1]:
                line += f" {d[num][1] / d[den][1]:>30.4f}"
            else:
                line += f" {'-':>30}"
        print(line)


if __name__ == "__main__":
    main(sys.argv[1:])
