#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB-like units of the
TCC EA request counters (value * 1024 bytes); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x,
so the read side is shown raw and doubled.   usage: pmc_summary.py <fetch results.db> <write results.db>"""
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event_")][0]
    q = (f"select s.kernel_name, count(*), sum(p.value), avg(d.end - d.start) from {pe} p join {kd} d on p.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id group by s.kernel_name")
    return {n: (c, v, a) for n, c, v, a in cur.execute(q)}


def main(fetch_db, write_db):
    F, W = per_kernel(fetch_db), per_kernel(write_db)
    print(f"# fetch: {fetch_db}\n# write: {write_db}")
    print(f"{'kernel':<44} {'calls':>7} {'avg_us':>9} {'read_MB/launch':>15} {'read_x2_MB':>11} {'write_MB/launch':>16}")
    for n in sorted(F, key=lambda k: -F[k][1]):
        c, v, a = F[n]
        w = W.get(n, (1, 0, 0))
        name = n.replace("_ZN6hipkkt", "").replace(".kd", "")[:44]
        rd = v * 1024 / c / 1e6
        print(f"{name:<44} {c:>7d} {a / 1e3:>9.1f} {rd:>15.3f} {2 * rd:>11.3f} {w[1] * 1024 / max(1, w[0]) / 1e6:>16.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
