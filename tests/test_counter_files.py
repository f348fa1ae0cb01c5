"""The hardware-counter figures a bench line quotes (`roofline.traffic`, `roofline.whole_refactor`) come from files that
tools/pmc_to_json.py writes out of SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md: one counter set per pass).  These CPU tests
hold that tool against a hand-made database of the tables it queries, and the committed counter files against what bench.py reads."""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, counters, rows, instances=2):
    """rows = [(kernel name, value per instance)] in dispatch order; every dispatch reports `instances` samples per counter"""
    db = sqlite3.connect(path)
    db.executescript("create table rocpd_info_pmc(id integer, name text); create table rocpd_pmc_event(pmc_id integer, event_id integer, value real);"
                     "create table rocpd_kernel_dispatch(event_id integer, kernel_id integer); create table rocpd_info_kernel_symbol(id integer, kernel_name text);")
    kid = {n: i + 1 for i, n in enumerate(sorted({r[0] for r in rows}))}
    for n, i in kid.items():
        db.execute("insert into rocpd_info_kernel_symbol values(?,?)", (i, n))
    for ci, c in enumerate(counters):
        db.execute("insert into rocpd_info_pmc values(?,?)", (ci + 1, c))
    for ev, (n, val) in enumerate(rows):
        db.execute("insert into rocpd_kernel_dispatch values(?,?)", (ev, kid[n]))
        for ci in range(len(counters)):
            for _ in range(instances):
                db.execute("insert into rocpd_pmc_event values(?,?,?)", (ci + 1, ev, val))
    db.commit()
    db.close()


def test_counter_tool_separates_launch_family_refactorisation_and_trace(tmp_path):
    one = [("_ZN7hipkkt13k_init_panelsEPd", 10), ("_ZN7hipkkt14k_update_denseILi4ELi4EEEvNS_7DevPlanEii", 1000),
           ("_ZN7hipkkt14k_factor_panelENS_7DevPlanEidd", 100), ("_ZN7hipkkt9k_fwd_segENS_7DevPlanE", 50)]
    rows = one + one + one
    f, w, s, g = (str(tmp_path / n) for n in ("f.db", "w.db", "s.db", "g.db"))
    _db(f, ["FETCH_SIZE"], rows)
    _db(w, ["WRITE_SIZE"], rows)
    _db(s, ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CU_CYCLES"], rows)
    _db(g, ["GRBM_GUI_ACTIVE"], rows)
    out = str(tmp_path / "c.json")
    env = dict(os.environ, PMC_COMMAND="a test")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), "2a", out, f, w, s, g], check=True, env=env, capture_output=True)
    d = json.load(open(out))
    kb = 1024 * 2                                    # two instances per dispatch, counters in KiB
    assert d["launches_in_trace"] == 3
    assert d["bytes_per_launch"] == 2 * 1000 * kb + 1000 * kb            # FETCH_SIZE counts half the bytes on gfx950: doubled
    wr = d["whole_refactor"]
    assert wr["refactorisations_in_trace"] == 3 and wr["dispatches_per_refactor"] == 3.0
    assert wr["bytes_per_refactor"] == 3 * (10 + 1000 + 100) * kb          # the solve kernel is not part of a refactorisation
    wt = d["whole_trace"]
    assert wt["bytes_per_refactor"] == 3 * (10 + 1000 + 100 + 50) * kb and wt["dispatches_per_refactor"] == 4.0
    assert wt["command"] == "a test"
    import bench
    assert d["kernel_sources_sha1"] == bench.kernel_sources_hash()


def test_committed_counter_files_are_the_ones_the_bench_line_will_quote():
    import bench
    newest = {}
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cfg*_counters.json"))):
        cfg = os.path.basename(p).split("_cfg")[1].split("_")[0]
        newest[cfg] = p
    assert {"2a", "3", "5"} <= set(newest)
    for cfg, p in newest.items():
        d = json.load(open(p))
        assert d["workload"] == cfg
        got, why = bench.pmc_counters(cfg)
        if d["kernel_sources_sha1"] == bench.kernel_sources_hash():
            assert got is not None and got["source"] == os.path.relpath(p, ROOT)
        else:                # kernels edited after the last counter run: the line must say so instead of quoting stale bytes
            assert got is None or got["source"] != os.path.relpath(p, ROOT)
            assert got is not None or "other kernel sources" in why
