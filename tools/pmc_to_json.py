#!/usr/bin/env python
"""Writes profiles/<tag>_cfg<cfg>_counters.json -- the hardware-counter figures bench.py quotes for the dominant kernel family (the
big dense updates k_update_dense<4,4> / k_update_dense_tail<P>) -- from SEPARATE rocprofv3 --pmc passes of the same bench command
(as /opt/skills/guides/MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE, the SQ matrix-core counters and GRBM_GUI_ACTIVE.
FETCH_SIZE is doubled (gfx950: it reports half the bytes; calibrated on this access shape, profiles/r03_a_traffic_calibration.txt).
mfma_busy_pct = rocprofv3's MfmaUtil formula: sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max-per-dispatch(GRBM_GUI_ACTIVE) * 1024 SIMDs).
usage: pmc_to_json.py <cfg> <out.json> <fetch.db> <write.db> <sq.db> <grbm.db>"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

FAMILY = ("k_update_denseILi4", "k_update_dense_tail")


def per_dispatch(path):
    """{counter: {kernel: [(sum over instances, max over instances), ...per dispatch]}}"""
    db = sqlite3.connect(path)
    q = ("select i.name, s.kernel_name, p.event_id, sum(p.value), max(p.value) from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id "
         "join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by i.name, s.kernel_name, p.event_id")
    out = {}
    for cn, kn, ev, sm, mx in db.execute(q):
        out.setdefault(cn, {}).setdefault(kn, []).append((sm, mx))
    return out


def family(d, counter, fam=FAMILY):
    rows = []
    for kn, v in d.get(counter, {}).items():
        if any(f in kn for f in fam):
            rows += v
    return rows


def main(cfg, out, fetch_db, write_db, sq_db, grbm_db):
    F, W, S, G = (per_dispatch(p) for p in (fetch_db, write_db, sq_db, grbm_db))
    fr, wr = family(F, "FETCH_SIZE"), family(W, "WRITE_SIZE")
    rd_mb = 2.0 * sum(r[0] for r in fr) * 1024 / max(1, len(fr)) / 1e6
    wr_mb = sum(r[0] for r in wr) * 1024 / max(1, len(wr)) / 1e6
    busy = sum(r[0] for r in family(S, "SQ_VALU_MFMA_BUSY_CYCLES"))
    active = sum(r[1] for r in family(G, "GRBM_GUI_ACTIVE"))
    nS, nG = len(family(S, "SQ_VALU_MFMA_BUSY_CYCLES")), len(family(G, "GRBM_GUI_ACTIVE"))
    mops = sum(r[0] for r in family(S, "SQ_INSTS_VALU_MFMA_MOPS_F64"))
    d = dict(workload=cfg, kernel_family="k_update_dense<4,4> + k_update_dense_tail<P> (every big dense-update launch is one of them)",
             kernel_sources_sha1=bench.kernel_sources_hash(),
             bytes_per_launch=round((rd_mb + wr_mb) * 1e6), read_x2_MB=round(rd_mb, 3), write_MB=round(wr_mb, 3), launches_in_trace=len(fr),
             mfma_busy_pct=round(100.0 * (busy / max(1, nS)) / ((active / max(1, nG)) * 1024.0), 2) if active else None,
             mfma_busy_note="SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / (GRBM_GUI_ACTIVE x 1024 SIMDs), per launch averages of separate passes",
             hw_flops_per_launch=round(mops * 512.0 / max(1, nS)) if mops else None,
             hw_flops_note="SQ_INSTS_VALU_MFMA_MOPS_F64 x 512: what the matrix cores executed, padding of ragged tiles and K remainders included",
             sources=[os.path.basename(os.path.dirname(p)) for p in (fetch_db, write_db, sq_db, grbm_db)])
    # the other big block of a refactorisation: the front-batch kernel (its own in-batch updates AND the far tiles riding in it as
    # extra workgroups) -- matrix-core flops per refactorisation from the counters (one k_init_panels dispatch per refactorisation)
    FB = ("k_front_block",)
    nref = max(1, len(family(S, "SQ_INSTS_VALU_MFMA_MOPS_F64", ("k_init_panels",))))
    fb_mops = sum(r[0] for r in family(S, "SQ_INSTS_VALU_MFMA_MOPS_F64", FB))
    fb_busy = sum(r[0] for r in family(S, "SQ_VALU_MFMA_BUSY_CYCLES", FB))
    fb_active = sum(r[1] for r in family(G, "GRBM_GUI_ACTIVE", FB))
    nfS, nfG = len(family(S, "SQ_VALU_MFMA_BUSY_CYCLES", FB)), len(family(G, "GRBM_GUI_ACTIVE", FB))
    d["front_block"] = dict(launches_in_trace=nfS, refactorisations_in_trace=nref,
                            hw_flops_per_refactor=round(fb_mops * 512.0 / nref) if fb_mops else None,
                            mfma_busy_pct=round(100.0 * (fb_busy / max(1, nfS)) / ((fb_active / max(1, nfG)) * 1024.0), 2) if fb_active else None)
    all_mops = sum(r[0] for kn, v in S.get("SQ_INSTS_VALU_MFMA_MOPS_F64", {}).items() for r in v)
    d["hw_flops_per_refactor_all_kernels"] = round(all_mops * 512.0 / nref) if all_mops else None
    # the factorisation as a whole (what the latency-regime workloads -- cfg 1 / 2b: no big dense-update launch ever runs -- quote):
    # the kernels that run between one k_init_panels and the end of that refactorisation, i.e. everything in FACTOR below
    FACTOR = ("k_scatter_values", "k_soc_batch", "k_maxabs_gather", "k_init_panels", "k_fb_reset", "k_front_block", "k_update_dense",
              "k_update_gather", "k_factor_panel", "k_factor_level", "k_split_reduce", "k_invert_diag", "k_invert_super")
    nrefF = max(1, len(family(F, "FETCH_SIZE", ("k_init_panels",))))
    nrefW = max(1, len(family(W, "WRITE_SIZE", ("k_init_panels",))))
    f_rd = 2.0 * sum(r[0] for r in family(F, "FETCH_SIZE", FACTOR)) * 1024 / nrefF
    f_wr = sum(r[0] for r in family(W, "WRITE_SIZE", FACTOR)) * 1024 / nrefW
    d["whole_refactor"] = dict(bytes_per_refactor=round(f_rd + f_wr), read_x2_MB=round(f_rd / 1e6, 3), write_MB=round(f_wr / 1e6, 3),
                               refactorisations_in_trace=nrefF, dispatches_per_refactor=round(len(family(F, "FETCH_SIZE", FACTOR)) / nrefF, 1),
                               kernels=sorted({kn.split("(")[0][:48] for kn in F.get("FETCH_SIZE", {}) if any(f in kn for f in FACTOR)}),
                               note="FETCH_SIZE x2 + WRITE_SIZE summed over every kernel of a refactorisation (update of the values, "
                                    "regularisation, panels, updates, block inverses) / refactorisations in the trace")
    # every kernel of the trace (solves, refinement, assembly, copies included) per refactorisation: what a batch of small problems
    # (cfg 4) moves per IPM iteration, roughly -- one refactorisation per iteration plus the few of each problem's set-up
    t_rd = 2.0 * sum(r[0] for v in F.get("FETCH_SIZE", {}).values() for r in v) * 1024 / nrefF
    t_wr = sum(r[0] for v in W.get("WRITE_SIZE", {}).values() for r in v) * 1024 / nrefW
    d["whole_trace"] = dict(bytes_per_refactor=round(t_rd + t_wr), read_x2_MB=round(t_rd / 1e6, 3), write_MB=round(t_wr / 1e6, 3),
                            refactorisations_in_trace=nrefF, dispatches_per_refactor=round(sum(len(v) for v in F.get("FETCH_SIZE", {}).values()) / nrefF, 1),
                            command=os.environ.get("PMC_COMMAND"))
    with open(out, "w") as f:
        json.dump(d, f, indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main(*sys.argv[1:7])
