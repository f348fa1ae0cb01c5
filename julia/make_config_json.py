#!/usr/bin/env python
"""Writes the BASELINE.json configurations as Clarabel JSON problem files (the reference's wire format, src/json.jl:118-210, through
clarabel.jl_amd/jsonio.py = SURVEY section 8(f) row N3) for julia/parity_dump.jl.  Problems are the generators of
clarabel.jl_amd/problems.py with the seeds SURVEY section 8(d) fixes.  usage: make_config_json.py <outdir> [cfg ...]
   cfgs: 1 2a 2b 3 5 and b<seed> for a problem of the cfg-4 batch (default: 1 2a 2b 3 5 b100 b126 b200)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import clarabel_jl_amd  # noqa: E402,F401
from clarabel_jl_amd import jsonio, problems  # noqa: E402
from clarabel_jl_amd.settings import Settings  # noqa: E402
import bench  # noqa: E402


def main(outdir, cfgs):
    os.makedirs(outdir, exist_ok=True)
    for c in cfgs:
        if c.startswith("b"):
            P, q, A, b, cones = problems.batch_problem(int(c[1:]))
            name = f"cfg4_seed{int(c[1:])}"
        else:
            (P, q, A, b, cones), _ = bench.make_problem(c)
            name = f"cfg{c}"
        st = Settings()
        st.chordal_decomposition_enable = False      # BASELINE.json cfg 5: chordal decomposition off
        path = os.path.join(outdir, name + ".json")
        jsonio.save_to_file(path, P, q, A, b, cones, st)
        print("wrote", path, f"n={A.shape[1]} m={A.shape[0]} cones={len(cones)}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "julia", "problems"), sys.argv[2:] or ["1", "2a", "2b", "3", "5", "b100", "b126", "b200"])
