#!/usr/bin/env python
"""Gate for julia/parity_dump.jl's output: for every problem, :hip against :qdldl of the REAL Clarabel.jl -- status equal, iterations
equal or +-1, objective and residuals within 1e-10 (x within 1e-6: the IPM stops at 1e-8).  usage: compare_parity.py <results dir>"""
import glob
import json
import os
import sys

import numpy as np


def main(d):
    bad = 0
    for f in sorted(glob.glob(os.path.join(d, "*.qdldl.json"))):
        g = f.replace(".qdldl.json", ".hip.json")
        if not os.path.exists(g):
            print(os.path.basename(f), "no :hip result")
            continue
        a, b = json.load(open(f)), json.load(open(g))
        dobj = abs(a["obj_val"] - b["obj_val"]) / max(1.0, abs(a["obj_val"]))
        dres = max(abs(a["r_prim"] - b["r_prim"]), abs(a["r_dual"] - b["r_dual"]))
        dx = float(np.max(np.abs(np.array(a["x"]) - np.array(b["x"]))) / max(1.0, np.max(np.abs(a["x"]))))
        ok = a["status"] == b["status"] and abs(a["iterations"] - b["iterations"]) <= 1 and (a["iterations"] != b["iterations"] or (dobj <= 1e-10 and dres <= 1e-10)) and dx <= 1e-6
        bad += not ok
        print(f"{a['name']:<14} {'PASS' if ok else 'FAIL'}  status {a['status']}/{b['status']}  it {a['iterations']}/{b['iterations']}  |dobj| {dobj:.2e}  |dres| {dres:.2e}  "
              f"|dx| {dx:.2e}  time qdldl {a['solve_time']:.3f} s  hip {b['solve_time']:.3f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "results")))
