#!/usr/bin/env python
"""Developer tool: dump the assembled KKT pattern of a bench config for tools/plan_stats.cpp."""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def kkt_pattern_oracle(cfg):
    """exact KKT pattern (incl. dense PSD / SOC expansion blocks) from the oracle's assembly"""
    import clarabel_jl_amd  # noqa: F401
    import julia_standin as cl
    from oracle.kkt_oracle import OracleKKTSolver
    (P, q, A, b, specs), _ = bench.make_problem(cfg)
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
    A = sp.csc_matrix(A); A.sort_indices()
    o = OracleKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings(), ordering="natural").k
    K = sp.csc_matrix((np.ones(len(o.rowval)), o.rowval, o.colptr), shape=(o.N, o.N))
    return K, A.shape[1]


def kkt_pattern(cfg):
    (P, q, A, b, cones), _ = bench.make_problem(cfg)
    n = P.shape[0]; m = A.shape[0]
    Pt = sp.triu(sp.csc_matrix(P))
    # NN-only pattern (diagonal Hs): K = [P A'; A -I] upper
    K = sp.bmat([[Pt + sp.identity(n) * 1e-300, sp.csc_matrix(A).T], [None, -sp.identity(m)]], format="csc")
    K = sp.triu(K, format="csc"); K.sort_indices()
    return K

if __name__ == "__main__":
    cfg, out = sys.argv[1], sys.argv[2]
    if cfg in ("3", "5"):
        K, n = kkt_pattern_oracle(cfg)
        print("n =", n)
    else:
        K = kkt_pattern(cfg)
    with open(out, "wb") as f:
        np.array([K.shape[0], K.nnz], dtype=np.int64).tofile(f)
        K.indptr.astype(np.int64).tofile(f)
        K.indices.astype(np.int64).tofile(f)
    print(K.shape, K.nnz)
