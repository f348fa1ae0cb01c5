"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper of ``oracle/liboracle_kkt.so`` (the CPU restatement of the reference's
DirectLDLKKTSolver + ``:qdldl`` engine; see kkt_oracle.h / qdldl_oracle.h for the reference
lines each function follows).  ``OracleKKTSolver`` exposes the same ``kktsolver_*`` interface as
the product's ``HipKKTSolver`` so that the stand-in IPM caller can run on either.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle_kkt.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    vp = C.c_void_p
    L.oracle_kkt_assemble.restype = vp
    L.oracle_kkt_assemble.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p,
                                      C.c_int64, _i64p, _i32p, _i32p, _i64p]
    L.oracle_kkt_free.argtypes = [vp]
    L.oracle_kkt_symbolic.restype = C.c_int
    L.oracle_kkt_symbolic.argtypes = [vp, vp, C.c_double, C.c_double]
    L.oracle_kkt_sizes.argtypes = [vp, _i64p]
    for nm in ("colptr", "rowval", "map_P", "map_A", "map_Hs", "map_diagP", "map_diag_full", "dsigns"):
        f = getattr(L, "oracle_kkt_" + nm)
        f.restype = C.POINTER(C.c_int64)
        f.argtypes = [vp]
    L.oracle_kkt_nzval.restype = C.POINTER(C.c_double)
    L.oracle_kkt_nzval.argtypes = [vp]
    L.oracle_kkt_sparse_map.restype = C.POINTER(C.c_int64)
    L.oracle_kkt_sparse_map.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_int64)]
    L.oracle_kkt_set_residual_order.argtypes = [vp, C.c_int]
    L.oracle_kkt_set_residual_order.restype = None
    L.oracle_kkt_last_norms.argtypes = [vp, _f64p]
    L.oracle_kkt_last_norms.restype = C.c_int
    L.oracle_kkt_update_Hs.argtypes = [vp, _f64p]
    L.oracle_kkt_update_soc.argtypes = [vp, C.c_int64, C.c_double, _f64p, _f64p]
    L.oracle_kkt_update_genpow.argtypes = [vp, C.c_int64, C.c_double, _f64p, _f64p, _f64p]
    L.oracle_kkt_update_values.argtypes = [vp, _i64p, _f64p, C.c_int64]
    L.oracle_kkt_scale_values.argtypes = [vp, _i64p, C.c_int64, C.c_double]
    L.oracle_kkt_update_P.argtypes = [vp, _f64p]
    L.oracle_kkt_update_A.argtypes = [vp, _f64p]
    L.oracle_kkt_regularize_and_refactor.restype = C.c_int
    L.oracle_kkt_regularize_and_refactor.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.oracle_kkt_setrhs.argtypes = [vp, _f64p, _f64p]
    L.oracle_kkt_solve.restype = C.c_int
    L.oracle_kkt_solve.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_double,
                                   C.POINTER(C.c_int64)]
    L.oracle_kkt_set_b.argtypes = [vp, _f64p]
    L.oracle_kkt_get_x.argtypes = [vp, _f64p]
    L.oracle_kkt_ldl_solve.argtypes = [vp, _f64p, _f64p]
    L.oracle_kkt_symv.argtypes = [vp, _f64p, _f64p]
    L.oracle_kkt_D.restype = C.POINTER(C.c_double)
    L.oracle_kkt_D.argtypes = [vp]
    L.oracle_kkt_nreg.restype = C.c_int64
    L.oracle_kkt_nreg.argtypes = [vp]
    L.oracle_kkt_sum_colcount_sq.restype = C.c_double
    L.oracle_kkt_sum_colcount_sq.argtypes = [vp]
    _LIB = L
    return L


def mmd_order(N, colptr, rowval):
    """AMD-class fill-reducing order for the oracle, independent of the product's own ordering:
    SuperLU's multiple-minimum-degree on A'+A (scipy), applied to the symmetric KKT pattern.
    The reference uses SuiteSparse AMD through QDLDL.jl (not available here); any such ordering
    only changes rounding."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu

    K = sp.csc_matrix((np.ones(len(rowval)), rowval, colptr), shape=(N, N))
    S = K + K.T
    S = S + sp.identity(N, format="csc") * (abs(S).sum(axis=0).max() + 1.0)  # diagonally dominant -> no pivoting
    lu = splu(sp.csc_matrix(S), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
              options=dict(SymmetricMode=True))
    return np.asarray(lu.perm_c, dtype=np.int64).argsort().astype(np.int64)


class OracleKKT:
    """Thin handle around the C object; all arrays 0-based."""

    def __init__(self, P, A, numel, hs_dense, sparse_kind, dim1):
        L = lib()
        self.L = L
        n = P.shape[0]
        m = A.shape[0]
        self._keep = [np.ascontiguousarray(P.indptr, dtype=np.int64), np.ascontiguousarray(P.indices, dtype=np.int64),
                      np.ascontiguousarray(P.data, dtype=np.float64), np.ascontiguousarray(A.indptr, dtype=np.int64),
                      np.ascontiguousarray(A.indices, dtype=np.int64), np.ascontiguousarray(A.data, dtype=np.float64)]
        self.h = L.oracle_kkt_assemble(n, m, *self._keep, len(numel), np.ascontiguousarray(numel, dtype=np.int64),
                                       np.ascontiguousarray(hs_dense, dtype=np.int32),
                                       np.ascontiguousarray(sparse_kind, dtype=np.int32),
                                       np.ascontiguousarray(dim1, dtype=np.int64))
        if not self.h:
            raise ValueError("oracle_kkt_assemble failed (cone dimensions do not sum to m?)")
        sz = np.zeros(8, dtype=np.int64)
        L.oracle_kkt_sizes(self.h, sz)
        self.N, self.n, self.m, self.p, self.nnzK, self.nHs, self.nsparse, _ = (int(v) for v in sz)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_kkt_free(self.h)
            self.h = None

    def _arr(self, name, count, dtype=np.int64):
        ptr = getattr(self.L, "oracle_kkt_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(max(count, 1),))[:count].copy()

    @property
    def colptr(self):
        return self._arr("colptr", self.N + 1)

    @property
    def rowval(self):
        return self._arr("rowval", self.nnzK)

    @property
    def nzval(self):
        return self._arr("nzval", self.nnzK, np.float64)

    def map(self, name):
        cnt = dict(map_P=len(self._keep[2]), map_A=len(self._keep[5]), map_Hs=self.nHs, map_diagP=self.n,
                   map_diag_full=self.N, dsigns=self.N)[name]
        return self._arr(name, cnt)

    def sparse_map(self, i, which):
        ln = C.c_int64(0)
        ptr = self.L.oracle_kkt_sparse_map(self.h, i, which, C.byref(ln))
        if not ptr or ln.value == 0:          # e.g. the third vector of an SOC map
            return np.zeros(0, dtype=np.int64)
        return np.ctypeslib.as_array(ptr, shape=(ln.value,)).copy()

    def symbolic(self, perm=None, dyn_eps=1e-13, dyn_delta=2e-7):
        if perm is None:
            pp = None
        else:
            self._perm = np.ascontiguousarray(perm, dtype=np.int64)
            pp = self._perm.ctypes.data_as(C.c_void_p)
        if not self.L.oracle_kkt_symbolic(self.h, pp, dyn_eps, dyn_delta):
            raise ValueError("oracle_kkt_symbolic failed (bad permutation?)")
        sz = np.zeros(8, dtype=np.int64)
        self.L.oracle_kkt_sizes(self.h, sz)
        self.nnzL = int(sz[7])

    def factor_D(self):
        """D of the last numeric factorisation in permuted order (copy)"""
        return np.ctypeslib.as_array(self.L.oracle_kkt_D(self.h), shape=(self.N,)).copy()

    def symv(self, x):
        y = np.zeros(self.N)
        self.L.oracle_kkt_symv(self.h, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def ldl_solve(self, b):
        x = np.zeros(self.N)
        self.L.oracle_kkt_ldl_solve(self.h, x, np.ascontiguousarray(b, dtype=np.float64))
        return x


class OracleKKTSolver:
    """Same interface as the product's HipKKTSolver (kktsolver_defaults.jl:2-47), CPU oracle inside.

    ``ordering``: "mmd" (scipy SuperLU MMD, independent of the product), "natural", or an explicit
    permutation array (e.g. the product's, so that both sides do identical flops)."""

    def __init__(self, P, A, cones, m, n, settings, ordering="mmd"):
        self.settings = settings
        self.cones_desc = cones.kkt_descriptors()
        self.k = OracleKKT(P, A, *self.cones_desc)
        self.m, self.n, self.p = m, n, self.k.p
        if isinstance(ordering, str):
            if ordering == "mmd":
                perm = mmd_order(self.k.N, self.k.colptr, self.k.rowval)
            elif ordering == "natural":
                perm = None
            else:
                raise ValueError(ordering)
        else:
            perm = np.asarray(ordering, dtype=np.int64)
        self.k.symbolic(perm, settings.dynamic_regularization_eps, settings.dynamic_regularization_delta)
        self.hs = np.zeros(self.k.nHs)
        self.last_ir_steps = 0
        self.total_ir_steps = 0
        self.nsolves = 0
        self.diagonal_regularizer = 0.0

    # kktsolver_directldl.jl:197-245
    def kktsolver_update(self, cones):
        L, h = self.k.L, self.k.h
        cones.get_Hs(self.hs)
        L.oracle_kkt_update_Hs(h, self.hs)
        si = 0
        for cone in cones:
            if cone.is_sparse_expandable:
                if getattr(cone, "sparse_kind", 1) == 2:      # directldl_datamaps.jl:146-167
                    L.oracle_kkt_update_genpow(h, si, float(np.sqrt(cone.mu)), np.ascontiguousarray(cone.p),
                                               np.ascontiguousarray(cone.q), np.ascontiguousarray(cone.r))
                else:
                    L.oracle_kkt_update_soc(h, si, cone.eta * cone.eta, np.ascontiguousarray(cone.u),
                                            np.ascontiguousarray(cone.v))
                si += 1
        st = self.settings
        eps = C.c_double(0.0)
        ok = L.oracle_kkt_regularize_and_refactor(h, int(st.static_regularization_enable),
                                                  st.static_regularization_constant,
                                                  st.static_regularization_proportional, C.byref(eps))
        self.diagonal_regularizer = eps.value
        return bool(ok)

    def kktsolver_setrhs(self, rhsx, rhsz):
        self.k.L.oracle_kkt_setrhs(self.k.h, np.ascontiguousarray(rhsx, dtype=np.float64),
                                   np.ascontiguousarray(rhsz, dtype=np.float64))

    def kktsolver_solve(self, lhsx, lhsz):
        st = self.settings
        steps = C.c_int64(0)
        px = lhsx.ctypes.data_as(C.c_void_p) if lhsx is not None else None
        pz = lhsz.ctypes.data_as(C.c_void_p) if lhsz is not None else None
        ok = self.k.L.oracle_kkt_solve(self.k.h, px, pz, int(st.iterative_refinement_enable),
                                       st.iterative_refinement_reltol, st.iterative_refinement_abstol,
                                       st.iterative_refinement_max_iter, st.iterative_refinement_stop_ratio,
                                       C.byref(steps))
        self.last_ir_steps = steps.value
        buf = np.zeros(16)
        self.last_norms = buf[: self.k.L.oracle_kkt_last_norms(self.k.h, buf)].copy() if st.iterative_refinement_enable else buf[:0]
        self.total_ir_steps += steps.value
        self.nsolves += 1
        return bool(ok)

    def kktsolver_update_P(self, P):
        self.k.L.oracle_kkt_update_P(self.k.h, np.ascontiguousarray(P.data, dtype=np.float64))

    def kktsolver_update_A(self, A):
        self.k.L.oracle_kkt_update_A(self.k.h, np.ascontiguousarray(A.data, dtype=np.float64))

    def kktsolver_linear_solver_info(self):
        return dict(name="qdldl-oracle", threads=1, direct=True, nnzA=self.k.nnzK, nnzL=self.k.nnzL)
