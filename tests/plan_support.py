"""ctypes access to tests/support/libplan_check.so (host interpreter of the symbolic plan)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(os.path.join(_HERE, "libplan_check.so"))
        L.plan_check_run.restype = C.c_int
        L.plan_check_run.argtypes = [C.c_int64, _i64p, _i64p, _f64p, _i64p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, _f64p, _f64p, _i64p, _f64p, C.c_int]
        L.plan_check_symmetric_product.restype = C.c_int
        L.plan_check_symmetric_product.argtypes = [C.c_int64, _i64p, _i64p, _f64p, C.c_int, C.c_int, _f64p, _f64p, _f64p]
        _lib = L
    return _lib


def symmetric_product(N, colptr, rowval, nzval, x, first_col=-1, min_dim=128):
    """K x through the plan's symmetric view + dense triangles (host restatement of the refinement's SpMV kernels)"""
    y = np.zeros(N)
    st = np.zeros(3)
    rc = lib().plan_check_symmetric_product(N, np.ascontiguousarray(colptr, dtype=np.int64), np.ascontiguousarray(rowval, dtype=np.int64),
                                            np.ascontiguousarray(nzval, dtype=np.float64), first_col, min_dim,
                                            np.ascontiguousarray(x, dtype=np.float64), y, st)
    return rc, y, dict(triangles=int(st[0]), view_entries=int(st[1]), triangle_dims=int(st[2]))


STAT_NAMES = ["nsuper", "nlevels", "nnzL", "panel_doubles", "ntasks", "ngroups", "etree_height",
              "flops_colcount", "flops_update", "flops_exec", "nreg", "max_group_tasks",
              "nfronts", "ngather_entries", "ndense_groups", "nmapped_tasks"]


def run(N, colptr, rowval, nzval, dsigns, b=None, perm=None, max_width=64, relax=1, policy=0,
        reg_eps=1e-13, reg_delta=2e-7, symbolic_only=False):
    L = lib()
    x = np.zeros(N)
    perm_out = np.zeros(N, dtype=np.int64)
    stats = np.zeros(16)
    if b is None:
        b = np.zeros(N)
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.int64)
        pp = perm.ctypes.data_as(C.c_void_p)
    rc = L.plan_check_run(N, np.ascontiguousarray(colptr, dtype=np.int64), np.ascontiguousarray(rowval, dtype=np.int64),
                          np.ascontiguousarray(nzval, dtype=np.float64), np.ascontiguousarray(dsigns, dtype=np.int64),
                          pp, max_width, relax, policy, reg_eps, reg_delta,
                          np.ascontiguousarray(b, dtype=np.float64), x, perm_out, stats, int(symbolic_only))
    return rc, x, perm_out, dict(zip(STAT_NAMES, stats))
