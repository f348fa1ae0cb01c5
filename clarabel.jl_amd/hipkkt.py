"""ctypes binding of ``libclarabel_hipkkt.so`` (C ABI declared in include/hipkkt.h).

This is the Python stand-in for the ``ccall`` layer a Julia maintainer would add;
it fails loudly when the HIP library is missing — there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The PRODUCT is libclarabel_hipkkt.so.  CLARABEL_HIPKKT_TESTING=1 (set by tests/conftest.py and by the developer tools that compare a
# mechanism with its off state) selects the testing build of the same sources, the only one whose hipkkt_debug_set accepts switches.
_TESTING = os.environ.get("CLARABEL_HIPKKT_TESTING", "0") == "1"
LIB_PATH = os.environ.get("CLARABEL_HIPKKT_LIB", os.path.join(_HERE, "libclarabel_hipkkt_testing.so" if _TESTING else "libclarabel_hipkkt.so"))

# switches of hipkkt_debug_set (struct DebugOpts, csrc/hipkkt_internal.h).  The C library reads no environment variable for them; this
# binding forwards HIPKKT_<KEY> from os.environ to the testing build whenever a handle is created (tests use monkeypatch.setenv) and
# refuses to create a handle on the production library while one of them is set.
DEBUG_KEYS = ["PLAN_CACHE", "FB_EXTRA", "FB_STREAM", "FB_V2", "FORCE_TWIN", "NO_GRAPH", "NO_PERSIST", "FULL_TILES", "FRONT_BLOCK", "SPLIT_K",
              "DENSE_TRI", "ORDERING", "NO_FRONT", "HOST_ASSEMBLY", "FRONT_BLOCK_MIN_ROWS", "SUPERHOP", "DEBUG_FLAGS", "SPIN_LIMIT",
              "PERSIST_RETRY", "ACCURATE", "FB_EXTRA_PW", "GATHER_OVERLAP", "GATHER_SIDE_BLOCKS", "GATHER_SORT"]

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

ABI_VERSION = 4   # include/hipkkt.h HIPKKT_ABI_VERSION (checked when the library is loaded)

# every symbol include/hipkkt.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "hipkkt_default_opts", "hipkkt_is_available", "hipkkt_abi_version", "hipkkt_trim_cache", "hipkkt_create", "hipkkt_create_from_parts", "hipkkt_destroy",
    "hipkkt_get_dims", "hipkkt_info", "hipkkt_get_cost_model", "hipkkt_get_kkt", "hipkkt_get_perm",
    "hipkkt_get_dsigns", "hipkkt_get_map", "hipkkt_get_sparse_map", "hipkkt_update_values", "hipkkt_scale_values",
    "hipkkt_set_hs", "hipkkt_set_hs_dev", "hipkkt_set_hs_psd", "hipkkt_set_cone_types", "hipkkt_update_scaling", "hipkkt_update_scaling_dev", "hipkkt_block_products", "hipkkt_set_soc", "hipkkt_set_soc_batch", "hipkkt_set_genpow",
    "hipkkt_update_P", "hipkkt_update_A", "hipkkt_refactor", "hipkkt_setrhs", "hipkkt_setrhs_dev", "hipkkt_solve",
    "hipkkt_solve_dev", "hipkkt_solve_multi", "hipkkt_solve_multi_dev", "hipkkt_kkt_solve_reduced", "hipkkt_kkt_solve_reduced_dev", "hipkkt_ldl_solve", "hipkkt_get_timing", "hipkkt_reset_timing", "hipkkt_get_profile", "hipkkt_get_profile_launches", "hipkkt_set_profiling",
    "hipkkt_get_counters", "hipkkt_debug_dump", "hipkkt_debug_extra_tiles", "hipkkt_debug_set", "hipkkt_debug_is_testing_build", "hipkkt_set_qb", "hipkkt_residuals", "hipkkt_residuals_dev",
    "hipkkt_selftest_mfma", "hipkkt_box_probe", "hipkkt_last_error",
]


class Opts(C.Structure):
    _fields_ = [("index_base", C.c_int32), ("supernode_max_width", C.c_int32), ("relax_supernodes", C.c_int32),
                ("update_policy", C.c_int32), ("update_batch", C.c_int32), ("front_min_panels", C.c_int32),
                ("dynamic_reg_eps", C.c_double), ("dynamic_reg_delta", C.c_double),
                ("amd_dense_scale", C.c_double), ("user_perm", C.c_void_p)]


class HipKKTError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipKKTError(f"{LIB_PATH} is missing: build it with clarabel.jl_amd/csrc/build.sh "
                          "(__graft_entry__.build()); there is no CPU fallback for the KKT path")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.hipkkt_default_opts.argtypes = [C.POINTER(Opts)]
    L.hipkkt_default_opts.restype = None
    L.hipkkt_is_available.restype = i32
    L.hipkkt_abi_version.restype = i32
    if L.hipkkt_abi_version() != ABI_VERSION:       # signatures change between versions: never call through a mismatch
        raise HipKKTError(f"{LIB_PATH} implements ABI version {L.hipkkt_abi_version()}, this binding was written against "
                          f"{ABI_VERSION} (include/hipkkt.h HIPKKT_ABI_VERSION): rebuild the library")
    L.hipkkt_create.restype = i32
    L.hipkkt_create.argtypes = [i32, i64, _i64p, _i64p, _f64p, _i64p, C.POINTER(Opts), C.POINTER(vp)]
    L.hipkkt_create_from_parts.restype = i32
    L.hipkkt_create_from_parts.argtypes = [i32, i64, i64, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p, i64, _i64p, _i32p,
                                           _i32p, _i64p, C.POINTER(Opts), C.POINTER(vp)]
    L.hipkkt_destroy.argtypes = [vp]
    L.hipkkt_destroy.restype = None
    L.hipkkt_get_dims.argtypes = [vp, _i64p]
    L.hipkkt_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.hipkkt_get_cost_model.argtypes = [vp, _f64p]
    L.hipkkt_get_kkt.argtypes = [vp, vp, vp, vp]
    L.hipkkt_get_perm.argtypes = [vp, _i64p]
    L.hipkkt_get_dsigns.argtypes = [vp, _i64p]
    L.hipkkt_get_map.argtypes = [vp, i32, _i64p]
    L.hipkkt_get_sparse_map.argtypes = [vp, i64, i32, vp, C.POINTER(i64)]
    L.hipkkt_update_values.argtypes = [vp, _i64p, _f64p, i64]
    L.hipkkt_scale_values.argtypes = [vp, _i64p, i64, f64]
    L.hipkkt_set_hs.argtypes = [vp, _f64p, i64]
    L.hipkkt_set_hs_dev.argtypes = [vp, vp, i64]
    L.hipkkt_set_hs_psd.argtypes = [vp, i64, _i64p, _i64p, _f64p]
    L.hipkkt_set_cone_types.argtypes = [vp, i64, _i32p]
    L.hipkkt_update_scaling.argtypes = [vp, _f64p, _f64p, vp, vp, vp, vp, C.POINTER(i32)]
    L.hipkkt_update_scaling_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.hipkkt_block_products.argtypes = [vp, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.hipkkt_set_soc.argtypes = [vp, i64, f64, _f64p, _f64p, i64]
    L.hipkkt_set_soc_batch.argtypes = [vp, i64, _f64p, _f64p, _f64p, i64]
    L.hipkkt_set_genpow.argtypes = [vp, i64, f64, _f64p, _f64p, _f64p]
    L.hipkkt_update_P.argtypes = [vp, _f64p, i64]
    L.hipkkt_update_A.argtypes = [vp, _f64p, i64]
    L.hipkkt_refactor.argtypes = [vp, i32, f64, f64, C.POINTER(f64), C.POINTER(i64)]
    L.hipkkt_setrhs.argtypes = [vp, _f64p, _f64p]
    L.hipkkt_setrhs_dev.argtypes = [vp, vp]
    L.hipkkt_solve.argtypes = [vp, vp, vp, i32, f64, f64, i64, f64, C.POINTER(i64)]
    L.hipkkt_solve_dev.argtypes = [vp, vp, i32, f64, f64, i64, f64, C.POINTER(i64)]
    L.hipkkt_ldl_solve.argtypes = [vp, _f64p, _f64p]
    L.hipkkt_solve_multi.argtypes = [vp, i64, _f64p, _f64p, vp, vp, i32, f64, f64, i64, f64, vp]
    L.hipkkt_solve_multi_dev.argtypes = [vp, i64, vp, vp, i32, f64, f64, i64, f64, vp]
    L.hipkkt_kkt_solve_reduced.argtypes = [vp, _f64p, _f64p, _f64p, _f64p, i32, vp, vp, _f64p, i32, f64, f64, i64, f64, vp]
    L.hipkkt_kkt_solve_reduced_dev.argtypes = [vp, vp, _f64p, i32, vp, _f64p, i32, f64, f64, i64, f64, vp]
    L.hipkkt_trim_cache.argtypes = [i32]
    L.hipkkt_get_timing.argtypes = [vp, _f64p]
    L.hipkkt_reset_timing.argtypes = [vp]
    L.hipkkt_get_profile.argtypes = [vp, _f64p, i64]
    L.hipkkt_set_profiling.argtypes = [vp, i32]
    L.hipkkt_get_profile_launches.argtypes = [vp, vp, vp, vp, i64, C.POINTER(i64)]
    L.hipkkt_get_counters.argtypes = [vp, _i64p, i64]
    L.hipkkt_set_qb.argtypes = [vp, _f64p, _f64p]
    L.hipkkt_residuals.argtypes = [vp, _f64p, _f64p, _f64p, f64, f64, vp, vp, vp, vp, vp, _f64p]
    L.hipkkt_residuals_dev.argtypes = [vp, vp, f64, f64, vp, _f64p]
    L.hipkkt_debug_dump.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
    L.hipkkt_debug_extra_tiles.argtypes = [i32, i32, i32, C.POINTER(i32)]
    L.hipkkt_debug_extra_tiles.restype = i32
    L.hipkkt_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    L.hipkkt_debug_set.restype = i32
    L.hipkkt_debug_is_testing_build.restype = i32
    L.hipkkt_selftest_mfma.argtypes = [i32, C.POINTER(f64)]
    L.hipkkt_box_probe.argtypes = [i32, vp, i64]
    L.hipkkt_last_error.argtypes = [vp]
    L.hipkkt_last_error.restype = C.c_char_p
    for nm in SYMBOLS:
        f = getattr(L, nm)
        if f.restype is C.c_int:  # default -> status code
            f.restype = i32
    _lib = L
    return L


def sync_debug_switches():
    """Forward the HIPKKT_<KEY> variables of os.environ to hipkkt_debug_set (testing build); called before every create."""
    L = lib()
    wanted = {k: os.environ.get("HIPKKT_" + k) for k in DEBUG_KEYS}
    if not L.hipkkt_debug_is_testing_build():
        bad = [k for k, v in wanted.items() if v is not None]
        if bad:
            raise HipKKTError(f"HIPKKT_{bad[0]} is set but {LIB_PATH} is the production library, which has no switches: "
                              "set CLARABEL_HIPKKT_TESTING=1 to load libclarabel_hipkkt_testing.so")
        return
    for k, v in wanted.items():
        rc = L.hipkkt_debug_set(k.encode(), None if v is None else v.encode())
        if rc != 0:
            raise HipKKTError(f"hipkkt_debug_set({k}, {v!r}) failed ({rc})")


def default_opts(**kw):
    o = Opts()
    lib().hipkkt_default_opts(C.byref(o))
    keep = None
    for k, v in kw.items():
        if k == "user_perm" and v is not None:
            keep = np.ascontiguousarray(v, dtype=np.int64)
            o.user_perm = keep.ctypes.data_as(C.c_void_p).value
        elif v is not None:
            setattr(o, k, v)
    return o, keep


class Handle:
    """Owning wrapper of a ``hipkkt_handle`` (the analogue of the Julia struct + finalizer)."""

    def __init__(self, ptr):
        self.L = lib()
        self.h = ptr
        d = np.zeros(16, dtype=np.int64)
        self.L.hipkkt_get_dims(self.h, d)
        (self.N, self.n, self.m, self.p, self.nnzK, self.nHs, self.nsparse, self.nnzP, self.nnzA, self.nnzL,
         self.nsuper, self.nlevels, self.panel_doubles, self.ntasks, self.etree_height, self.ordering) = (int(v) for v in d)

    @classmethod
    def from_kkt(cls, colptr, rowval, nzval, dsigns, device=0, **optkw):
        L = lib()
        sync_debug_switches()
        o, keep = default_opts(**optkw)
        out = C.c_void_p()
        N = len(colptr) - 1
        rc = L.hipkkt_create(device, N, np.ascontiguousarray(colptr, dtype=np.int64),
                             np.ascontiguousarray(rowval, dtype=np.int64), np.ascontiguousarray(nzval, dtype=np.float64),
                             np.ascontiguousarray(dsigns, dtype=np.int64), C.byref(o), C.byref(out))
        if rc != 0:
            raise HipKKTError(f"hipkkt_create failed ({rc}): {L.hipkkt_last_error(None).decode()}")
        return cls(out)

    @classmethod
    def from_parts(cls, P, A, numel, hs_dense, sparse_kind, dim1, device=0, **optkw):
        L = lib()
        sync_debug_switches()
        o, keep = default_opts(**optkw)
        out = C.c_void_p()
        n, m = P.shape[0], A.shape[0]
        rc = L.hipkkt_create_from_parts(
            device, n, m, np.ascontiguousarray(P.indptr, dtype=np.int64), np.ascontiguousarray(P.indices, dtype=np.int64),
            np.ascontiguousarray(P.data, dtype=np.float64), np.ascontiguousarray(A.indptr, dtype=np.int64),
            np.ascontiguousarray(A.indices, dtype=np.int64), np.ascontiguousarray(A.data, dtype=np.float64), len(numel),
            np.ascontiguousarray(numel, dtype=np.int64), np.ascontiguousarray(hs_dense, dtype=np.int32),
            np.ascontiguousarray(sparse_kind, dtype=np.int32), np.ascontiguousarray(dim1, dtype=np.int64), C.byref(o),
            C.byref(out))
        if rc != 0:
            raise HipKKTError(f"hipkkt_create_from_parts failed ({rc}): {L.hipkkt_last_error(None).decode()}")
        obj = cls(out)
        obj._cone_numel = np.array(numel, dtype=np.int64)
        return obj

    def close(self):
        if getattr(self, "h", None):
            self.L.hipkkt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise HipKKTError(f"{what} failed ({rc}): {self.L.hipkkt_last_error(self.h).decode()}")
        return rc

    @staticmethod
    def _out_ptr(a, need, what):
        """raw pointer of an OUTPUT array the library writes `need` doubles into (None = Julia `nothing`)"""
        if a is None:
            return None
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
                and a.size >= need):
            raise HipKKTError(f"{what}: need a writeable C-contiguous float64 ndarray of at least {need} elements")
        return a.ctypes.data

    # ---- introspection
    def kkt(self):
        colptr = np.zeros(self.N + 1, dtype=np.int64)
        rowval = np.zeros(self.nnzK, dtype=np.int64)
        nzval = np.zeros(self.nnzK)
        self._chk(self.L.hipkkt_get_kkt(self.h, colptr.ctypes.data, rowval.ctypes.data, nzval.ctypes.data), "get_kkt")
        return colptr, rowval, nzval

    def perm(self):
        p = np.zeros(self.N, dtype=np.int64)
        self.L.hipkkt_get_perm(self.h, p)
        return p

    def dsigns(self):
        p = np.zeros(self.N, dtype=np.int64)
        self.L.hipkkt_get_dsigns(self.h, p)
        return p

    def map(self, which):
        cnt = [self.nnzP, self.nnzA, self.nHs, self.n, self.N][which]
        out = np.zeros(max(cnt, 1), dtype=np.int64)
        self._chk(self.L.hipkkt_get_map(self.h, which, out), "get_map")
        return out[:cnt]

    def sparse_map(self, i, which):
        ln = C.c_int64(0)
        self._chk(self.L.hipkkt_get_sparse_map(self.h, i, which, None, C.byref(ln)), "get_sparse_map")
        out = np.zeros(max(ln.value, 1), dtype=np.int64)
        self.L.hipkkt_get_sparse_map(self.h, i, which, out.ctypes.data, C.byref(ln))
        return out[: ln.value]

    def cost_model(self):
        o = np.zeros(8)
        self.L.hipkkt_get_cost_model(self.h, o)
        return dict(flops_factor=o[0], flops_exec=o[1], flops_solve=o[2], bytes_factor=o[3], bytes_solve=o[4],
                    bytes_spmv=o[5], flops_update=o[6], flops_update_dense=o[7])

    def timing(self):
        o = np.zeros(8)
        self.L.hipkkt_get_timing(self.h, o)
        return dict(last_factor_ms=o[0], last_solve_ms=o[1], acc_factor_ms=o[2], acc_solve_ms=o[3], n_factor=int(o[4]),
                    n_solve_calls=int(o[5]), n_ldl_solves=int(o[6]), last_update_ms=o[7])

    def counters(self):
        o = np.zeros(15, dtype=np.int64)
        self.L.hipkkt_get_counters(self.h, o, len(o))
        return dict(sweep_timeouts=int(o[0]), persistent=bool(o[1]), twin_refactors=int(o[2]), twin_exists=bool(o[3]),
                    in_twin=bool(o[4]), ordering=int(o[5]), fronts=int(o[6]), segments=int(o[7]), front_batches=int(o[8]),
                    front_block=bool(o[9]), plan_cache_hits=int(o[10]), plan_cache_misses=int(o[11]), streamed_chain=bool(o[12]),
                    accurate_factorisations=int(o[13]), deferred_gather_entries=int(o[14]))

    def profile_launches(self):
        n = C.c_int64(0)
        self.L.hipkkt_get_profile_launches(self.h, None, None, None, 0, C.byref(n))
        ms, fl, tl = np.zeros(max(n.value, 1)), np.zeros(max(n.value, 1)), np.zeros(max(n.value, 1))
        self.L.hipkkt_get_profile_launches(self.h, ms.ctypes.data, fl.ctypes.data, tl.ctypes.data, n.value, C.byref(n))
        return ms[: n.value], fl[: n.value], tl[: n.value]

    def debug_dump(self, what):
        ln = C.c_int64(0)
        self._chk(self.L.hipkkt_debug_dump(self.h, what, None, 0, C.byref(ln)), "debug_dump")
        out = np.zeros(max(ln.value, 1))
        self._chk(self.L.hipkkt_debug_dump(self.h, what, out.ctypes.data, ln.value, C.byref(ln)), "debug_dump")
        return out[: ln.value]

    def profile(self):
        o = np.zeros(12)
        self.L.hipkkt_get_profile(self.h, o, len(o))
        return dict(update_ms=o[0], dense4_ms=o[1], dense4_flops=o[2], dense4_launches=int(o[3]), front_block_ms=o[4],
                    front_block_launches=int(o[5]), front_block_panels=int(o[6]), front_block_update_flops=o[7],
                    front_block_extra_tiles=int(o[8]), front_block_extra_flops=o[9], refined_blocks=int(o[10]),
                    factorisations_with_refined_blocks=int(o[11]))

    # ---- numeric
    def update_values(self, index, values):
        self._chk(self.L.hipkkt_update_values(self.h, np.ascontiguousarray(index, dtype=np.int64),
                                              np.ascontiguousarray(values, dtype=np.float64), len(index)), "update_values")

    def scale_values(self, index, scale):
        self._chk(self.L.hipkkt_scale_values(self.h, np.ascontiguousarray(index, dtype=np.int64), len(index), scale),
                  "scale_values")

    def set_hs(self, hs):
        self._chk(self.L.hipkkt_set_hs(self.h, np.ascontiguousarray(hs, dtype=np.float64), len(hs)), "set_hs")

    def set_hs_psd(self, hs_off, dims, w_all):
        """Hs blocks of PSD cones formed on the device from W = R R^T (include/hipkkt.h hipkkt_set_hs_psd)."""
        hs_off = np.ascontiguousarray(hs_off, dtype=np.int64)
        dims = np.ascontiguousarray(dims, dtype=np.int64)
        self._chk(self.L.hipkkt_set_hs_psd(self.h, len(dims), hs_off, dims, np.ascontiguousarray(w_all, dtype=np.float64)), "set_hs_psd")

    # ---- N1: update_scaling! + get_Hs! on the device (include/hipkkt.h hipkkt_update_scaling)
    def set_cone_types(self, kinds):
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        self._chk(self.L.hipkkt_set_cone_types(self.h, len(kinds), kinds), "set_cone_types")
        self._n_soc_all = int(np.sum(kinds == 2))
        # doubles of the concatenated n x n R factors of the PSD cones (numel = n (n + 1) / 2), for the length check of update_scaling
        nel = getattr(self, "_cone_numel", np.zeros(0, dtype=np.int64))[kinds == 3]
        nn = ((np.sqrt(8.0 * nel + 1.0) - 1.0) / 2.0 + 0.5).astype(np.int64)
        self._psd_r_len = int(np.sum(nn * nn))

    def update_scaling(self, s, z, psd_R=None, want_outputs=True):
        """-> (ok, w, lam, soc_eta): Hs blocks / sparse second-order terms of K are rewritten on the device from (s, z)."""
        s = np.ascontiguousarray(s, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float64)
        if len(s) != self.m or len(z) != self.m:
            raise ValueError("update_scaling: s and z must have length m")
        R = None if psd_R is None else np.ascontiguousarray(psd_R, dtype=np.float64)
        if R is not None and R.size != self._psd_r_len:
            raise ValueError(f"update_scaling: psd_R must hold {self._psd_r_len} doubles (the n x n factors of the PSD cones), got {R.size}")
        w = np.zeros(max(self.m, 1)) if want_outputs else None
        lam = np.zeros(max(self.m, 1)) if want_outputs else None
        eta = np.zeros(max(self._n_soc_all, 1)) if want_outputs else None
        ok = C.c_int32(0)
        p = lambda a: None if a is None else a.ctypes.data
        self._chk(self.L.hipkkt_update_scaling(self.h, s, z, p(R), p(w), p(lam), p(eta), C.byref(ok)), "update_scaling")
        if not want_outputs:
            return bool(ok.value), None, None, None
        return bool(ok.value), w[: self.m], lam[: self.m], eta[: self._n_soc_all]

    def update_scaling_dev(self, s_ptr, z_ptr, R_ptr=None, w_ptr=None, lam_ptr=None, eta_ptr=None):
        ok = C.c_int32(0)
        self._chk(self.L.hipkkt_update_scaling_dev(self.h, s_ptr, z_ptr, R_ptr, w_ptr, lam_ptr, eta_ptr, C.byref(ok)), "update_scaling_dev")
        return bool(ok.value)

    def set_soc(self, i, eta2, u, v):
        self._chk(self.L.hipkkt_set_soc(self.h, i, eta2, np.ascontiguousarray(u), np.ascontiguousarray(v), len(u)), "set_soc")

    def set_soc_batch(self, eta2, u_all, v_all):
        self._chk(self.L.hipkkt_set_soc_batch(self.h, len(eta2), np.ascontiguousarray(eta2, dtype=np.float64),
                                              np.ascontiguousarray(u_all, dtype=np.float64),
                                              np.ascontiguousarray(v_all, dtype=np.float64), len(u_all)), "set_soc_batch")

    def set_genpow(self, i, sqrtmu, p, q, r):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self._chk(self.L.hipkkt_set_genpow(self.h, i, sqrtmu, f(p), f(q), f(r)), "set_genpow")

    def update_P(self, vals):
        self._chk(self.L.hipkkt_update_P(self.h, np.ascontiguousarray(vals, dtype=np.float64), len(vals)), "update_P")

    def update_A(self, vals):
        self._chk(self.L.hipkkt_update_A(self.h, np.ascontiguousarray(vals, dtype=np.float64), len(vals)), "update_A")

    def refactor(self, static_enable=True, eps_const=1e-8, eps_prop=float(np.finfo(np.float64).eps) ** 2):
        eps = C.c_double(0.0)
        nreg = C.c_int64(0)
        rc = self._chk(self.L.hipkkt_refactor(self.h, int(static_enable), eps_const, eps_prop, C.byref(eps), C.byref(nreg)),
                       "refactor")
        return rc == 0, eps.value, nreg.value

    def setrhs(self, rhsx, rhsz):
        self._chk(self.L.hipkkt_setrhs(self.h, np.ascontiguousarray(rhsx, dtype=np.float64),
                                       np.ascontiguousarray(rhsz, dtype=np.float64)), "setrhs")

    def solve(self, lhsx, lhsz, ir_enable=True, reltol=1e-13, abstol=1e-12, max_iter=10, stop_ratio=5.0):
        steps = C.c_int64(0)
        px = self._out_ptr(lhsx, self.n, "lhsx")
        pz = self._out_ptr(lhsz, self.m, "lhsz")
        rc = self._chk(self.L.hipkkt_solve(self.h, px, pz, int(ir_enable), reltol, abstol, max_iter, stop_ratio,
                                           C.byref(steps)), "solve")
        return rc == 0, steps.value

    def solve_multi(self, rhsx, rhsz, lhsx, lhsz, ir_enable=True, reltol=1e-13, abstol=1e-12, max_iter=10, stop_ratio=5.0):
        """nrhs right-hand sides on one factorisation (include/hipkkt.h hipkkt_solve_multi): rhsx [nrhs, n], rhsz [nrhs, m];
        lhsx / lhsz = writeable arrays of the same shapes (or None).  Returns (ok, steps[nrhs])."""
        rhsx = np.ascontiguousarray(rhsx, dtype=np.float64)
        rhsz = np.ascontiguousarray(rhsz, dtype=np.float64)
        if self.n and self.m:
            rhsx, rhsz = rhsx.reshape(-1, self.n), rhsz.reshape(-1, self.m)
        elif self.n:
            rhsx = rhsx.reshape(-1, self.n)
            rhsz = np.zeros((rhsx.shape[0], 0))
        else:
            rhsz = rhsz.reshape(-1, self.m) if self.m else np.zeros((0, 0))
            rhsx = np.zeros((rhsz.shape[0], 0))
        if rhsx.shape[0] != rhsz.shape[0]:
            raise HipKKTError(f"solve_multi: {rhsx.shape[0]} right-hand sides in rhsx but {rhsz.shape[0]} in rhsz")
        nrhs = rhsx.shape[0]
        steps = np.zeros(max(nrhs, 1), dtype=np.int64)
        px = self._out_ptr(lhsx, nrhs * self.n, "lhsx")
        pz = self._out_ptr(lhsz, nrhs * self.m, "lhsz")
        rc = self._chk(self.L.hipkkt_solve_multi(self.h, nrhs, rhsx, rhsz, px, pz, int(ir_enable), reltol, abstol, max_iter,
                                                 stop_ratio, steps.ctypes.data), "solve_multi")
        return rc == 0, steps[:nrhs]

    def kkt_solve_reduced(self, rhs_x, workz, var_x, tau, kappa, rhs_tau, rhs_kappa, const_pending, lhs_x, lhs_z, ir_enable=True,
                          reltol=1e-13, abstol=1e-12, max_iter=10, stop_ratio=5.0):
        """kkt_solve! between the caller's cone algebra and mul_Hs! (include/hipkkt.h hipkkt_kkt_solve_reduced; needs set_qb).
        Returns (ok, dtau, scal[10], steps[2]); lhs_x / lhs_z = writeable arrays (or None)."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        scal_in = np.array([tau, kappa, rhs_tau, rhs_kappa], dtype=np.float64)
        scal = np.zeros(10)
        steps = np.zeros(2, dtype=np.int64)
        px = self._out_ptr(lhs_x, self.n, "lhs_x")
        pz = self._out_ptr(lhs_z, self.m, "lhs_z")
        rc = self._chk(self.L.hipkkt_kkt_solve_reduced(self.h, f(rhs_x), f(workz), f(var_x), scal_in, int(bool(const_pending)), px, pz, scal,
                                                       int(ir_enable), reltol, abstol, max_iter, stop_ratio, steps.ctypes.data),
                       "kkt_solve_reduced")
        return rc == 0, float(scal[0]), scal, steps

    def kkt_solve_reduced_dev(self, in_ptr, tau, kappa, rhs_tau, rhs_kappa, const_pending, out_ptr, ir_enable=True, reltol=1e-13,
                              abstol=1e-12, max_iter=10, stop_ratio=5.0):
        scal_in = np.array([tau, kappa, rhs_tau, rhs_kappa], dtype=np.float64)
        scal = np.zeros(10)
        steps = np.zeros(2, dtype=np.int64)
        rc = self._chk(self.L.hipkkt_kkt_solve_reduced_dev(self.h, in_ptr, scal_in, int(bool(const_pending)), out_ptr, scal, int(ir_enable),
                                                           reltol, abstol, max_iter, stop_ratio, steps.ctypes.data), "kkt_solve_reduced_dev")
        return rc == 0, float(scal[0]), scal, steps

    def solve_multi_dev(self, nrhs, rhs_ptr, out_ptr, ir_enable=True, reltol=1e-13, abstol=1e-12, max_iter=10, stop_ratio=5.0):
        steps = np.zeros(max(nrhs, 1), dtype=np.int64)
        rc = self._chk(self.L.hipkkt_solve_multi_dev(self.h, nrhs, rhs_ptr, out_ptr, int(ir_enable), reltol, abstol, max_iter,
                                                     stop_ratio, steps.ctypes.data), "solve_multi_dev")
        return rc == 0, steps[:nrhs]

    # device-pointer variants (inputs already resident in HBM, e.g. torch tensors' data_ptr())
    def set_hs_dev(self, ptr, n):
        self._chk(self.L.hipkkt_set_hs_dev(self.h, ptr, n), "set_hs_dev")

    def setrhs_dev(self, ptr):
        self._chk(self.L.hipkkt_setrhs_dev(self.h, ptr), "setrhs_dev")

    def solve_dev(self, out_ptr, ir_enable=True, reltol=1e-13, abstol=1e-12, max_iter=10, stop_ratio=5.0):
        steps = C.c_int64(0)
        rc = self._chk(self.L.hipkkt_solve_dev(self.h, out_ptr, int(ir_enable), reltol, abstol, max_iter, stop_ratio,
                                               C.byref(steps)), "solve_dev")
        return rc == 0, steps.value

    def set_profiling(self, on):
        self.L.hipkkt_set_profiling(self.h, int(on))

    def reset_timing(self):
        self.L.hipkkt_reset_timing(self.h)

    def block_products(self, x, z):
        """Px = Symmetric(P) x, ATz = A' z, Ax = A x from the resident values (include/hipkkt.h hipkkt_block_products)."""
        Px, ATz, Ax = np.zeros(self.n), np.zeros(self.n), np.zeros(self.m)
        self._chk(self.L.hipkkt_block_products(self.h, np.ascontiguousarray(x, dtype=np.float64),
                                               np.ascontiguousarray(z, dtype=np.float64), Px, ATz, Ax), "block_products")
        return Px, ATz, Ax

    def set_qb(self, q, b):
        self._chk(self.L.hipkkt_set_qb(self.h, np.ascontiguousarray(q, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)), "set_qb")

    def residuals(self, x, z, s, tau, kappa, rx, rz, rx_inf, rz_inf, Px):
        """residuals_update! on the device (include/hipkkt.h hipkkt_residuals); returns (dot_qx, dot_bz, dot_sz, dot_xPx, r_tau)"""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        scal = np.zeros(5)
        ptrs = [self._out_ptr(a, k, nm) for a, k, nm in ((rx, self.n, "rx"), (rz, self.m, "rz"), (rx_inf, self.n, "rx_inf"),
                                                          (rz_inf, self.m, "rz_inf"), (Px, self.n, "Px"))]
        self._chk(self.L.hipkkt_residuals(self.h, f(x), f(z), f(s), float(tau), float(kappa), *ptrs, scal), "residuals")
        return tuple(float(v) for v in scal)

    def ldl_solve(self, b):
        x = np.zeros(self.N)
        self._chk(self.L.hipkkt_ldl_solve(self.h, x, np.ascontiguousarray(b, dtype=np.float64)), "ldl_solve")
        return x


def box_probe(device=0):
    """hipkkt_box_probe: the shader clock one busy wavefront gets and the flag round trip between workgroups on the same / on
    different XCDs -- what the latency-bound kernels depend on (include/hipkkt.h)."""
    out = np.zeros(8)
    rc = lib().hipkkt_box_probe(device, out.ctypes.data, 8)
    if rc != 0:
        raise HipKKTError(f"hipkkt_box_probe failed ({rc})")
    return {"shader_ghz_one_busy_wave": round(float(out[0]), 4), "flag_round_trip_ns_same_xcd": float(out[1]),
            "flag_round_trip_ns_other_xcd": float(out[2]), "xcds_seen": int(out[3]), "core_clock_mhz": float(out[4]),
            "memory_clock_mhz": float(out[5]), "compute_units": int(out[6])}
