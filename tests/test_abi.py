"""CPU-side checks of the drop-in boundary: the library builds, loads and exports every symbol
include/hipkkt.h declares; no compute call is made (no GPU here)."""
import os
import re

from clarabel_jl_amd import hipkkt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hipkkt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hipkkt_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_bound_and_exported():
    L = hipkkt.lib()
    decl = _declared_symbols()
    assert decl, "no declarations parsed"
    for s in decl:
        assert hasattr(L, s), f"{s} declared in include/hipkkt.h but not exported"
    assert sorted(hipkkt.SYMBOLS) == decl


def test_both_builds_export_the_same_interface_and_only_the_testing_build_takes_switches():
    """libclarabel_hipkkt.so is the product: it exports everything the header declares, and hipkkt_debug_set refuses every switch
    there (the library reads no HIPKKT_* variable for them: a caller's environment cannot change its arithmetic).  The testing build
    accepts them, checks numbers to their end, and does contain the first form of the front-batch kernel."""
    import ctypes as C
    import subprocess

    pkg = os.path.join(ROOT, "clarabel.jl_amd")
    prod, test = (C.CDLL(os.path.join(pkg, n)) for n in ("libclarabel_hipkkt.so", "libclarabel_hipkkt_testing.so"))
    for L in (prod, test):
        for s in _declared_symbols():
            assert hasattr(L, s), s
        L.hipkkt_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    assert prod.hipkkt_debug_is_testing_build() == 0 and test.hipkkt_debug_is_testing_build() == 1
    for k in hipkkt.DEBUG_KEYS:
        assert prod.hipkkt_debug_set(k.encode(), b"0") != 0, k          # refused
        assert test.hipkkt_debug_set(k.encode(), b"0") == 0, k
        assert test.hipkkt_debug_set(k.encode(), None) == 0, k          # back to the default
    assert prod.hipkkt_debug_set(None, None) == 0                        # "reset everything" is a no-op there
    assert test.hipkkt_debug_set(b"NO_SUCH_SWITCH", b"1") != 0
    for bad in (b"abc", b"1x", b""):                                     # numbers are parsed to their end
        assert test.hipkkt_debug_set(b"ACCURATE", bad) != 0 and test.hipkkt_debug_set(b"SPIN_LIMIT", bad) != 0
    for ok in (b"0", b"0.0", b"00", b"0 ", b"-1", b"64"):
        assert test.hipkkt_debug_set(b"ACCURATE", ok) == 0
    assert test.hipkkt_debug_set(None, None) == 0
    # no environment reads for switches in the product's host sources: VERBOSE and FB_TRACE only
    n = 0
    for f in os.listdir(os.path.join(pkg, "csrc")):
        if f.endswith((".cpp", ".hip", ".h")):
            n += len(re.findall(r"\bgetenv\s*\(", open(os.path.join(pkg, "csrc", f)).read()))
    assert n == 2, n
    # the first form of the front-batch kernel is in the testing build only
    syms = {lib: subprocess.run(["strings", "-a", os.path.join(pkg, lib)], capture_output=True, text=True).stdout
            for lib in ("libclarabel_hipkkt.so", "libclarabel_hipkkt_testing.so")}
    assert "k_front_block2" in syms["libclarabel_hipkkt.so"] and "13k_front_blockI" not in syms["libclarabel_hipkkt.so"]
    assert "13k_front_blockI" in syms["libclarabel_hipkkt_testing.so"]


def test_is_available_never_fails():
    assert hipkkt.lib().hipkkt_is_available() >= 0


def test_default_opts_match_reference_settings():
    o, _ = hipkkt.default_opts()
    assert (o.dynamic_reg_eps, o.dynamic_reg_delta, o.amd_dense_scale) == (1e-13, 2e-7, 1.5)  # settings.jl:123-124, directldl_qdldl.jl:24
    assert o.supernode_max_width == 64 and o.index_base == 0


def test_product_never_imports_oracle_or_the_caller_standin():
    """the product path must not reach into oracle/ (parity would be void) nor into julia_standin/ (the numpy
    stand-in of the Julia caller is test / bench infrastructure, not product)"""
    pkg = os.path.join(ROOT, "clarabel.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".sh")):
                src = open(os.path.join(dp, f)).read()
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+(oracle|julia_standin)", src, flags=re.M), f
                else:
                    assert "oracle" not in src.lower() and "julia_standin" not in src, f
    assert not os.path.exists(os.path.join(pkg, "ipm.py")) and not os.path.exists(os.path.join(pkg, "cones.py"))


def test_extra_tile_choice_is_consistent():
    """hipkkt_factor.cpp fb_extra_tiles_of_stage (host logic, no device): the far tiles of a front batch that ride in the next
    k_front_block launch never include tiles of the next batch's columns, fit the compute units that launch leaves idle, and leave
    the stage's own launch on a step of its cost function (or as small as it can be)."""
    import ctypes as C

    L = hipkkt.lib()
    seen_two = False
    for nd in list(range(0, 600, 37)) + list(range(600, 4200, 101)):
        for ncrit in (0, nd // 8, nd // 3, nd):
            for next_blk in (2, 37, 87, 200, 254):
                pw = C.c_int32(0)
                r = L.hipkkt_debug_extra_tiles(nd, ncrit, next_blk, C.byref(pw))
                assert 0 <= r <= nd - ncrit
                assert pw.value in (1, 2)
                seen_two = seen_two or (r > 0 and pw.value == 2)
                assert r <= 4 * pw.value * max(0, 254 - next_blk)
                if r > 0:
                    m = nd - r
                    assert r >= 64
                    assert m <= 384 or m == ncrit or m == 768 or m % 1024 in (0, 256, 512), (nd, ncrit, next_blk, r)
    assert not seen_two      # round 5: one tile per wavefront (the panel chain of front_block2.hip is shorter than two tile times)
    assert L.hipkkt_debug_extra_tiles(3403, 435, 82, None) > 0 and L.hipkkt_debug_extra_tiles(50, 10, 10, None) == 0
