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
