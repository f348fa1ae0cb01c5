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
