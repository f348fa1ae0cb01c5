"""Cone specification types of the solver API (ref: src/cones/cone_api.jl) — what a problem's ``cones``
vector holds.  The KKT plugin needs them for the JSON wire format (jsonio.py) and the synthetic problem
generators (problems.py); the cone ALGEBRA lives with the caller (julia_standin/cones.py), not here."""
from __future__ import annotations

from dataclasses import dataclass


# ------------------------------------------------------------------ user-facing cone specs
@dataclass(frozen=True)
class ZeroConeT:
    dim: int


@dataclass(frozen=True)
class NonnegativeConeT:
    dim: int


@dataclass(frozen=True)
class SecondOrderConeT:
    dim: int


@dataclass(frozen=True)
class PSDTriangleConeT:
    dim: int  # matrix side dimension


def triangular_number(k: int) -> int:
    return (k * (k + 1)) >> 1


def nvars(spec) -> int:
    """cone_api.jl: number of rows a cone spec occupies."""
    if isinstance(spec, PSDTriangleConeT):
        return triangular_number(spec.dim)
    return spec.dim


def cones_new_collapsed(specs):
    """cone_api.jl:96-153: merge runs of NN / 1-dim SOC / 1-dim PSD into one NN cone, drop empties."""
    out = []
    i = 0
    n = len(specs)

    def collapsible(c):
        return (
            isinstance(c, NonnegativeConeT)
            or (isinstance(c, SecondOrderConeT) and c.dim == 1)
            or (isinstance(c, PSDTriangleConeT) and c.dim == 1)
        )

    while i < n:
        c = specs[i]
        i += 1
        if nvars(c) == 0:
            continue
        if collapsible(c):
            total = nvars(c)
            while i < n:
                d = specs[i]
                if nvars(d) == 0:
                    pass
                elif collapsible(d):
                    total += nvars(d)
                else:
                    break
                i += 1
            out.append(NonnegativeConeT(total))
        else:
            out.append(c)
    return out
