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


@dataclass(frozen=True)
class ExponentialConeT:
    """cone_api.jl:47-49: no fields, always three rows"""


@dataclass(frozen=True)
class PowerConeT:
    """cone_api.jl:33-36: three rows, exponent alpha"""
    alpha: float


@dataclass(frozen=True)
class GenPowerConeT:
    """cone_api.jl:38-46: exponents alpha (all > 0, summing to one) for the first len(alpha) rows, then dim2 rows"""
    alpha: tuple
    dim2: int

    def __post_init__(self):
        a = tuple(float(v) for v in self.alpha)
        object.__setattr__(self, "alpha", a)
        if not all(v > 0.0 for v in a) or abs(sum(a) - 1.0) > 2.220446049250313e-16 * len(a) / 2:
            raise ValueError("GenPowerConeT: the exponents must be positive and sum to one")


def triangular_number(k: int) -> int:
    return (k * (k + 1)) >> 1


def nvars(spec) -> int:
    """cone_api.jl:56-69: number of rows a cone spec occupies."""
    if isinstance(spec, PSDTriangleConeT):
        return triangular_number(spec.dim)
    if isinstance(spec, (ExponentialConeT, PowerConeT)):
        return 3
    if isinstance(spec, GenPowerConeT):
        return len(spec.alpha) + spec.dim2
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
