"""Problem wire format of the reference: reader / writer for the JSON files of
``Clarabel.save_to_file`` / ``load_from_file`` (src/json.jl:21-98 the file layout, :118-140 matrices,
:142-158 / :190-213 cones, :75-110 the +-Inf <-> floatmax convention of the settings).

    {"settings": {...}, "P": CSC, "q": [...], "A": CSC, "b": [...], "cones": [{"NonnegativeConeT": 3}, ...]}
    CSC = {"m": rows, "n": cols, "colptr": [...], "rowval": [...], "nzval": [...]}     (0-based indices)

so that problems can be exchanged with a Julia / Rust Clarabel.  All seven cone types of the reference are carried
(Zero, Nonnegative, SecondOrder, PSDTriangle: {type: dim}; PowerConeT: {type: alpha}; ExponentialConeT: {type: []};
GenPowerConeT: {type: [alpha, dim2]}); the reference writes the data it solves (after presolve and
chordal decomposition, unscaled), and so does this writer: it stores exactly what it is given."""
import dataclasses
import json
import math
import sys

import numpy as np
import scipy.sparse as sp

from .cone_api import (ExponentialConeT, GenPowerConeT, NonnegativeConeT, PowerConeT, PSDTriangleConeT, SecondOrderConeT,
                       ZeroConeT)
from .settings import Settings

_FLOATMAX = sys.float_info.max
_CONES = {"ZeroConeT": ZeroConeT, "NonnegativeConeT": NonnegativeConeT, "SecondOrderConeT": SecondOrderConeT,
          "PSDTriangleConeT": PSDTriangleConeT}


def _lower_csc(M):  # json.jl:131-139
    M = sp.csc_matrix(M)
    M.sort_indices()
    return {"m": int(M.shape[0]), "n": int(M.shape[1]), "colptr": [int(v) for v in M.indptr],
            "rowval": [int(v) for v in M.indices], "nzval": [float(v) for v in M.data]}


def _parse_csc(d):  # json.jl:161-170
    return sp.csc_matrix((np.asarray(d["nzval"], dtype=np.float64), np.asarray(d["rowval"], dtype=np.int64),
                          np.asarray(d["colptr"], dtype=np.int64)), shape=(int(d["m"]), int(d["n"])))


def _lower_cone(c):  # json.jl:142-158: {type name: its single field}; PowerConeT: alpha; ExponentialConeT: (); GenPowerConeT: [alpha, dim2]
    if isinstance(c, PowerConeT):
        return {"PowerConeT": float(c.alpha)}
    if isinstance(c, ExponentialConeT):
        return {"ExponentialConeT": []}
    if isinstance(c, GenPowerConeT):
        return {"GenPowerConeT": [[float(v) for v in c.alpha], int(c.dim2)]}
    for name, typ in _CONES.items():
        if isinstance(c, typ):
            return {name: int(c.dim)}
    raise TypeError(f"cone {c!r} has no JSON form in this package")


def _parse_cone(d):  # json.jl:190-213
    (key, val), = d.items()
    if key == "GenPowerConeT":
        return GenPowerConeT(tuple(float(v) for v in val[0]), int(val[1]))
    if key == "ExponentialConeT":
        return ExponentialConeT()
    if key == "PowerConeT":
        return PowerConeT(float(val))
    if key not in _CONES:
        raise ValueError(f"unknown cone type {key!r}")
    return _CONES[key](int(val))


def _lower_settings(st):  # sanitize_settings!, json.jl:87-97: +-Inf is written as +-floatmax
    out = {}
    for f in dataclasses.fields(st):
        if f.name == "extra":
            continue
        v = getattr(st, f.name)
        if isinstance(v, float) and math.isinf(v):
            v = math.copysign(_FLOATMAX, v)
        out[f.name] = v
    return out


def _parse_settings(d):  # json.jl:176-188 + desanitize_settings!, :100-110.  Unknown keys (fields of the reference
    st = Settings()      # that this mirror does not carry) are kept in `extra`.
    names = {f.name: f.type for f in dataclasses.fields(st)}
    for k, v in d.items():
        if k in names and k != "extra":
            cur = getattr(st, k)
            if isinstance(cur, bool):
                v = bool(v)
            elif isinstance(cur, int):
                v = int(v)
            elif isinstance(cur, float):
                v = float(v)
                if abs(v) == _FLOATMAX:
                    v = math.copysign(math.inf, v)
            setattr(st, k, v)
        else:
            st.extra[k] = v
    return st


def save_to_file(filename, P, q, A, b, cones, settings=None):
    """ref: save_to_file(solver, filename), json.jl:21-58 (the data are written as given)."""
    data = {"settings": _lower_settings(settings if settings is not None else Settings()), "P": _lower_csc(P),
            "q": [float(v) for v in np.asarray(q, dtype=np.float64)], "A": _lower_csc(A),
            "b": [float(v) for v in np.asarray(b, dtype=np.float64)], "cones": [_lower_cone(c) for c in cones]}
    with open(filename, "w") as f:
        json.dump(data, f)


def load_from_file(filename, settings=None):
    """ref: load_from_file(filename[, settings]), json.jl:61-85.  Returns (P, q, A, b, cones, settings); explicitly
    passed settings replace the ones stored in the file."""
    with open(filename) as f:
        d = json.load(f)
    P = _parse_csc(d["P"])
    A = _parse_csc(d["A"])
    q = np.asarray(d["q"], dtype=np.float64)
    b = np.asarray(d["b"], dtype=np.float64)
    cones = [_parse_cone(c) for c in d["cones"]]
    if settings is None:
        settings = _parse_settings(d.get("settings", {}))
    return P, q, A, b, cones, settings
