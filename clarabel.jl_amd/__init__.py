"""clarabel.jl_amd — MI355X-native KKT linear-system path for Clarabel.jl.

Product  : csrc/ (HIP kernels + C ABI ``libclarabel_hipkkt.so``), hipkkt.py (ctypes binding),
           kktsolver.py (``HipKKTSolver`` = host mirror of the AbstractKKTSolver plugin).
Caller   : cones.py, ipm.py — numpy stand-in for the untouched Julia IPM loop (not accelerated).
"""
from .cones import (CompositeCone, NonnegativeConeT, PSDTriangleConeT, SecondOrderConeT, ZeroConeT,
                    cones_new_collapsed)
from .ipm import Solver
from .settings import Settings

__all__ = ["Settings", "Solver", "CompositeCone", "ZeroConeT", "NonnegativeConeT", "SecondOrderConeT",
           "PSDTriangleConeT", "cones_new_collapsed"]
