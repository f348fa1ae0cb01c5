"""clarabel.jl_amd — MI355X-native KKT linear-system path for Clarabel.jl.

Product only: csrc/ (HIP kernels + the C ABI ``libclarabel_hipkkt.so``), hipkkt.py (ctypes binding = the
Python twin of the Julia ``ccall`` layer), kktsolver.py (``HipKKTSolver`` = host mirror of the
AbstractKKTSolver plugin), settings.py, cone_api.py (cone specification types), jsonio.py (the
reference's JSON problem format), batch.py (one-problem-per-GPU sharding helpers), problems.py
(synthetic generators of the BASELINE.json configs).  The numpy stand-in of the Julia caller that
drives this plugin in tests / bench lives in ``julia_standin/`` and is not imported from here.
"""
from .cone_api import (NonnegativeConeT, PSDTriangleConeT, SecondOrderConeT, ZeroConeT, cones_new_collapsed)
from .settings import Settings

__all__ = ["Settings", "ZeroConeT", "NonnegativeConeT", "SecondOrderConeT", "PSDTriangleConeT", "cones_new_collapsed"]
