"""julia_standin — numpy stand-in for the reference's UNTOUCHED Julia caller of the KKT path.

NOT part of the product.  north_star keeps ``solver.jl`` / ``variables.jl`` / ``residuals.jl`` /
``kktsystem.jl`` and the cone algebra (``src/cones``) in Julia; this image has no ``julia``, so the
tests and ``bench.py`` need *some* caller that drives the ``kktsolver_*`` plugin interface
(src/kktsolvers/kktsolver_defaults.jl:2-47) the way the reference does: ``ipm.py`` (the IPM loop and
the reduced-system algebra), ``cones.py`` (cone algebra: update_scaling!, get_Hs!, step lengths) and ``cones_nonsym.py``
(the Exponential / Power / Generalized Power cones).
The product (``clarabel.jl_amd/``) never imports this package; the dependency points the other way
(the default ``kktsolver_factory`` of ``Solver`` is the product's ``HipKKTSolver``).
"""
from clarabel_jl_amd.cone_api import (ExponentialConeT, GenPowerConeT, NonnegativeConeT, PowerConeT, PSDTriangleConeT,
                                       SecondOrderConeT, ZeroConeT, cones_new_collapsed)
from clarabel_jl_amd.settings import Settings

from .cones import CompositeCone
from .ipm import Solver

__all__ = ["Settings", "Solver", "CompositeCone", "ZeroConeT", "NonnegativeConeT", "SecondOrderConeT",
           "PSDTriangleConeT", "ExponentialConeT", "PowerConeT", "GenPowerConeT", "cones_new_collapsed"]
