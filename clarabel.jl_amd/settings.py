"""Settings mirror of the reference's ``Settings{T}`` (src/settings.jl:70-148).

Only the fields the KKT path and the stand-in IPM caller read are kept; defaults are the
struct defaults of the reference (the docstring table at settings.jl:44-57 is stale)."""
from dataclasses import dataclass, field

import numpy as np

_EPS = float(np.finfo(np.float64).eps)


@dataclass
class Settings:
    max_iter: int = 200                         # settings.jl:72
    time_limit: float = float("inf")
    verbose: bool = False
    max_step_fraction: float = 0.99             # :75
    tol_gap_abs: float = 1e-8                   # :78-83
    tol_gap_rel: float = 1e-8
    tol_feas: float = 1e-8
    tol_infeas_abs: float = 1e-8
    tol_infeas_rel: float = 1e-8
    tol_ktratio: float = 1e-6
    reduced_tol_gap_abs: float = 5e-5           # :90-95
    reduced_tol_gap_rel: float = 5e-5
    reduced_tol_feas: float = 1e-4
    reduced_tol_infeas_abs: float = 5e-12
    reduced_tol_infeas_rel: float = 5e-5
    reduced_tol_ktratio: float = 1e-4
    equilibrate_enable: bool = True             # :98-101
    equilibrate_max_iter: int = 10
    equilibrate_min_scaling: float = 1e-4
    equilibrate_max_scaling: float = 1e4
    linesearch_backtrack_step: float = 0.8      # :104-106 (line search of the non-symmetric cones)
    min_switch_step_length: float = 1e-1
    min_terminate_step_length: float = 1e-4
    max_threads: int = 0                        # :110
    direct_solve_method: str = "hip"            # :114 (reference default :auto)
    static_regularization_enable: bool = True   # :117-119
    static_regularization_constant: float = 1e-8
    static_regularization_proportional: float = _EPS * _EPS
    dynamic_regularization_enable: bool = True  # :122-124 (never read by the reference either)
    dynamic_regularization_eps: float = 1e-13
    dynamic_regularization_delta: float = 2e-7
    iterative_refinement_enable: bool = True    # :127-132
    iterative_refinement_reltol: float = 1e-13
    iterative_refinement_abstol: float = 1e-12
    iterative_refinement_max_iter: int = 10
    iterative_refinement_stop_ratio: float = 5.0
    presolve_enable: bool = True                # :135
    chordal_decomposition_enable: bool = False  # :139 (out of scope here; configs run with it off)
    # device selection for the :hip KKT solver (not in the reference)
    device_id: int = 0
    # SURVEY section 8(f) rows widened beyond the reference's plugin seam (all off by default = the reference's call pattern)
    device_residuals: bool = False            # N4: residuals_update! computed by the plugin from the resident P, A
    device_scaling: bool = False              # N1: update_scaling! / get_Hs! of the symmetric cones formed by the plugin from (s, z)
    device_reduced: bool = False              # N2: the reduced-system algebra of kkt_solve! (d tau dots, quad_form, dx, dz) done by the plugin
    extra: dict = field(default_factory=dict)
