"""Import shim: the package directory is named ``clarabel.jl_amd`` (a dot is not a legal
Python module name), so ``import clarabel_jl_amd`` loads that directory as a package."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkgdir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "clarabel.jl_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_pkgdir, "__init__.py"), submodule_search_locations=[_pkgdir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
