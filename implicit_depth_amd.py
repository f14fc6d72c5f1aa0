"""Import shim: ``import implicit_depth_amd`` loads the package that lives in the
directory ``implicit-depth_amd/`` (the hyphen is mandated by the repo layout and is
not a legal Python identifier).  On import this module replaces itself in
``sys.modules`` with the real package, so ``implicit_depth_amd.cost_volume`` etc.
resolve as ordinary submodules.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "implicit-depth_amd")
_spec = importlib.util.spec_from_file_location(
    "implicit_depth_amd",
    os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["implicit_depth_amd"] = _pkg
_spec.loader.exec_module(_pkg)
