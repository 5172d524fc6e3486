"""Import shim: the package directory is named ``caesar-mrcnn_amd`` (the hyphen is the
reference project's spelling) which Python cannot import by name.  Importing
``caesar_mrcnn_amd`` loads that directory as a regular package under this module name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "caesar-mrcnn_amd")
_spec = _ilu.spec_from_file_location(
    "caesar_mrcnn_amd", _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["caesar_mrcnn_amd"] = _mod
_spec.loader.exec_module(_mod)
