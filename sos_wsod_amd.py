"""Import shim: registers the package directory `sos-wsod_amd/` as module `sos_wsod_amd`."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sos-wsod_amd")
_spec = importlib.util.spec_from_file_location("sos_wsod_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sos_wsod_amd"] = _mod
_spec.loader.exec_module(_mod)
