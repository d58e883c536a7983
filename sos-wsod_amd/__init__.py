"""sos-wsod_amd — MI355X-native (gfx950) implementation of SoS-WSOD's Stage-1 OICR+ hot path.

The directory name is not a Python identifier; import it as `sos_wsod_amd`
(the repo-root `sos_wsod_amd.py` shim registers this package under that name).
Importing the package loads the HIP C-ABI library and fails loudly if it is missing.
"""
from . import _lib  # noqa: F401  (raises ImportError when libsoswsod_hip.so is absent)

__version__ = "0.1.0"
