"""Importable alias of the package directory `1d-spectral-optimal-transport_amd/` (whose name is
not a valid Python identifier).  `import sot_amd` / `from sot_amd.losses import Wasserstein1D`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = "1d-spectral-optimal-transport_amd"
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name == _REAL or _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
