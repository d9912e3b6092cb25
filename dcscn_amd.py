"""Importable alias of the ``dcscn-super-resolution_amd`` package (its directory name has hyphens).

``import dcscn_amd`` returns the package object itself, so ``dcscn_amd.engine`` etc. work and
``from dcscn_amd import engine`` resolves through the real package.
"""

import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)

_pkg = importlib.import_module("dcscn-super-resolution_amd")
for _sub in ("build", "ckpt", "engine", "flags", "imaging", "shard", "model"):
    importlib.import_module("dcscn-super-resolution_amd." + _sub)
sys.modules[__name__] = _pkg
