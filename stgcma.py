"""Alias so that `import stgcma` (and `from stgcma import kernels, ops, model`) resolves to the package directory
`stg-cma_amd/`, whose hyphenated name is not a Python identifier."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("stg-cma_amd")
sys.modules[__name__] = _pkg
