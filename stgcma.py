"""Alias so that `import stgcma` (and `from stgcma import kernels, ops, model`, `import stgcma.model.Swin_AVE`) resolves to the
package directory `stg-cma_amd/`, whose hyphenated name is not a Python identifier.  Every `stgcma.<sub>` import is mapped onto
the ONE module object `stg-cma_amd.<sub>` (a second copy under the alias name would carry its own caches and profiling hooks)."""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

_ALIAS, _REAL = "stgcma", "stg-cma_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])     # the existing / freshly imported real module

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
