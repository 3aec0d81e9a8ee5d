"""`model` package with the reference's module paths.  The runners do `import model` and `model.Swin_AVE.<Class>`
(AVE/run_adapt_ave29.py:12,156), `import model.Swin_AVSModel as AVSModel` / `import model.Swin_AVSModel_Base` (AVS/run_adapt_avs.py:15-16),
`import model.Swin_AVQAModel_V1` (AVQA/run_adapt_avqa.py:20) and `import model.Swin_AVQAModel` (AVQA/test.py:8) -- the reference
forgot the __init__.py that makes `import model` work; this one exports the submodules.

Two ways in, ONE set of module objects:
  * `import stgcma.model` / `from stgcma.model import Swin_AVE` (this repository's tests and bench), and
  * top-level `import model` with `stg-cma_amd/` on sys.path ahead of the reference's own model/ directory (INTEGRATION.md
    section 2).  In that case this file is executed under the name `model`; it then imports the real package
    (`stg-cma_amd.model`, whose relative imports reach ops / kernels / the C-ABI library), installs it as `sys.modules['model']`
    and maps every `model.<sub>` import onto `stg-cma_amd.model.<sub>`, so there is never a second copy of a class."""
import sys as _sys

_REAL = "stg-cma_amd.model"

if __name__ != _REAL:
    import importlib as _il
    import importlib.abc as _abc
    import importlib.machinery as _mach
    import os as _os

    _alias = __name__
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # holds the stg-cma_amd/ directory
    if _root not in _sys.path:
        _sys.path.append(_root)

    class _ModelAlias(_abc.MetaPathFinder, _abc.Loader):
        def find_spec(self, fullname, path=None, target=None):
            if fullname.startswith(_alias + "."):
                return _mach.ModuleSpec(fullname, self)
            return None

        def create_module(self, spec):
            return _il.import_module(_REAL + spec.name[len(_alias):])

        def exec_module(self, module):
            pass

    if not any(type(f).__name__ == "_ModelAlias" for f in _sys.meta_path):
        _sys.meta_path.insert(0, _ModelAlias())
    _sys.modules[_alias] = _il.import_module(_REAL)
else:
    from . import Swin_AVE  # noqa: F401
    from . import CLIP_AVE  # noqa: F401,E402
    from . import Swin_AVSModel  # noqa: F401,E402        (AVS/model/Swin_AVSModel.py, Swin-L widths)
    from . import Swin_AVSModel_Base  # noqa: F401,E402   (AVS/model/Swin_AVSModel_Base.py, Swin-B widths)
    from . import Swin_AVQAModel_V1  # noqa: F401,E402    (AVQA/model/Swin_AVQAModel_V1.py, 1536-d QA head)
    from . import Swin_AVQAModel  # noqa: F401,E402       (AVQA/model/Swin_AVQAModel.py, 512-d QA head)
