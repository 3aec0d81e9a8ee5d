"""`model` package with the reference's module paths: the runners do `import model` and then `model.Swin_AVE.<Class>`
(AVE/run_adapt_ave29.py:12,156) -- the reference forgot the __init__.py that makes that work; this one exports the
submodules.  Put the directory that contains this package (stg-cma_amd/) on sys.path ahead of the reference's AVE/ dir."""
from . import Swin_AVE  # noqa: F401
from . import CLIP_AVE  # noqa: F401,E402
from . import Swin_AVS  # noqa: F401,E402      (reference: AVS/model/Swin_AVSModel.py -- backbone)
from . import Swin_AVQA  # noqa: F401,E402     (reference: AVQA/model/Swin_AVQAModel_V1.py -- backbone)
