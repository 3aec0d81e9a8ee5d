"""Mirror of AVQA/model/Swin_AVQAModel.py (the 512-d head variant that AVQA/test.py:8 imports): the same backbone and QA head as
Swin_AVQAModel_V1.py, with the backbone features first projected down -- `avqatask_yb_fc_v` Linear(1536, 512) on the positive and the
negative video tokens, `avqatask_yb_fc_a` Linear(1536, 128) on the audio tokens (:1472-1473, :1772-1783) -- an extra
`avqatask_fc_a1` Linear(128, 512) + ReLU in front of `avqatask_fc_a2` (:1421, :1798), and every head width 512 instead of 1536
(:1421-1470).  Runs on libstgcma_hip.so through ..ops_head like the V1 head."""
from .Swin_AVQAModel_V1 import QstEncoder, SwinTransformer2D_Adapter_AVQA as _V1  # noqa: F401


class SwinTransformer2D_Adapter_AVQA(_V1):
    HEAD_DIM = 512
    PROJECT_FEATURES = True
