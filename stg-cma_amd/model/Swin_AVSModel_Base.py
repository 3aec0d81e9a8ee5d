"""Module path of the reference's AVS/model/Swin_AVSModel_Base.py (imported by AVS/run_adapt_avs.py:16 as `model.Swin_AVSModel_Base`):
the Swin-B-width AVS model.  The two reference files differ only in the hard-coded stage widths of the decoder's Linear taps
(:1490-1495); the mirror's taps follow `embed_dim`, so one implementation (Swin_AVSModel.py) serves both names."""
from .Swin_AVSModel import (Classifier_Module, FeatureFusionBlock, Interpolate, ResidualConvUnit,  # noqa: F401
                            SwinTransformer2D_Adapter_AVS, SwinTransformer2D_Adapter_AVS_Base, TPAVIModule)
