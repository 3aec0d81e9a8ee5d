"""Algorithmic GFLOP per clip of the AVS dense decoder, by BASELINE.md section 2's rule: torch's FlopCounterMode over the
fp32 CPU restatement (oracle/avs_decoder.py) of AVS/model/Swin_AVSModel_Base.py:1824-1894, forward + backward, weight gradients for the trainable
avstask_* parameters, data gradients into the four backbone taps and the audio feature; B = 1 clip of T = 5 frames at 224^2.
TPAVI: the reference (and the oracle) form the explicit (T H W) x (T H W) affinity in `dot` mode -- 4 N^2 C_i FLOPs per block forward -- which
is associativity away from theta (phi^T g) / N, 4 N C_i^2: the product computes the collapsed form, and THAT is what the bench line's FLOP count
takes (the conservative choice: crediting the explicit form would add ~390 GFLOP per clip of arithmetic nobody needs to do).
    python tools/avs_decoder_flops.py          (CPU, ~1 minute, ~6 GB)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils.flop_counter import FlopCounterMode

import oracle.avs_decoder as OD


def main():
    torch.manual_seed(0)
    os.environ.setdefault("STG_ALLOW_CPU_IMPORT", "1")
    import stgcma  # noqa: F401
    from stgcma.model import Swin_AVSModel
    m = Swin_AVSModel.SwinTransformer2D_Adapter_AVS_Base(patch_size=[1, 4, 4], img_size=224, num_frames=5, embed_dim=128, depths=[2, 2, 18, 2],
                                                         num_heads=[4, 8, 16, 32], window_size=7, pretrained=None, ftmode="fusion",
                                                         adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125])
    P = {k: v.detach().clone().float() for k, v in m.state_dict().items() if k.startswith("avstask_")}
    for k, v in P.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    B, T = 1, 5
    taps = [torch.randn(B * T, (56 >> s) ** 2, 128 << s, requires_grad=True) for s in range(4)]
    f_a = torch.randn(B * T, 49, 1024, requires_grad=True)
    with FlopCounterMode(display=False) as fc:
        pred, fmaps, afeas = OD.avs_decoder(P, taps, f_a, B, T, bn_training=True)
        (pred.square().mean() + sum(f.square().mean() for f in fmaps)).backward()
    tot = fc.get_total_flops()
    per = {str(k): v for k, v in fc.get_flop_counts()[next(iter(fc.get_flop_counts()))].items()} if fc.get_flop_counts() else {}
    bmm = sum(v for k, v in per.items() if "bmm" in k)
    Ci = P["avstask_tpavi_b1.g.weight"].shape[0]
    collapsed = sum(3 * 4.0 * (T * (56 >> s) ** 2) * Ci * Ci for s in range(4))        # forward 4 N Ci^2 per block, backward twice that
    print("per op (GFLOP per clip):", {k: round(v / 1e9, 2) for k, v in per.items()})
    print(f"decoder, TPAVI affinity explicit as the reference computes it: {tot / 1e9:.1f} GFLOP per clip (bmm {bmm / 1e9:.1f})")
    print(f"decoder, TPAVI collapsed (what the bench line counts):        {(tot - bmm + collapsed) / 1e9:.1f} GFLOP per clip (affinity {collapsed / 1e9:.2f})")


if __name__ == "__main__":
    main()
