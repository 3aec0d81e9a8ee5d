"""How well conditioned is the AVS dense decoder's gradient under bf16 rounding?  CPU only, fp32 ORACLE only (test infrastructure):
the decoder of the avs_full_tiny_evalbn fixture is run on the oracle's own taps, with chosen tensors rounded to bf16 on the way
forward (straight-through) and / or their gradients rounded on the way back, and d(taps) / parameter gradients are compared with the
unrounded run.

    python tools/avs_grad_conditioning.py                 # the table quoted in DESIGN.md section 8
    python tools/avs_grad_conditioning.py y,wz             # a chosen set of rounding points

Rounding points -- forward: conv (every Conv2d output), x (TPAVI input), tpg (theta / phi / g), y (theta . M), wz (W_z.0 output),
bn (BatchNorm output), sum (w_y + x), z (LayerNorm output), allf (all of them); backward: allg (the gradient of every one of them).
Finding: gradients alone rounded: 0.5-2 %; ONE forward tensor rounded (relative 2^-9): 4-8 % on d(taps) -- the ReLU masks of
ResidualConvUnit / output_conv flip for ~eps of the units and the gradient moves by ~sqrt(eps)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.avs_decoder as OD                                     # noqa: E402
from oracle.swin import swin_backbone                               # noqa: E402
from test_oracle_cpu import _avs_evalbn_state, load_case            # noqa: E402
from params import seeded_tensor                                    # noqa: E402

MODE = set()


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _g(t):
    return _RoundGrad.apply(t) if "allg" in MODE else t


def _f(tag, t):
    return t + (t.bfloat16().float() - t).detach() if (tag in MODE or "allf" in MODE) else t


def _tpavi(P, pre, x, audio, bn_training=False, bn_stats=None, eps=1e-5):          # oracle.avs_decoder.tpavi with the rounding points
    Bn, C, T, H, W = x.shape
    x = _g(_f("x", x))
    audio_temp = F.linear(audio, P[pre + ".align_channel.weight"], P[pre + ".align_channel.bias"])
    au = audio_temp.permute(0, 2, 1)[:, :, :, None, None].expand(Bn, C, T, H, W)

    def c1(name, t):
        return F.conv3d(t, P[f"{pre}.{name}.weight"], P[f"{pre}.{name}.bias"])
    Ci = P[pre + ".g.weight"].shape[0]
    g_x = _g(_f("tpg", c1("g", x))).reshape(Bn, Ci, -1).permute(0, 2, 1)
    theta_x = _g(_f("tpg", c1("theta", x))).reshape(Bn, Ci, -1).permute(0, 2, 1)
    phi_x = _g(_f("tpg", c1("phi", au))).reshape(Bn, Ci, -1)
    f = theta_x @ phi_x
    y = _g(_f("y", (f / f.shape[-1]) @ g_x)).permute(0, 2, 1).reshape(Bn, Ci, T, H, W)
    w_y = _g(_f("wz", c1("W_z.0", y)))
    mean, var = P[pre + ".W_z.1.running_mean"], P[pre + ".W_z.1.running_var"]
    sh = (1, C, 1, 1, 1)
    w_y = _g(_f("bn", (w_y - mean.view(sh)) * torch.rsqrt(var.view(sh) + eps) * P[pre + ".W_z.1.weight"].view(sh) + P[pre + ".W_z.1.bias"].view(sh)))
    z = _g(_f("sum", w_y + x)).permute(0, 2, 3, 4, 1)
    z = F.layer_norm(z, (C,), P[pre + ".norm_layer.weight"], P[pre + ".norm_layer.bias"])
    return _g(_f("z", z.permute(0, 4, 1, 2, 3))), audio_temp


_conv0 = OD.conv


def _conv(P, pre, x, dilation=1, padding=None):
    return _g(_f("conv", _conv0(P, pre, x, dilation, padding)))


def main():
    global MODE
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    OD.tpavi, OD.conv = _tpavi, _conv
    z, cfg, shapes, names = load_case("avs_full_tiny_evalbn")
    P = _avs_evalbn_state(z, cfg, shapes)
    for n in names:
        P[n].requires_grad_(True)
    B, seed = cfg["B"], cfg["seed"]
    a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5)
    v = seeded_tensor((B, 5, 3, 224, 224), seed + 2)
    with torch.no_grad():
        out = swin_backbone(P, a, v, cfg)
    taps, fa = [t.detach() for t in out["taps"]], out["f_a"].detach()

    def run(mode):
        global MODE
        MODE = set(mode)
        tp = [t.clone().requires_grad_(True) for t in taps]
        f_ = fa.clone().requires_grad_(True)
        for n in names:
            P[n].grad = None
        pred, _, _ = OD.avs_decoder(P, tp, f_, B, 5, bn_training=False)
        (pred * seeded_tensor(pred.shape, seed + 3, 1e-2)).sum().backward()
        return pred.detach(), [t.grad for t in tp] + [f_.grad], {n: P[n].grad.clone() for n in names if P[n].grad is not None}

    def rel(x, y):
        return float((x - y).norm() / y.norm())

    ref = run(())
    modes = [m.split(",") for m in sys.argv[1:]] or [["allg"], ["conv"], ["x"], ["tpg"], ["y"], ["wz"], ["bn"], ["sum"], ["z"], ["allf"], ["allf", "allg"]]
    for mode in modes:
        pred, dt, pg = run(mode)
        errs = [(rel(pg[n], ref[2][n]), n) for n in pg if float(ref[2][n].norm()) > 0]
        print("%-12s pred %.2e | d(taps) %s | parameter gradients: median %.3f, worst %.3f (%s)" %
              (",".join(mode), rel(pred, ref[0]), " ".join("%.3f" % rel(x, y) for x, y in zip(dt, ref[1])),
               float(np.median([e for e, _ in errs])), max(errs)[0], max(errs)[1]), flush=True)


if __name__ == "__main__":
    main()
