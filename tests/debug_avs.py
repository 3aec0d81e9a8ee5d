"""Debug (checker script, lives under tests/ because it imports the oracle): TPAVI gradient norms -- HIP decoder, oracle on HIP taps, oracle on oracle taps -- same script, same upstream."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import stgcma
import test_avs_decoder_gpu as TG
from params import seeded_tensor
from stgcma import ops_dec
import oracle.avs_decoder as OD, oracle.swin as OS
gpu = torch.device('cuda:0')
m, z, cfg, names = TG._build_full(gpu)
B, seed = cfg["B"], cfg["seed"]
a = seeded_tensor((B, 5, 224, 224), seed + 1, 0.5).to(gpu); v = seeded_tensor((B, 5, 3, 224, 224), seed + 2).to(gpu)
Pc = {k: (val.detach().cpu().float().clone() if val.is_floating_point() else val.detach().cpu().clone()) for k, val in m.state_dict().items()}
dn = [n for n in names if n.startswith("avstask_")]
keys = ("avstask_tpavi_b1.g.weight", "avstask_tpavi_b2.g.weight", "avstask_tpavi_b3.g.weight", "avstask_tpavi_b4.g.weight", "avstask_x2_linear.weight", "avstask_x3_linear.weight")
with torch.no_grad():
    ms, a_feat = m.forward_features(a, v)
    oo = OS.swin_backbone(Pc, a.cpu(), v.cpu(), cfg)
def ups_for(pred, fmaps, afeas):
    return [seeded_tensor(pred.shape, seed + 3, 1e-2)] + [seeded_tensor(f.shape, seed + 10 + i, 1e-2) for i, f in enumerate(fmaps)] + [seeded_tensor(x.shape, seed + 20 + i, 1e-1) for i, x in enumerate(afeas)]
def run_oracle(taps, fa, label):
    for n in dn: Pc[n].grad = None; Pc[n].requires_grad_(True)
    t = [x.detach().clone().requires_grad_(True) for x in taps]; f = fa.detach().clone().requires_grad_(True)
    pc, fc, ac = OD.avs_decoder(Pc, t, f, B, 5, bn_training=True)
    sum((o * u).sum() for o, u in zip([pc] + list(fc) + list(ac), ups_for(pc, fc, ac))).backward()
    print(label, {k.replace("avstask_", ""): round(float(Pc[k].grad.norm()), 1) for k in keys})
run_oracle(oo["taps"], oo["f_a"], "oracle / oracle taps ")
run_oracle([t.float().cpu() for t in ms], a_feat.float().cpu(), "oracle / HIP taps    ")
ms_g = [t.clone().requires_grad_(True) for t in ms]; af_g = a_feat.clone().requires_grad_(True)
pred, fmaps, afeas = ops_dec.avs_decoder_forward(m, ms_g, af_g, B, 5, True)
sum((o * u.to(gpu)).sum() for o, u in zip([pred] + list(fmaps) + list(afeas), ups_for(pred, fmaps, afeas))).backward()
d = dict(m.named_parameters())
print("HIP / HIP taps       ", {k.replace("avstask_", ""): round(float(d[k].grad.norm()), 1) for k in keys})
