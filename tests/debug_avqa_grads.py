"""Debug (checker script, lives under tests/ because it imports the oracle): HIP AVQA head vs the fp32 oracle head on the SAME backbone features (isolates head error from backbone error)."""
import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import stgcma
from golden_util import build_state, load_case
from stgcma import recipe, ops_head
from stgcma.model import Swin_AVQA
from params import seeded_tensor
import oracle.avqa_head as OH
gpu = torch.device('cuda:0')
z, cfg, shapes, names = load_case("avqa_full_tiny")
m = Swin_AVQA.SwinTransformer2D_Adapter_AVQA(pretrained=None, num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"]).eval()
P = build_state(shapes, cfg["seed"], kind="swin", T=cfg["num_frames"])
sd = m.state_dict()
for k in sd:
    if sd[k].is_floating_point() and not k.endswith("attn_mask"): sd[k] = P[k]
m.load_state_dict(sd, strict=True); m = m.to(gpu)
for n, p in m.named_parameters(): p.requires_grad = recipe.is_trainable(n)
B, T, seed = cfg["B"], cfg["num_frames"], cfg["seed"]
a = seeded_tensor((B, T, 224, 224), seed + 1, 0.5).to(gpu); v = seeded_tensor((B, T, 3, 224, 224), seed + 2).to(gpu); vn = seeded_tensor((B, T, 3, 224, 224), seed + 3).to(gpu)
q = torch.as_tensor(z["question"]).to(gpu)
with torch.no_grad():
    f_v, f_a, f_n = m.forward_features(a, v, vn)
fv, fa = f_v.clone().requires_grad_(True), f_a.clone().requires_grad_(True)
out = ops_head.avqa_head_forward(m, fv, fa, f_n, q, B, T, False)
gs = [seeded_tensor(o.shape, seed + 5 + i).to(gpu) for i, o in enumerate(out)]
sum((o * g).sum() for o, g in zip(out, gs)).backward()
Pc = {k: val.detach().cpu().float().clone() for k, val in m.state_dict().items()}
hn = [n for n in names if n.startswith("avqatask_")]
for n in hn: Pc[n].requires_grad_(True)
fvc, fac = f_v.cpu().clone().requires_grad_(True), f_a.cpu().clone().requires_grad_(True)
oc = OH.avqa_head(Pc, fvc, fac, f_n.cpu(), q.cpu(), B, T)
sum((o * g.cpu()).sum() for o, g in zip(oc, gs)).backward()
def rel(a_, b_): return float((a_.detach().cpu().float() - b_.detach()).norm() / max(float(b_.detach().norm()), 1e-12))
for nm, o1, o2 in zip(("out_qa", "match_posi", "match_nega"), out, oc): print(nm, "rel", rel(o1, o2))
print("d f_v rel", rel(fv.grad, fvc.grad), "d f_a rel", rel(fa.grad, fac.grad))
d = dict(m.named_parameters())
rows = sorted(((rel(d[n].grad, Pc[n].grad), n) for n in hn), reverse=True)
for r, n in rows[:16]: print(f"{n:55s} rel {r:.4f}")
