"""Condense gpurun_out/model_parity_report.txt (+ fp8_parity_report.txt) -- written by the -m gpu tests on the MI355X box -- into the table
the test bounds are derived from:  python tools/parity_report.py > profiles/r03_parity_report.txt
Per fixture: max-abs deviation / tensor max and relative L2 against the REFERENCE golden (tests/golden/*.npz); logits also absolute."""
import collections, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "model_parity_report.txt")
lines = [l.strip() for l in open(src)] if os.path.exists(src) else []
pat = re.compile(r"^(.*?): max/scale=([0-9.e+-]+) relL2=([0-9.e+-]+)(.*)$")
groups = collections.OrderedDict()
absl = []
other = []
for l in lines:
    m = pat.match(l)
    if m:
        what, a, b = m.group(1), float(m.group(2)), float(m.group(3))
        key = re.sub(r"grad\[(.*)\]", lambda mm: "grad[scalar gate]" if "gate_" in mm.group(1) else "grad[per tensor, worst]", what)
        g = groups.setdefault(key, [0.0, 0.0, 0, "", ""])
        if a >= g[0]:
            g[0], g[3] = a, what
        if b >= g[1]:
            g[1], g[4] = b, what
        g[2] += 1
    elif "max abs err" in l or "max-abs" in l:
        absl.append(l)
    elif l:
        other.append(l)
print("# Parity of the HIP path against the reference goldens, measured on MI355X (-m gpu suite; bf16 pipeline vs fp32 reference)")
print("# test bounds = 1.5 x these numbers (tests/test_model_gpu.py, test_vit_gpu.py); scalar gates: noise model, see _block_grad_bounds\n")
print("## logits / outputs, ABSOLUTE (north star: <= 1e-2 max-abs at the reference's initialisation scale)")
for l in dict.fromkeys(absl):
    print("  " + l)
print("\n## per fixture: worst max-abs / tensor max, worst relative L2 (number of tensors)")
print(f"{'fixture / quantity':58s} {'max/scale':>10s} {'relL2':>10s}  n")
for k, g in groups.items():
    print(f"{k:58s} {g[0]:10.3e} {g[1]:10.3e}  {g[2]}")
print("\n## aggregate lines of the AVS / AVQA / ViT tests")
for l in dict.fromkeys(other):
    if not l.startswith("avs grad[") and not l.startswith("avqa grad["):
        print("  " + l[:400])
f8 = os.path.join(ROOT, "gpurun_out", "fp8_parity_report.txt")
if os.path.exists(f8):
    print("\n## fp8 (block-scaled e4m3) frozen-weight path, opt-in (see profiles/r03_fp8_sites.txt for the per-site table)")
    for l in dict.fromkeys(x.strip() for x in open(f8)):
        print("  " + l)
