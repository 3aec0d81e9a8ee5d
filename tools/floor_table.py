"""Round 6: the step FLOOR of the headline's dataflow, from one bench.py run's per-class table.

    python tools/floor_table.py gpurun_out/bench_detail.json > profiles/r06_floor.md

For every GEMM class and every non-GEMM family class of the eager pass: launches per step, algorithmic bytes and FLOPs per step (every
operand / output once; 2 M N K), the time those bytes take at the measured copy rate (6.3 TB/s, MI355X_MICROARCH.md; peak 8) and the time the FLOPs
take at the best rate this tree's GEMM main loop has measured on long-K shapes (1.25 PFLOP/s = 0.50 of the 2.5 PFLOP/s dense bf16 peak; the
vendor library is within -7 .. +2 % of it, profiles/r05_blaslt_ab.txt), the class FLOOR = the larger of the two (perfect overlap of bytes and
arithmetic inside a kernel, nothing else resident), beside the measured milliseconds.  The step floor is the SUM of the class floors: consecutive
kernels of a chain do not overlap, and the two micro-batch chains of the default step form share one memory system and one set of matrix pipes,
so running two at once moves no class below its own floor."""
import json
import sys

HBM = 6.3e12          # measured copy rate, bytes / s
MFMA = 1.25e15        # best measured long-K rate of gemm_nt_8ph_kernel, FLOP / s
ROWS = {2007040: 0, 501760: 1, 125440: 2, 31360: 3}


def stage_of_rows(m):
    for r, s in ROWS.items():
        if abs(m - r) <= 0.02 * r or abs(m - r // 2) <= 0.02 * r:
            return f"s{s}"
    return "-"


def main(path):
    d = json.load(open(path))
    rows = []
    for c in d["roofline_classes"]:
        n = c["launches_per_step"]
        fl = c["gflop_per_launch"] * 1e9
        by = c["algorithmic_bytes_per_launch"]
        m = fl / (2.0 * c["N"] * c["K"])
        rows.append(dict(kind="gemm", name=f"{c['kernel'].replace('gemm_nt_', '').replace('_kernel', '')} N{c['N']} K{c['K']} {c['epilogue'] or 'plain'}", stage=stage_of_rows(m),
                         n=n, gb=by * n / 1e9, tf=fl * n / 1e12, ms=c["est_ms_per_step"]))
    for c in d["roofline_family_classes"]:
        n = c["launches_per_step"]
        by = c["algorithmic_bytes_per_launch"]
        fl = c["tflops"] * 1e12 * c["avg_launch_us"] * 1e-6
        rows.append(dict(kind=c["family"], name=f"{c['family']} {c['key']}", stage="", n=n, gb=by * n / 1e9, tf=fl * n / 1e12, ms=c["est_ms_per_step"]))
    for r in rows:
        r["t_hbm"] = r["gb"] * 1e9 / HBM * 1e3
        r["t_mfma"] = r["tf"] * 1e12 / MFMA * 1e3
        r["floor"] = max(r["t_hbm"], r["t_mfma"])
    hd = d.get("headline", {})
    print(f"# Step floor of the headline's dataflow ({hd.get('metric', '?')}: {hd.get('value', '?')} {hd.get('unit', '')}, {hd.get('ms_per_step', '?')} ms per replayed step)\n")
    print(__doc__.split("\n\n", 2)[2].strip() + "\n")
    tot = lambda key, f=lambda r: True: sum(r[key] for r in rows if f(r))
    g = lambda r: r["kind"] == "gemm"
    ng = lambda r: r["kind"] != "gemm"
    print("| part | launches | algorithmic GB | algorithmic TFLOP | bytes at 6.3 TB/s (ms) | FLOPs at 1.25 PFLOP/s (ms) | floor (ms) | measured, eager pass (ms) | measured / floor |")
    print("|---|---|---|---|---|---|---|---|---|")
    for label, f in (("GEMM classes", g), ("non-GEMM families", ng), ("**step**", lambda r: True)):
        print(f"| {label} | {tot('n', f):.0f} | {tot('gb', f):.1f} | {tot('tf', f):.2f} | {tot('t_hbm', f):.1f} | {tot('t_mfma', f):.1f} | **{tot('floor', f):.1f}** | {tot('ms', f):.1f} | {tot('ms', f) / tot('floor', f):.2f} |")
    print(f"\nBytes alone: {tot('gb'):.0f} GB / 6.3 TB/s = {tot('t_hbm'):.1f} ms; FLOPs alone: {tot('tf'):.1f} TFLOP / 1.25 PFLOP/s = {tot('t_mfma'):.1f} ms; "
          f"per-class maximum summed: **{tot('floor'):.1f} ms** = {32.0 / tot('floor') * 1e3:.0f} clips/s = "
          f"{32.0 / tot('floor') * 1e3 * 1.5875 / 2500 * 100:.1f} % of the 2.5 PFLOP/s peak at 1 587.5 GFLOP per clip.\n")
    print("## Per class, largest floor first\n")
    print("| class | stage | x / step | GB / step | TFLOP / step | bytes (ms) | FLOPs (ms) | floor (ms) | measured (ms) | measured / floor | bound |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -r["floor"]):
        if r["ms"] < 0.05 and r["floor"] < 0.05:
            continue
        print(f"| `{r['name']}` | {r['stage']} | {r['n']:.0f} | {r['gb']:.2f} | {r['tf']:.3f} | {r['t_hbm']:.2f} | {r['t_mfma']:.2f} | {r['floor']:.2f} | {r['ms']:.2f} | "
              f"{r['ms'] / max(r['floor'], 1e-9):.2f} | {'bytes' if r['t_hbm'] >= r['t_mfma'] else 'FLOPs'} |")
    fam = {}
    for r in rows:
        k = "GEMM" if r["kind"] == "gemm" else r["kind"]
        a = fam.setdefault(k, dict(floor=0.0, ms=0.0, gb=0.0))
        a["floor"] += r["floor"]; a["ms"] += r["ms"]; a["gb"] += r["gb"]
    print("\n## Per family\n\n| family | GB / step | floor (ms) | measured (ms) | above its floor (ms) |\n|---|---|---|---|---|")
    for k, a in sorted(fam.items(), key=lambda kv: -(kv[1]["ms"] - kv[1]["floor"])):
        print(f"| {k} | {a['gb']:.1f} | {a['floor']:.2f} | {a['ms']:.2f} | {a['ms'] - a['floor']:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_detail.json")
