"""Byte / time ledger of one training step from the three rocprofv3 runs of tools/ledger_run.sh.

usage: python tools/ledger.py gpurun_out/ledger/<tag> profiles/<tag> [gemm_seq.json]
       ->  profiles/<tag>_ledger.json, profiles/<tag>_ledger.md (+ profiles/<tag>_pmc_by_class.json with the sequence file)

With the launch-ordered GEMM class list of one step (bench.py --gemm-seq, written by the trace run) the i-th gemm_nt* dispatch of the
step is the i-th entry of the list (kernel names are checked), which gives every GEMM CLASS (kernel x N x K x epilogue) its own PMC
bytes and duration: what bench.py quotes as roofline.traffic.

A "class" is kernel x grid size (x workgroup size): the same kernel on the same problem shape.  One step = the dispatches between the
last two `adam_multi_kernel` launches (the optimizer step closes a step), so warm-up casts are excluded.  Per class:
  launches per step, mean duration (kernel-trace run, no counters), HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KiB
  (MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes; WRITE_SIZE exact for 16-byte stores),
  achieved GB/s = bytes / duration.
The three runs execute the same program, so dispatch i of the step is the same launch in each; names are checked.
"""
import csv, gzip, json, os, re, sys, collections


def rows(path):
    with gzip.open(path, "rt") as f:
        rs = list(csv.DictReader(f))
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))          # file order is completion order in the trace, dispatch order under --pmc
    return rs


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def last_step(rs, key="Kernel_Name"):
    idx = [i for i, r in enumerate(rs) if "adam_multi_kernel" in r[key]]
    if len(idx) < 2:
        raise SystemExit("need >= 2 optimizer steps in the trace")
    return rs[idx[-2] + 1: idx[-1] + 1]


SIMDS, XCDS = 1024, 8          # MI355X: 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE comes summed over the 8 XCDs


def mfma_util(mf_cycles, gui_cycles):
    """rocprof MFMA utilisation of a set of dispatches: matrix-pipe busy cycles (SQ_VALU_MFMA_BUSY_CYCLES, summed over every SIMD of the
    chip; = 16 x the number of 16x16x32 bf16 MFMAs, MI355X_MICROARCH.md) / (GPU-active cycles per XCD x 1024 SIMDs)."""
    return round(mf_cycles / max(gui_cycles / XCDS * SIMDS, 1.0), 4)


def is_gemm(name):
    """dispatches of stg_gemm_nt: the tile kernels and, since round 6, the adapters' down-projection row stream it routes to (csrc/skinny.hip)"""
    return name.startswith("gemm_nt") or name.startswith("skinny_down")


def by_class(tr, fe, wr, seq_path, dst, summary, mf=None, gu=None):
    seq = json.load(open(seq_path))["step"]
    extra = list(zip(mf, gu)) if mf is not None else [None] * len(tr)
    gd = [(a, b, c, e) for a, b, c, e in zip(tr, fe, wr, extra) if is_gemm(short(a["Kernel_Name"]))]
    if len(gd) != len(seq):
        raise SystemExit(f"{len(gd)} gemm dispatches in the step, {len(seq)} in the sequence file")
    cls = collections.OrderedDict()
    for (a, b, c, e), (kern, M, N, K, epi, nbytes) in zip(gd, seq):
        if short(a["Kernel_Name"]).split("<")[0] != kern.split("<")[0] or (("<" in kern) and short(a["Kernel_Name"]) != kern):
            raise SystemExit(f"sequence mismatch: trace {short(a['Kernel_Name'])} vs log {kern}")
        d = cls.setdefault(f"{kern}|{N}|{K}|{epi}", {"n": 0, "ns": 0, "bytes": 0.0, "alg": 0.0, "flops": 0.0, "rows": set(), "mf": 0.0, "gui": 0.0})
        d["n"] += 1
        if e is not None:
            d["mf"] += float(e[0]["Counter_Value"]); d["gui"] += float(e[1]["Counter_Value"])
        d["ns"] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        d["bytes"] += (2 * float(b["Counter_Value"]) + float(c["Counter_Value"])) * 1024
        d["alg"] += nbytes
        d["flops"] += 2.0 * M * N * K
        d["rows"].add(M)
    out = {}
    for k, d in cls.items():
        out[k] = {"launches_per_step": d["n"], "us_per_launch": round(d["ns"] / d["n"] / 1e3, 2), "ms_per_step": round(d["ns"] / 1e6, 3),
                  "hbm_bytes_per_launch": round(d["bytes"] / d["n"]), "algorithmic_bytes_per_launch": round(d["alg"] / d["n"]),
                  "traffic_over_algorithmic": round(d["bytes"] / max(d["alg"], 1.0), 3), "tflops": round(d["flops"] / max(d["ns"], 1) / 1e3, 1),
                  "hbm_gbs": round(d["bytes"] / max(d["ns"], 1), 1), "rows": sorted(d["rows"])}
        if d["gui"] > 0:
            out[k]["mfma_util_pmc"] = mfma_util(d["mf"], d["gui"])
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"]))
    json.dump({"note": "per GEMM class (kernel|N|K|epilogue): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of `bench.py --steps 2 "
                       "--warmup 1`, bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch (gfx950: FETCH_SIZE tallies 128-B requests as 64 B), "
                       "dispatch i of the step joined with entry i of bench.py --gemm-seq; durations from the counter-free trace run",
               # round 6: which tree / command the counters were taken on -- bench.py quotes it in roofline.traffic_source
               "meta": {"commit": os.environ.get("LEDGER_COMMIT", "unknown"), "command": os.environ.get("LEDGER_COMMAND", "bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline")},
               "step": summary, "classes": out}, open(dst + "_pmc_by_class.json", "w"), indent=1)
    print("per-class file:", dst + "_pmc_by_class.json", len(out), "classes")
    for k, v in list(out.items())[:16]:
        print(f"  {k:62s} n={v['launches_per_step']:>3d} us={v['us_per_launch']:>7.1f} PMC MB={v['hbm_bytes_per_launch']/1e6:>7.1f} alg MB={v['algorithmic_bytes_per_launch']/1e6:>7.1f} "
              f"x{v['traffic_over_algorithmic']:.2f} TF={v['tflops']}")


def main(src, dst, seq_path=None):
    import os
    tr = last_step(rows(src + "_trace.csv.gz"))
    fe = last_step([r for r in rows(src + "_fetch.csv.gz") if r["Counter_Name"] == "FETCH_SIZE"])
    wr = last_step([r for r in rows(src + "_write.csv.gz") if r["Counter_Name"] == "WRITE_SIZE"])
    if not (len(tr) == len(fe) == len(wr)):
        raise SystemExit(f"dispatch counts differ: {len(tr)} {len(fe)} {len(wr)}")
    mf = gu = None
    if os.path.exists(src + "_mfma.csv.gz") and os.path.exists(src + "_gui.csv.gz"):
        mf = last_step([r for r in rows(src + "_mfma.csv.gz") if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"])
        gu = last_step([r for r in rows(src + "_gui.csv.gz") if r["Counter_Name"] == "GRBM_GUI_ACTIVE"])
        if not (len(mf) == len(gu) == len(tr)):
            raise SystemExit(f"dispatch counts differ (MFMA passes): {len(tr)} {len(mf)} {len(gu)}")
    cls = collections.OrderedDict()
    t0, t1 = int(tr[0]["Start_Timestamp"]), int(tr[-1]["End_Timestamp"])
    for i, (a, b, c) in enumerate(zip(tr, fe, wr)):
        if not (short(a["Kernel_Name"]) == short(b["Kernel_Name"]) == short(c["Kernel_Name"])):
            raise SystemExit(f"dispatch order differs: {a['Kernel_Name'][:60]} / {b['Kernel_Name'][:60]} / {c['Kernel_Name'][:60]}")
        grid = int(a["Grid_Size_X"]) * int(a["Grid_Size_Y"]) * int(a["Grid_Size_Z"])
        wg = int(a["Workgroup_Size_X"]) * int(a["Workgroup_Size_Y"]) * int(a["Workgroup_Size_Z"])
        k = (short(a["Kernel_Name"]), grid // wg, wg)
        d = cls.setdefault(k, {"n": 0, "ns": 0, "fetch_kb": 0.0, "write_kb": 0.0, "vgpr": int(a["VGPR_Count"]) + int(a["Accum_VGPR_Count"]),
                               "lds": int(a["LDS_Block_Size"]), "scratch": int(a["Scratch_Size"]), "mf": 0.0, "gui": 0.0})
        d["n"] += 1
        if mf is not None:
            if not (short(mf[i]["Kernel_Name"]) == short(gu[i]["Kernel_Name"]) == short(a["Kernel_Name"])):
                raise SystemExit(f"dispatch order differs (MFMA passes): {a['Kernel_Name'][:60]}")
            d["mf"] += float(mf[i]["Counter_Value"]); d["gui"] += float(gu[i]["Counter_Value"])
        d["ns"] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        d["fetch_kb"] += float(b["Counter_Value"])
        d["write_kb"] += float(c["Counter_Value"])
    out = []
    for (name, wgs, wg), d in cls.items():
        byts = (2 * d["fetch_kb"] + d["write_kb"]) * 1024
        out.append({"kernel": name, "workgroups": wgs, "wg_size": wg, "launches_per_step": d["n"], "us_per_launch": round(d["ns"] / d["n"] / 1e3, 2),
                    "ms_per_step": round(d["ns"] / 1e6, 3), "hbm_mb_per_launch": round(byts / d["n"] / 1e6, 2), "hbm_gb_per_step": round(byts / 1e9, 3),
                    "read_gb_per_step": round(2 * d["fetch_kb"] * 1024 / 1e9, 3), "write_gb_per_step": round(d["write_kb"] * 1024 / 1e9, 3),
                    "achieved_gbs": round(byts / max(d["ns"], 1), 1), "vgpr": d["vgpr"], "lds": d["lds"], "scratch": d["scratch"],
                    **({"mfma_util_pmc": mfma_util(d["mf"], d["gui"])} if d["gui"] > 0 else {})})
    out.sort(key=lambda r: -r["ms_per_step"])
    tot_ms = sum(r["ms_per_step"] for r in out)
    tot_gb = sum(r["hbm_gb_per_step"] for r in out)
    gemm_ms = sum(r["ms_per_step"] for r in out if is_gemm(r["kernel"]))
    summary = {"step_kernel_ms": round(tot_ms, 2), "step_wall_ms": round((t1 - t0) / 1e6, 2), "launches": sum(r["launches_per_step"] for r in out),
               "hbm_gb_per_step": round(tot_gb, 1), "gemm_ms": round(gemm_ms, 2), "non_gemm_ms": round(tot_ms - gemm_ms, 2),
               "gemm_gb": round(sum(r["hbm_gb_per_step"] for r in out if is_gemm(r["kernel"])), 1),
               "bytes": "(2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch, rocprofv3 --pmc in separate runs; durations from the counter-free run"}
    if mf is not None:
        tmf, tgu = sum(d["mf"] for d in cls.values()), sum(d["gui"] for d in cls.values())
        gmf = sum(d["mf"] for k, d in cls.items() if is_gemm(k[0])); ggu = sum(d["gui"] for k, d in cls.items() if is_gemm(k[0]))
        summary["mfma_util_pmc"] = mfma_util(tmf, tgu)
        summary["mfma_util_pmc_gemm_kernels"] = mfma_util(gmf, ggu)
        summary["mfma_util_pmc_what"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), both summed over the step's dispatches, separate "
                                         "rocprofv3 --pmc runs: the share of SIMD-cycles in which the matrix pipe is busy, at the clock the chip actually runs")
    json.dump({"summary": summary, "classes": out}, open(dst + "_ledger.json", "w"), indent=1)
    with open(dst + "_ledger.md", "w") as f:
        f.write(f"# Step ledger ({src.split('/')[-1]})\n\n{json.dumps(summary)}\n\n")
        f.write("| kernel | workgroups | x/step | us/launch | ms/step | MB/launch | GB/step | GB/s | MFMA busy |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in out:
            f.write(f"| `{r['kernel']}` | {r['workgroups']} | {r['launches_per_step']} | {r['us_per_launch']} | {r['ms_per_step']} | "
                    f"{r['hbm_mb_per_launch']} | {r['hbm_gb_per_step']} | {r['achieved_gbs']} | {r.get('mfma_util_pmc', '')} |\n")
    print(json.dumps(summary))
    if seq_path:
        by_class(tr, fe, wr, seq_path, dst, summary, mf, gu)
    for r in out[:70]:
        print(f"{r['kernel'][:58]:58s} wg={r['workgroups']:>7d} n={r['launches_per_step']:>3d} us={r['us_per_launch']:>8.1f} ms={r['ms_per_step']:>7.3f} "
              f"MB={r['hbm_mb_per_launch']:>8.1f} GB/s={r['achieved_gbs']:>7.1f} vgpr={r['vgpr']} lds={r['lds']}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
