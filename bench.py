#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): clips/sec forward+backward(+Adam step on the adapters) of Swin-B + STG-CMA,
ftmode='fusion', AVE shape (10 frames + 10 one-second spectrogram segments per clip, 224^2), bf16 MFMA compute,
B = 32 clips per GPU, on N MI355X of one node.

    python bench.py --gpus 1 --steps 8 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic clips resident in HBM: forward, soft-target cross-entropy
(the harness's loss, AVE/traintest_adapt_ave29.py:113,159), backward through the frozen backbone with weight gradients for
the 5.6 M trainable parameters, one RCCL all-reduce of the flat gradient arena (N > 1), Adam step with the reference's
hyper-parameters (:68).  Rank 0 prints ONE JSON line; `value` is the whole-job clips/s (all ranks), timed between
barrier + synchronize on both sides and taking the max over ranks.

The line is <= 6144 bytes (LINE_MAX).  Extra objects in it:
  roofline        the dominant rocprof KERNEL of the step aggregated over its (N, K, epilogue) classes: achieved = algorithmic FLOPs (2*M*N*K,
                  no padding) / HIP-event time of its sampled launches on the launch stream, inside this run; peak 2.5 PFLOP/s dense bf16;
                  traffic = PMC HBM bytes per launch (committed rocprofv3 --pmc passes of this same command).
  roofline_class  the largest single class, priced against ITS bound (MFMA above 312 FLOP/B, HBM 8 TB/s below).
  roofline_families  one row per non-GEMM kernel family (algorithmic bytes / HIP-event time).
  roofline_pass   the eager pass those samples come from, and how much of its step they account for.
  cpu_baseline    the oracle (oracle/swin.py, fp32 PyTorch-CPU restatement, "port") timed on this box's host cores on a bounded
                  sample (B=1 clip, fwd+bwd), rank 0 at N=1 only.
The per-class / per-family tables go to gpurun_out/bench_detail.json (`detail` names it) and to a 'BENCH_DETAIL ...' stdout line BEFORE the
JSON line; nothing follows the JSON line on stdout.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GFLOP_PER_CLIP = 1587.5      # fwd+bwd algorithmic GEMM FLOPs per clip, Swin-B AVE fusion
PEAK_BF16_TFLOPS = 2500.0    # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0            # HBM3E spec peak (MI355X_MICROARCH.md; 6.3 TB/s is the measured copy rate)

SWIN_B = dict(label_dim=29, patch_size=[1, 4, 4], num_frames=10, embed_dim=128, depths=[2, 2, 18, 2],
              num_heads=[4, 8, 16, 32], window_size=7, pretrained=None, ftmode="fusion",
              adapter_mlp_ratio=[0.125, 0.125, 0.0625, 0.0625])


SWIN_L = dict(label_dim=29, patch_size=[1, 4, 4], num_frames=10, embed_dim=192, depths=[2, 2, 18, 2],
              num_heads=[6, 12, 24, 48], window_size=7, pretrained=None, ftmode="fusion",
              adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
VIT_B = dict(label_dim=29, layers=12, num_video_frames=10, embed_dim=768, patch_size=16, heads=8, pretrained=None, ftmode="fusion")
# workload -> (fwd+bwd algorithmic GEMM GFLOP per clip, description).  swin_b is the headline metric (BASELINE.json configs[2]
# at N GPUs); vit_b is configs[1] (ViT-B/16 + STG-CMA full stack, AVE shape, 197 video + 49 audio tokens, heads = 8 as the
# reference runner builds it, AVE/run_adapt_ave29.py:130-139), selectable with --workload vit_b.
WORKLOADS = {
    "swin_b": (1587.5, "Swin-B + STG-CMA ftmode=fusion, AVE shape (10 frames + 10 spectrogram segments, 224^2), "
                       "fwd+bwd+Adam on 5.6M adapter/head params"),
    "swin_l": (3989.0, "Swin-L + STG-CMA ftmode=fusion, AVE shape, adapter ratios [.5,.25,.125,.0625] (AVE/run_swin_adapt_ave29.sh:52), "
                       "fwd+bwd+Adam on the adapter/head params"),
    # backbone-only workloads (kept from round 1, when the AVS decoder / AVQA head -- SURVEY 8f -- were not built yet; "avs" / "avqa" below are the
    # full models): fwd + bwd from seeded upstream gradients on every output the reference's head consumes + Adam on the adapters; GFLOP figures
    # are analytic approximations
    "avs_backbone": (843.0, "Swin-B + STG-CMA AVS BACKBONE only (AVS/run_adapt_avs.py:146-160: T=5, adapter ratios [.25,.25,.125,.125]), "
                            "multi-scale taps + audio feature, synthetic upstream gradients, no decoder"),
    "avqa_backbone": (4920.0, "Swin-L + STG-CMA AVQA BACKBONE only (AVQA/run_adapt_avqa.py:288-301: T=10, third negative-video stream "
                              "forward-only), synthetic upstream gradients, no QA head"),
    # 843.0 (backbone) + 590.8 (decoder: FlopCounterMode over oracle/avs_decoder.py, fwd + bwd with weight gradients of avstask_*, TPAVI's affinity in
    # the collapsed form the product computes -- 4 N C_i^2 instead of the reference's explicit 4 N^2 C_i, which would add 399 more; tools/avs_decoder_flops.py)
    "avs": (1433.8, "Swin-B + STG-CMA AVS shape, FULL model (AVS/run_adapt_avs.py:146-160): backbone (T=5) + dense decoder (ASPP, TPAVI, "
                    "FeatureFusion path, output convolutions), loss = BCE on the first frame of each clip (AVS/loss.py:7-26), fwd+bwd+Adam on "
                    "adapters + avstask_*; GFLOP = backbone 843.0 + decoder 590.8 (TPAVI affinity counted collapsed)"),
    "avqa": (4920.0, "Swin-L + STG-CMA AVQA shape, FULL model (AVQA/run_adapt_avqa.py:288-301): backbone with the negative-video stream + "
                     "QA head (question LSTM, grounding, single-query attentions), loss = CE(qa) + 0.5 CE(match) "
                     "(traintest_adapt_avqa.py:173-179), fwd+bwd+Adam on adapters + avqatask_* (no fp8 path)"),
    "vit_b": (1158.7, "ViT-B/16 (CLIP) + STG-CMA ftmode=fusion, AVE shape (10 frames 224^2 + 10 spectrogram segments 102x128), "
                      "fwd+bwd+Adam on the adapter/head params"),
}


def build_model(torch, device, workload="swin_b"):
    import stgcma  # noqa: F401
    from stgcma.recipe import is_trainable
    torch.manual_seed(0)
    if workload == "vit_b":
        from stgcma.model import CLIP_AVE as Cm
        m = Cm.MM_CLIP_AVE(**VIT_B)
    elif workload in ("avs_backbone", "avs"):
        from stgcma.model import Swin_AVSModel
        m = Swin_AVSModel.SwinTransformer2D_Adapter_AVS_Base(patch_size=[1, 4, 4], img_size=224, num_frames=5, embed_dim=128, depths=[2, 2, 18, 2],
                                                   num_heads=[4, 8, 16, 32], window_size=7, pretrained=None, ftmode="fusion",
                                                   adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125])
    elif workload in ("avqa_backbone", "avqa"):
        from stgcma.model import Swin_AVQAModel_V1
        m = Swin_AVQAModel_V1.SwinTransformer2D_Adapter_AVQA(patch_size=[1, 4, 4], img_size=224, num_frames=10, embed_dim=192,
                                                     depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7, pretrained=None,
                                                     ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
    else:
        from stgcma.model import Swin_AVE as S
        m = S.SwinTransformer2D_Adapter_New(**(SWIN_L if workload == "swin_l" else SWIN_B))
    # de-zero what the reference zero-initialises, so no kernel can shortcut zeros
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "W_z.1.weight" in n:                              # TPAVI's BatchNorm scale (zero-initialised, TPAVI.py:62-63)
                p.fill_(0.1)
            elif "D_fc2" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif "gate_" in n:
                p.fill_(0.1)
    for n, p in m.named_parameters():                       # the reference's freeze filter
        p.requires_grad = is_trainable(n)
    return m.to(device).train()


def synth_batch(torch, B, device, rank, workload="swin_b"):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    v = torch.randn((B, 3, 10, 224, 224), generator=g, device=device)
    a = torch.randn((B, 10) + ((102, 128) if workload == "vit_b" else (224, 224)), generator=g, device=device) * 0.5
    cls = torch.randint(0, 29, (B * 10,), generator=g, device=device)
    labels = torch.nn.functional.one_hot(cls, 29).float()   # float one-hot rows, '(b t) c'
    return a, v, labels


CPU_THREADS_CAP = 16     # the oracle's many small ops stop scaling (and oversubscribe a cgroup-limited box) beyond this


def cpu_baseline_measure(torch, model, max_passes=3, budget_s=25.0):
    """The oracle (fp32 PyTorch-CPU restatement) fwd+bwd at B=1 on the host cores: a BOUNDED sample -- one untimed warm-up pass,
    then timed passes until `budget_s` seconds are spent or `max_passes` are done; the MEDIAN pass is reported."""
    import oracle.swin as OS
    from params import seeded_tensor
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(CPU_THREADS_CAP, avail))
    torch.set_num_threads(cores)
    P = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    for n in names:
        P[n].requires_grad_(True)
    cfg = dict(SWIN_B, img_size=224)
    a = seeded_tensor((1, 10, 224, 224), 7, 0.5)
    v = seeded_tensor((1, 3, 10, 224, 224), 8)
    tgt = torch.nn.functional.one_hot(torch.arange(10) % 29, 29).float()
    times = []
    t_begin = time.perf_counter()
    warm = True
    while len(times) < max_passes and (not times or time.perf_counter() - t_begin < budget_s):
        t0 = time.perf_counter()
        logits = OS.swin_forward(P, a, v, cfg, "fusion")
        OS.soft_target_cross_entropy(logits, tgt).backward()
        if warm:
            warm = False                                  # the first pass pays allocator / thread-pool start-up: not timed
        else:
            times.append(time.perf_counter() - t0)
        for n in names:
            P[n].grad = None
    med = sorted(times)[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle/swin.py fp32 fwd+bwd of ONE clip (B=1), median of {len(times)} pass(es) after 1 warm-up within a "
                      f"{budget_s:.0f} s budget, torch {cores} threads (the box's CPU share for one GPU)"}


def cpu_baseline_child():
    """`bench.py --cpu-baseline-child`: runs in a fresh process that never touches the GPU; prints one tagged JSON line."""
    import torch
    model = build_model(torch, "cpu")
    print("CPU_BASELINE " + json.dumps(cpu_baseline_measure(torch, model)), flush=True)


def cpu_baseline(timeout_s=240):
    """Time the oracle in a CHILD process (started, not exec'ed: this process owns the GPU) under a hard timeout, so a slow
    host can never take the bench line down with it."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(CPU_THREADS_CAP), MKL_NUM_THREADS=str(CPU_THREADS_CAP))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    fail = {"value": None, "unit": "clips/s", "cores": CPU_THREADS_CAP, "kind": "port", "sample": "oracle/swin.py fp32 fwd+bwd, B=1"}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], env=env, capture_output=True,
                           text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return dict(fail, error=f"timed out after {timeout_s} s")
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    return dict(fail, error=(r.stderr or "no output")[-300:])


def self_launch(n):
    """`python bench.py --gpus N` typed without a launcher (no WORLD_SIZE in the environment): start the N ranks as CHILD processes through
    torch.distributed.run -- the launch line the module docstring gives -- relay their output (rank 0's JSON line) and return their exit
    code.  This parent imports no torch and makes no HIP call before or after: a child process, never an exec of a GPU-initialised one
    (what replaces nn.DataParallel's in-process replication, AVE/traintest_adapt_ave29.py:32-35)."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


PMC_BY_CLASS = ("r06b_final_pmc_by_class.json", "r06_final_pmc_by_class.json", "r05_final_pmc_by_class.json", "r04_final_pmc_by_class.json", "r03_pmc_by_class.json")      # tools/ledger.py: PMC bytes per launch per GEMM class, joined by launch order


def pmc_traffic(kernel, N, K, epi):
    """HBM bytes per launch of THIS class (kernel x N x K x epilogue; at the bench's row count) from the committed rocprofv3 PMC
    passes of this same command: (2 * FETCH_SIZE + WRITE_SIZE) KiB per dispatch, the gfx950 correction of MI355X_MICROARCH.md,
    dispatches matched to classes by launch order (bench.py --gemm-seq + tools/ledger.py -> profiles/r03_pmc_by_class.json).  PMC
    counters cannot be read from inside the process, so the line quotes the last committed pass; None when the class is absent."""
    for cand in PMC_BY_CLASS:
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        c = d.get("classes", {}).get(f"{kernel}|{N}|{K}|{epi}")
        if c is not None:
            PMC_MFMA[(kernel, N, K, epi)] = c.get("mfma_util_pmc")
            meta = d.get("meta") or {}
            where = f"; counters taken on commit {meta['commit']} with `{meta.get('command', '?')}`" if meta.get("commit") else "; counters of an earlier tree (file without a commit record)"
            return int(c["hbm_bytes_per_launch"]), f"profiles/{cand} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {c['launches_per_step']} launches of this class per step{where})"
    return None, None


PMC_MFMA = {}        # (kernel, N, K, epilogue) -> rocprof MFMA utilisation of the class's dispatches (tools/ledger.py), filled by pmc_traffic


def pmc_step():
    """Whole-step HBM traffic and the GEMM / non-GEMM split of the kernel time from the same committed passes (None when absent)."""
    for cand in PMC_BY_CLASS:
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                return dict(json.load(f)["step"], source=f"profiles/{cand}")
        except (OSError, ValueError, KeyError):
            continue
    return None


LINE_MAX = 6144        # bytes of the final stdout line: the driver keeps an 8 KB tail and parses its last line (BENCH_r04 lost a 28 KB one)
MFMA_FAMILIES = ("xattn_fwd", "xattn_bwd", "mlp_fused_fwd", "mlp_fused_bwd", "mha_fwd", "mha_bwd", "dec_conv_wgrad", "dec_tpavi_bmm")
METRIC_NAMES = {"swin_b": "Swin-B+STG-CMA AVE-shape", "swin_l": "Swin-L+STG-CMA AVE-shape", "vit_b": "ViT-B/16+STG-CMA AVE-shape",
                "avs_backbone": "Swin-B+STG-CMA AVS-shape backbone", "avqa_backbone": "Swin-L+STG-CMA AVQA-shape backbone",
                "avqa": "Swin-L+STG-CMA AVQA-shape (backbone + QA head)", "avs": "Swin-B+STG-CMA AVS-shape (backbone + dense decoder)"}


def build_report(gp, fams, ctx):
    """(line, detail) from the pass-1 samples.  `gp` = kernels.gemm_profile_stop() rows (one per GEMM class = kernel the C dispatch chose x
    N x K x epilogue), `fams` = kernels.family_profile_stop() rows (one per non-GEMM family x shape key), `ctx` = the run's scalars.  Pure
    host arithmetic (tests/test_bench_line_cpu.py feeds it synthetic profiles).

    line["roofline"]        the dominant rocprof KERNEL aggregated over its classes (the object profiles/*_kernel_stats.csv names): achieved
                            = algorithmic FLOPs (2 M N K, no padding) / HIP-event time of its sampled launches; traffic = PMC bytes per launch
                            averaged over its classes' dispatches (committed rocprofv3 --pmc passes of this same command).
    line["roofline_class"]  its / the step's largest single class, priced against ITS bound (MFMA above 312 FLOP/B, else HBM).
    line["roofline_families"]  one row per non-GEMM family.  Per-class and per-family-key tables -> detail."""
    batch, world, n_steps, prof_steps = ctx["batch"], ctx["world"], ctx["n_steps"], max(ctx["prof_steps"], 1)
    value = batch * world * n_steps / ctx["dt"]
    eager_ms = ctx["dt_eager"] / ctx["steps_arg"] * 1e3 if ctx.get("dt_eager") is not None else None
    RIDGE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_TBS * 1e12)
    classes = []
    for pc in gp:
        if pc["sampled"] == 0 or pc["sampled_ms"] <= 0:
            continue
        ns = pc["sampled"]
        avg_us = pc["sampled_ms"] * 1e3 / ns
        sec = pc["sampled_ms"] * 1e-3
        intensity = pc["sampled_flops"] / max(pc["sampled_bytes"], 1.0)
        tf, tbs = pc["sampled_flops"] / sec / 1e12, pc["sampled_bytes"] / sec / 1e12
        bound = "mfma" if intensity > RIDGE else "hbm"
        traffic, traffic_src = pmc_traffic(pc["kernel"], pc["N"], pc["K"], pc["epi"])
        classes.append({"bound": bound, "kernel": pc["kernel"], "N": pc["N"], "K": pc["K"], "epilogue": pc["epi"],
                        "achieved": round(tf if bound == "mfma" else tbs * 1e3, 2), "peak": PEAK_BF16_TFLOPS if bound == "mfma" else PEAK_HBM_TBS * 1e3,
                        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                        "frac": round(tf / PEAK_BF16_TFLOPS if bound == "mfma" else tbs / PEAK_HBM_TBS, 4),
                        "tflops": round(tf, 1), "gbs": round(tbs * 1e3, 0), "flop_per_byte": round(intensity, 1),
                        "launches_per_step": round(pc["launches"] / prof_steps, 2), "avg_launch_us": round(avg_us, 2),
                        "gflop_per_launch": round(pc["sampled_flops"] / ns / 1e9, 2), "algorithmic_bytes_per_launch": round(pc["sampled_bytes"] / ns),
                        "sampled_launches": ns, "est_ms_per_step": round(avg_us * 1e-3 * pc["launches"] / prof_steps, 3),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "mfma_util_pmc": PMC_MFMA.get((pc["kernel"], pc["N"], pc["K"], pc["epi"]))})
    classes.sort(key=lambda c: -c["est_ms_per_step"])
    gemm_ms = round(sum(c["est_ms_per_step"] for c in classes), 2)
    # the other half of the step, the same way: bound HBM except the families whose floor is the matrix pipe / VALU issue
    fam_rows = []
    for fc in fams:
        if fc["sampled"] == 0 or fc["sampled_ms"] <= 0:
            continue
        ns, sec = fc["sampled"], fc["sampled_ms"] * 1e-3
        avg_us = fc["sampled_ms"] * 1e3 / ns
        tf, tbs = fc["sampled_flops"] / sec / 1e12, fc["sampled_bytes"] / sec / 1e12
        bound = "mfma" if fc["family"] in MFMA_FAMILIES else "hbm"
        fam_rows.append({"family": fc["family"], "key": fc["key"], "bound": bound,
                         "achieved": round(tf if bound == "mfma" else tbs * 1e3, 2), "peak": PEAK_BF16_TFLOPS if bound == "mfma" else PEAK_HBM_TBS * 1e3,
                         "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(tf / PEAK_BF16_TFLOPS if bound == "mfma" else tbs / PEAK_HBM_TBS, 4),
                         "gbs": round(tbs * 1e3), "tflops": round(tf, 1), "launches_per_step": round(fc["launches"] / prof_steps, 2),
                         "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": round(fc["sampled_bytes"] / ns),
                         "sampled_launches": ns, "est_ms_per_step": round(avg_us * 1e-3 * fc["launches"] / prof_steps, 3)})
    fam_rows.sort(key=lambda r: -r["est_ms_per_step"])
    fam_ms = round(sum(r["est_ms_per_step"] for r in fam_rows), 2)
    fam_tot = {}
    for r in fam_rows:
        t = fam_tot.setdefault(r["family"], {"family": r["family"], "bound": r["bound"], "ms": 0.0, "_b": 0.0, "_f": 0.0})
        t["ms"] += r["est_ms_per_step"]
        t["_b"] += r["algorithmic_bytes_per_launch"] * r["launches_per_step"]
        t["_f"] += r["tflops"] * r["est_ms_per_step"]
    fam_summary = []
    for t in sorted(fam_tot.values(), key=lambda t: -t["ms"]):
        ms = max(t["ms"], 1e-9)
        gbs, tfl = t["_b"] / ms / 1e6, t["_f"] / ms
        fam_summary.append({"family": t["family"], "bound": t["bound"], "ms": round(ms, 2), "gbs": round(gbs), "tflops": round(tfl, 1),
                            "frac": round(tfl / PEAK_BF16_TFLOPS if t["bound"] == "mfma" else gbs / (PEAK_HBM_TBS * 1e3), 3)})
    # the dominant kernel, aggregated over its classes
    by_kernel = {}
    for c in classes:
        k = by_kernel.setdefault(c["kernel"], {"kernel": c["kernel"], "ms": 0.0, "gflop": 0.0, "gb": 0.0, "launches": 0.0, "pmc_b": 0.0, "pmc_l": 0.0, "src": None})
        k["ms"] += c["est_ms_per_step"]
        k["gflop"] += c["gflop_per_launch"] * c["launches_per_step"]
        k["gb"] += c["algorithmic_bytes_per_launch"] * c["launches_per_step"] / 1e9
        k["launches"] += c["launches_per_step"]
        if c["traffic"] is not None:
            k["pmc_b"] += c["traffic"] * c["launches_per_step"]
            k["pmc_l"] += c["launches_per_step"]
            src = c["traffic_source"] or ""
            k["src"] = src.split(" ")[0] + (" (" + src.split("; ", 1)[1] if "; " in src else "")     # file + which commit / command its counters are from
    roofline, roofline_class = None, None
    if by_kernel:
        k = max(by_kernel.values(), key=lambda k: k["ms"])
        ms, nl = max(k["ms"], 1e-9), max(k["launches"], 1e-9)
        tf, gbs = k["gflop"] / ms, k["gb"] / ms * 1e3
        bound = "mfma" if k["gflop"] / max(k["gb"], 1e-9) > RIDGE else "hbm"
        roofline = {"kernel": k["kernel"], "bound": bound, "achieved": round(tf if bound == "mfma" else gbs, 1),
                    "peak": PEAK_BF16_TFLOPS if bound == "mfma" else PEAK_HBM_TBS * 1e3, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                    "frac": round(tf / PEAK_BF16_TFLOPS if bound == "mfma" else gbs / (PEAK_HBM_TBS * 1e3), 4),
                    "traffic": round(k["pmc_b"] / k["pmc_l"]) if k["pmc_l"] else None,
                    "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE+WRITE_SIZE KiB, rocprofv3 --pmc), mean over this kernel's dispatches",
                    "traffic_source": k["src"], "algorithmic_bytes_per_launch": round(k["gb"] * 1e9 / nl),
                    "gflop_per_launch": round(k["gflop"] / nl, 2), "avg_launch_us": round(ms * 1e3 / nl, 2), "launches_per_step": round(nl, 1),
                    "est_ms_per_step": round(ms, 3), "tflops": round(tf, 1), "gbs": round(gbs),
                    "what": "dominant kernel of the step, all its (N, K, epilogue) classes: algorithmic FLOPs / HIP-event time in this run"}
        top = classes[0]
        roofline_class = {kk: top[kk] for kk in ("kernel", "N", "K", "epilogue", "bound", "achieved", "peak", "unit", "frac", "tflops", "gbs", "avg_launch_us",
                                                 "launches_per_step", "algorithmic_bytes_per_launch", "traffic", "mfma_util_pmc", "est_ms_per_step")}
    st = pmc_step()
    if st is not None:
        st = {kk: st[kk] for kk in ("hbm_gb_per_step", "step_kernel_ms", "launches", "gemm_ms", "non_gemm_ms", "gemm_gb", "mfma_util_pmc",
                                    "mfma_util_pmc_gemm_kernels", "source") if kk in st}
    line = {
        "metric": ("SECONDARY (bf16 residual stream, not the headline dataflow) " if ctx["residual"] == "bf16" else "") + "clips/sec fwd+bwd, " + METRIC_NAMES[ctx["workload"]],
        "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": n_steps, "warmup": ctx["warmup"],
        "ms_per_step": round(ctx["dt"] / n_steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8-e4m3 (frozen weights + their inputs, E8M0 block scales) / bf16" if ctx["fp8"] else "bf16", "data": "synthetic",
        "config": {"workload": ctx["workload_desc"], "clips_per_gpu": batch, "global_batch": batch * world, "parallelism": f"dp{world}",
                   "residual_dtype": ctx["residual"], "step": ctx["step_how"], "step_forms_measured": ctx["pick"],
                   "side_streams_overlap": ctx["overlap"]},
        "model_tflops": round(value * ctx["gflop_per_clip"] / 1e3, 2),
        "mfma_frac_whole_step": round(value * ctx["gflop_per_clip"] / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
        "final_loss": round(ctx["final_loss"], 4),
        "roofline": roofline, "roofline_class": roofline_class,
        "gemm_est_ms_per_step": gemm_ms, "non_gemm_est_ms_per_step": round(eager_ms - gemm_ms, 2) if eager_ms else None,
        "roofline_families": fam_summary, "family_est_ms_per_step": fam_ms, "step_traffic": st,
        "roofline_pass": {
            "what": "eager single-stream pass of this run (HIP events around sampled launches): "
                    + (f"{ctx['steps_arg']} timed steps" if world == 1 else f"{prof_steps} warm-up step(s)")
                    + "; `value` steps replay HIP graphs (no events inside)",
            "ms_per_step": round(eager_ms, 3) if eager_ms else None,
            "value": round(batch * world / (eager_ms * 1e-3), 3) if eager_ms else None, "unit": "clips/s",
            "accounted_ms_per_step": round(gemm_ms + fam_ms, 2),
            "accounted_frac": round((gemm_ms + fam_ms) / eager_ms, 4) if eager_ms else None},
    }
    detail = {"roofline_classes": classes, "roofline_family_classes": fam_rows, "options": ctx.get("options"),
              "headline": {kk: line[kk] for kk in ("metric", "value", "unit", "n_gpus", "ms_per_step", "config")}}
    return line, detail


def write_detail(detail):
    """Per-class / per-family tables -> gpurun_out/bench_detail.json (merged back by gpurun) and one 'BENCH_DETAIL ' stdout line that does
    not start with '{', printed BEFORE the JSON line.  Returns the path (or None when the directory is not writable)."""
    print("BENCH_DETAIL " + json.dumps(detail), flush=True)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_detail.json")
        with open(path, "w") as f:
            json.dump(detail, f)
        return "gpurun_out/bench_detail.json"
    except OSError:
        return None


def finalize_line(line):
    """json.dumps(line) within LINE_MAX bytes: drop the least important objects, in order, until it fits (never value / roofline / cpu_baseline)."""
    s = json.dumps(line)
    for victim in ("step_traffic", "roofline_class", "roofline_families", "roofline_pass"):
        if len(s) <= LINE_MAX:
            break
        if victim == "roofline_families" and line.get(victim):
            line[victim] = line[victim][:8]                 # the largest eight first, then drop entirely
            s = json.dumps(line)
            if len(s) <= LINE_MAX:
                break
        line[victim] = None
        s = json.dumps(line)
    if len(s) > LINE_MAX:
        line["config"]["workload"] = line["config"]["workload"][:120]
        line["config"]["step"] = str(line["config"]["step"])[:120]
        s = json.dumps(line)
    assert len(s) <= LINE_MAX, len(s)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="swin_b",
                    help="swin_b = the headline metric (default); vit_b = BASELINE configs[1]")
    ap.add_argument("--fp8", action="store_true",
                    help="frozen backbone Linears on the block-scaled e4m3 MFMA path (BASELINE config 5; opt-in, Swin workloads); the "
                         "line then carries dtype 'fp8-e4m3(frozen weights + their inputs)/bf16' and is NOT the headline metric")
    ap.add_argument("--graph", action="store_true",
                    help="after the eager measurement (which stays `value`), also time the identical step replayed from a HIP graph (N = 1); "
                         "opt-in: a fault inside capture / replay must not be able to take the headline line with it")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)      # accepted for older command lines: the default now
    ap.add_argument("--gemm-seq", default=None, metavar="PATH",
                    help="write the launch-ordered GEMM class list of ONE step (kernel, M, N, K, epilogue, algorithmic bytes) as JSON: "
                         "tools/ledger.py joins it with a rocprofv3 trace of the same command")
    ap.add_argument("--microbatch", type=int, default=2,
                    help="micro-batches per step, each a HIP graph on its own stream, running concurrently (default 2; 1 = one launch chain)")
    ap.add_argument("--eager", action="store_true", help="time eager single-stream steps only (no HIP graphs, no micro-batch streams): the round-3 form")
    ap.add_argument("--ddp-one-graph", action="store_true",
                    help="N > 1: capture the RCCL all-reduce inside the graph too (default: two graphs with the collective eager between them)")
    ap.add_argument("--residual", choices=("fp32", "bf16"), default="fp32",
                    help="residual-stream dtype.  fp32 (default) is what the reference's autocast loop keeps; bf16 is a LABELLED SECONDARY "
                         "measurement (config.residual_dtype says so), never the headline")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    import torch
    import torch.distributed as dist
    import stgcma  # noqa: F401
    from stgcma import ddp, kernels
    if args.residual == "bf16":
        stgcma.configure(residual="bf16")

    # STG_DDP_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices;
    # RCCL refuses two ranks on one device).  The driver's runs use the default, RCCL.
    rank, local_rank, world = ddp.init_from_env(os.environ.get("STG_DDP_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    if os.environ.get("STG_DDP_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if os.environ.get("STG_BENCH_POISON_GB"):
        # robustness rehearsal (tests / tools only): fill that many GB of the caching allocator with NaN patterns and hand them back, so that
        # every later torch.empty starts as NaN -- a kernel that reads a byte nobody wrote shows up as a non-finite loss
        n = int(float(os.environ["STG_BENCH_POISON_GB"]) * 2**30 / 4)
        junk = [torch.full((n // 8,), float("nan"), device=device) for _ in range(8)]
        torch.cuda.synchronize()
        del junk

    model = build_model(torch, device, args.workload)

    trace_nan = bool(os.environ.get("STG_BENCH_TRACE_NAN"))

    def probe(where):                  # debugging aid (STG_BENCH_TRACE_NAN=1): where do non-finite parameters / gradients first appear?
        if not trace_nan:
            return
        torch.cuda.synchronize()
        nb_p = sum(1 for p_ in model.parameters() if p_.requires_grad and not bool(torch.isfinite(p_).all()))
        nb_g = sum(1 for p_ in model.parameters() if p_.grad is not None and not bool(torch.isfinite(p_.grad).all()))
        print(f"bench.py[rank {rank}] probe {where}: non-finite params {nb_p}, grads {nb_g}", file=sys.stderr, flush=True)

    gflop_per_clip, workload_desc = WORKLOADS[args.workload]
    if args.fp8:
        from stgcma import fp8 as stg_fp8
        if args.workload == "vit_b":
            raise SystemExit("--fp8 covers the Swin workloads")
        stg_fp8.enable(model)
        workload_desc += "; frozen qkv / proj / fc1 / fc2 / reduction GEMMs (forward + dgrad) on block-scaled e4m3 MFMA"
    sync = None
    if world > 1:
        ddp.broadcast_parameters(model)
        sync = ddp.attach(model)
    from stgcma import recipe
    opt = recipe.build_optimizer(model, lr=1e-4, head_lr=0.1, capturable=(world == 1))   # reference recipe: Adam(0.95, 0.999), wd 5e-7, two groups
    loss_fn = torch.nn.CrossEntropyLoss()
    a, v, labels = synth_batch(torch, args.batch, device, rank, args.workload)

    labels2 = labels.reshape(-1, labels.shape[-1])

    def fwd_bwd():                                                  # traintest_adapt_ave29.py:136-163 (recipe.train_step without the optimizer step)
        loss = loss_fn(model(a, v, "fusion"), labels2)
        opt.zero_grad()
        loss.backward()
        return loss

    if args.workload == "avqa":
        g = torch.Generator(device=device).manual_seed(77 + rank)
        vv = torch.randn((args.batch, 10, 3, 224, 224), generator=g, device=device)
        vn = torch.randn((args.batch, 10, 3, 224, 224), generator=g, device=device)
        aa = torch.randn((args.batch, 10, 224, 224), generator=g, device=device) * 0.5
        qq = torch.randint(0, 93, (args.batch, 14), generator=g, device=device)
        ans = torch.randint(0, 42, (args.batch,), generator=g, device=device)
        match = torch.tensor([1, 0] * (args.batch * 10), device=device)              # batch_organize: posi / nega interleaved

        def fwd_bwd():                                              # noqa: F811   traintest_adapt_avqa.py:168-185
            out_qa, mp, mn = model(aa, vv, vn, qq, "fusion")
            loss = loss_fn(out_qa, ans) + 0.5 * loss_fn(torch.stack((mp, mn), dim=1).reshape(-1, 2), match)
            opt.zero_grad()
            loss.backward()
            return loss

    if args.workload == "avs":
        g = torch.Generator(device=device).manual_seed(55 + rank)
        vv = torch.randn((args.batch, 5, 3, 224, 224), generator=g, device=device)
        aa = torch.randn((args.batch, 5, 224, 224), generator=g, device=device) * 0.5
        gt = (torch.rand((args.batch, 1, 224, 224), generator=g, device=device) < 0.3).float()
        bce = torch.nn.BCELoss()

        def fwd_bwd():                                              # noqa: F811   traintest_adapt_avs.py:158-170, AVS/loss.py:7-26
            pred, _, _ = model(aa, vv, "fusion")
            loss = bce(torch.sigmoid(pred)[::5], gt)
            opt.zero_grad()
            loss.backward()
            return loss

    if args.workload in ("avs_backbone", "avqa_backbone"):
        T = 5 if args.workload == "avs_backbone" else 10
        g = torch.Generator(device=device).manual_seed(99 + rank)
        vv = torch.randn((args.batch, T, 3, 224, 224), generator=g, device=device)          # 'b t c h w' (AVS :1793 / AVQA :1742)
        aa = torch.randn((args.batch, T, 224, 224), generator=g, device=device) * 0.5
        vn = torch.randn((args.batch, T, 3, 224, 224), generator=g, device=device) if args.workload == "avqa_backbone" else None
        ups = {}

        def fwd_bwd():                                              # noqa: F811
            if vn is None:
                ms, a_feat = model.forward_features(aa, vv)
                outs = list(ms) + [a_feat]
            else:
                outs = [o for o in model.forward_features(aa, vv, vn) if o.requires_grad]
            loss = 0.
            for i, o in enumerate(outs):
                if i not in ups:
                    ups[i] = torch.randn(o.shape, generator=g, device=device) * 1e-3
                loss = loss + (o * ups[i]).sum()
            opt.zero_grad()
            loss.backward()
            return loss

    def step():
        loss = fwd_bwd()
        if trace_nan:
            probe(f"eager step: after backward (loss {float(loss.detach()):.4f})")
        opt.step()
        if trace_nan:
            probe("eager step: after optimizer")
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # micro-batches are exact where no operation looks across the clips of a batch: the AVE models and the AVQA model (LayerNorm, per-clip
    # attention, mean-reduced losses) -- not the AVS decoder (BatchNorm over the batch: two half batches are not the same function; it replays
    # capture_train_step_ddp's graphs).  Round 6: under DDP the AVQA task head's gradients ride the micro-batch form too (its join graph packs them
    # into one bucket), so configs 3 and 5 share one N > 1 step form.
    MB_OK = ("swin_b", "swin_l", "vit_b", "avqa")
    nmb = args.microbatch if (args.workload in MB_OK and args.batch % max(args.microbatch, 1) == 0 and not args.fp8) else 1
    labels3 = labels.view(args.batch, -1, labels.shape[-1])
    mb_tensors = (a, v, labels3)

    def fwd_loss(a_, v_, y_):                                       # one micro-batch: forward + the harness's loss
        return loss_fn(model(a_, v_, "fusion"), y_.reshape(-1, y_.shape[-1]))

    if args.workload == "avqa":
        mb_tensors = (aa, vv, vn, qq, ans, match.view(args.batch, -1))

        def fwd_loss(aa_, vv_, vn_, qq_, ans_, match_):             # noqa: F811   traintest_adapt_avqa.py:168-179 on a chunk of clips
            out_qa, mp, mn = model(aa_, vv_, vn_, qq_, "fusion")
            return loss_fn(out_qa, ans_) + 0.5 * loss_fn(torch.stack((mp, mn), dim=1).reshape(-1, 2), match_.reshape(-1))

    # ---------------------------------------------------------------------------------------------------------------- pass 1: eager
    # One stream, one launch per kernel (the round-3 form): the pass the per-class / per-family roofline samples come from (HIP events
    # around individual launches need eager launches).  At N = 1 it is timed as a whole too (`eager_single_stream`); at N > 1 it is
    # the warm-up (its first step, which learns the kernel of every call signature, is not sampled).
    kernels.gemm_profile_start(log_sequence=args.gemm_seq is not None)   # learns, during the warm-up, which kernel the C dispatch picks per call signature
    kernels.family_profile_start()
    for i in range(args.warmup):
        step()
        if world > 1 and i == 0:
            kernels.gemm_profile_reset()
            kernels.family_profile_reset()
    fence()
    probe("after eager warm-up")
    prof_steps = max(args.warmup - 1, 1)
    dt_eager, loss = None, None
    if world == 1:
        kernels.gemm_profile_reset()
        kernels.family_profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        dt_eager = time.perf_counter() - t0
        prof_steps = args.steps
    if args.gemm_seq is not None and rank == 0:
        seq = kernels.gemm_profile_sequence()
        per = len(seq) // max(prof_steps, 1)
        with open(args.gemm_seq, "w") as f:
            json.dump({"steps": prof_steps, "launches_per_step": per, "step": seq[-per:] if per else []}, f)
    gp, fams = kernels.gemm_profile_stop(), kernels.family_profile_stop()

    def emit(dt, n_steps, step_how, pick, overlap, final_loss):
        """Rank 0: build and print THE JSON line (everything sampled in pass 1 is closed over; the timed-region results are arguments, so the
        N > 1 watchdog can print a line from the eager measurement when the replayed form never comes back).  The line is <= LINE_MAX bytes;
        the per-class / per-family tables go to gpurun_out/bench_detail.json (named in the line) and to a stdout line that starts with
        'BENCH_DETAIL ' BEFORE the JSON line.  Nothing follows the JSON line on stdout."""
        if rank != 0:
            return
        ctx = {"batch": args.batch, "world": world, "n_steps": n_steps, "warmup": args.warmup, "steps_arg": args.steps, "dt": dt,
               "dt_eager": dt_eager, "prof_steps": prof_steps, "workload": args.workload, "workload_desc": workload_desc,
               "gflop_per_clip": gflop_per_clip, "fp8": args.fp8, "residual": args.residual, "step_how": step_how, "pick": pick,
               "overlap": overlap, "final_loss": final_loss, "options": stgcma.options() if hasattr(stgcma, "options") else None}
        line, detail = build_report(gp, fams, ctx)
        try:
            if world == 1 and not args.no_cpu_baseline and args.workload == "swin_b":
                line["cpu_baseline"] = cpu_baseline()
        finally:
            line["detail"] = write_detail(detail)
            print(finalize_line(line), flush=True)

    # ---------------------------------------------------------------------------------------------------------------- pass 2: the timed steps
    # The product's step (recipe.capture_train_step_mb): `nmb` micro-batches, each forward + backward a HIP graph on its own stream,
    # running CONCURRENTLY (two independent launch chains fill each other's ramps and tails: DESIGN.md section 5.3), then a join graph
    # (gradient sum + Adam).  N > 1: the summed arena is all-reduced eagerly between the join graph and the optimizer graph, so a rank's
    # host issues ~4 graph launches + 1 collective per step instead of ~1 200 kernel launches.  --microbatch 1: N = 1 keeps the eager
    # pass as the measurement; N > 1 replays recipe.capture_train_step_ddp's two graphs.  Every rank must take the same path: the ranks
    # agree on "captured" before the first replay and fall back to eager steps together.
    replay, static_loss, step_how = None, None, "eager, one stream"
    # a workload that cannot be split into micro-batches (the AVS model: BatchNorm over the batch) replays ONE graph of the whole step at N = 1 --
    # no second stream to overlap with, but no launch gaps either; the eager pass stays the measurement if that graph is not faster on the box
    whole_step_graph = world == 1 and nmb == 1 and args.microbatch > 1 and not args.fp8
    want_graph = (nmb > 1 or world > 1 or whole_step_graph) and not args.eager

    def timed2(fn):
        fence()
        t0_ = time.perf_counter()
        for _ in range(2):
            fn()
        fence()
        tt = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt) / 2 * 1e3

    # N > 1: the eager form (with its collectives) is timed over two steps BEFORE anything is captured -- the comparison the form selection
    # below needs, and what a WATCHDOG falls back on: RCCL has never run under the replayed form on this pool's hardware, and a collective or
    # a replay that never comes back would otherwise cost the whole line.  If pass 2 has not finished after STG_BENCH_WATCHDOG_S seconds
    # (default 300), rank 0 prints the line from that eager measurement and every rank leaves.
    ms_e_pre, watchdog = None, None
    if want_graph and world > 1:
        import threading
        ms_e_pre = timed2(step)
        probe("after 2 eager steps (form selection)")
        last_eager_loss = float(step().detach())
        limit_s = float(os.environ.get("STG_BENCH_WATCHDOG_S", "300"))

        def fire():
            # a hang is a FAILURE: rank 0 still prints the line from the eager measurement (so the number is not lost), then every rank
            # leaves with exit code 4 (not 3: gpurun's own 3 means "no box free, nothing ran" and tools/grun.sh retries on it -- ADVICE r5).  The timer thread is NOT a daemon, so the interpreter waits for it and os._exit decides the code
            try:
                if rank == 0 and last_eager_loss == last_eager_loss and abs(last_eager_loss) != float("inf"):
                    emit(ms_e_pre * 2 / 1e3, 2, f"eager, one stream (WATCHDOG: the replayed form did not finish within {limit_s:.0f} s; value = the two eager steps "
                         "timed before the capture)", {"eager_ms_per_step": round(ms_e_pre, 3), "replay_ms_per_step": None}, None, last_eager_loss)
                else:
                    time.sleep(5.0)           # the launcher tears every rank down when the first one fails: let rank 0 print first
            finally:
                os._exit(4)
        watchdog = threading.Timer(limit_s, fire)
        watchdog.daemon = False
        watchdog.start()
    # the eager pass's last loss, DETACHED (its autograd graph must not survive into the capture): reported whenever the timed region falls back to dt_eager
    eager_loss = loss.detach().clone() if loss is not None else None
    try:
        if want_graph:
            loss = None
            import gc
            gc.collect()                  # the last eager step's autograd graph holds AccumulateGrad nodes bound to the default stream
            ok, why = 1, ""
            try:
                if os.environ.get("STG_BENCH_FAIL_CAPTURE"):                 # tests only: rehearse the capture-failure fallback
                    raise RuntimeError("capture failure forced by STG_BENCH_FAIL_CAPTURE")
                if nmb > 1:
                    replay, static_loss, step_how = recipe.capture_train_step_mb(fwd_loss, mb_tensors, opt, splits=nmb, sync=sync, warmup=1)
                elif world == 1:
                    replay, static_loss, step_how = recipe.capture_train_step_ddp(fwd_bwd, opt, None, warmup=1, collective_in_graph=True)
                    step_how = "one graph of the whole step (forward + backward + Adam), one stream"
                else:
                    replay, static_loss, step_how = recipe.capture_train_step_ddp(fwd_bwd, opt, sync, warmup=1, collective_in_graph=args.ddp_one_graph)
            except Exception as e:                                       # noqa: BLE001  (any capture failure -> eager, on every rank)
                ok, why = 0, repr(e)[:200]
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag)
            if not ok:
                if replay is not None and hasattr(replay, "release"):
                    replay.release()
                replay, step_how = None, f"eager, one stream (graph capture failed{': ' + why if why else ' on another rank'})"
            else:
                probe("after capture")
                for _ in range(2):
                    replay()
                    probe("after a warm replay")
            fence()
        # N > 1 has no timed eager pass to compare with: time two steps of each form (max over ranks) and keep the faster for the timed region,
        # so that a box on which the replayed form loses (streams sharing a hardware queue, a collective that serialises behind a graph) is
        # measured on the form that wins there.  Every rank takes the same decision (it is made on all-reduced times).
        pick = None
        if replay is not None and world > 1:
            ms_e = ms_e_pre
            ms_r = timed2(replay)
            probe("after 2 replays (form selection)")
            pick = {"eager_ms_per_step": round(ms_e, 3), "replay_ms_per_step": round(ms_r, 3)}
            if ms_r > ms_e:
                if hasattr(replay, "release"):
                    replay.release()
                replay = None
                step_how = f"eager, one stream (the replayed form measured {ms_r:.1f} ms per step against {ms_e:.1f} eager on this box)"
        if replay is not None or world > 1:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                if replay is not None:
                    replay()
                    loss = static_loss
                else:
                    loss = step()
            fence()
            dt = time.perf_counter() - t0
            probe("after the timed steps")
            if world == 1 and dt_eager is not None and dt > dt_eager:
                # the replayed micro-batch form lost to the eager pass on this box (both are K timed steps of the same workload): the line
                # reports the faster one as `value` and says so
                pick = {"eager_ms_per_step": round(dt_eager / args.steps * 1e3, 3), "replay_ms_per_step": round(dt / args.steps * 1e3, 3)}
                step_how = f"eager, one stream (the replayed micro-batch form measured {dt / args.steps * 1e3:.1f} ms per step: slower on this box)"
                dt, loss = dt_eager, eager_loss
        else:
            dt, loss = dt_eager, eager_loss                              # N = 1 without graphs (or capture failed): the eager pass IS the measurement
    finally:                                     # an exception in pass 2 must surface at once with its own exit code, not after the watchdog's limit
        if watchdog is not None:
            watchdog.cancel()
    final_loss = float(loss.detach())
    if final_loss != final_loss or final_loss in (float("inf"), float("-inf")):
        # say where the non-finite values are before giving up (stderr; the line itself is never printed for an invalid run)
        try:
            bad_p = [n for n, p_ in model.named_parameters() if not bool(torch.isfinite(p_).all())]
            bad_g = [n for n, p_ in model.named_parameters() if p_.grad is not None and not bool(torch.isfinite(p_.grad).all())]
            print(f"bench.py[rank {rank}]: step form '{step_how}', pick {pick}; non-finite parameters: {len(bad_p)} {bad_p[:4]}; non-finite gradients: "
                  f"{len(bad_g)} {bad_g[:4]}; inputs finite: {[bool(torch.isfinite(t_.float()).all()) for t_ in mb_tensors] if mb_tensors else None}",
                  file=sys.stderr, flush=True)
        except Exception as e_:                                      # noqa: BLE001
            print(f"bench.py[rank {rank}]: diagnostics failed: {e_!r}", file=sys.stderr, flush=True)
        # a step that produced NaN / inf is not a measurement (and NaN-filled tensors toggle fewer bits, so the power-limited chip clocks
        # every kernel HIGHER: such a run looks faster -- DESIGN.md 5.2)
        raise SystemExit(f"bench.py: non-finite loss {final_loss} after {args.warmup + args.steps} steps -- the run is invalid")
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)

    emit(dt, args.steps, step_how, pick, getattr(replay, "streams_overlap", None) if replay is not None else None, final_loss)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
