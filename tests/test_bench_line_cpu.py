"""bench.py's final stdout line: <= 6144 bytes whatever the profile looks like (the driver keeps an 8 KB tail and parses the last line;
BENCH_r04 lost a 28 KB line), with `roofline` (dominant kernel over its classes) and room for `cpu_baseline`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def synthetic_profile(n_classes=60, n_fams=60):
    gp, fams = [], []
    kernels = ["gemm_nt_8ph_kernel", "gemm_nt_8phm_kernel", "gemm_nt_128_kernel", "gemm_nt_small_kernel"]
    for i in range(n_classes):
        M, N, K = 125440 * (1 + i % 4), 512 * (1 + i % 8), 512 * (1 + i % 3)
        gp.append({"kernel": kernels[i % 4], "N": N, "K": K, "epi": ["none", "bap8", "d8", "res_gate_bias"][i % 4], "launches": 18 * 8, "sampled": 20,
                   "sampled_ms": 20 * (0.05 + 0.01 * i), "sampled_flops": 20 * 2.0 * M * N * K, "sampled_bytes": 20 * 2.0 * (M * K + N * K + M * N)})
    names = ["upln_fwd", "winattn_bwd", "ln_bwd_down", "wgrad", "winattn_fwd", "tattn_fwd", "tattn_bwd", "xattn_fwd", "xattn_bwd", "mlp_fused_fwd",
             "mlp_fused_bwd", "ln_fwd", "ln_bwd", "elementwise", "mha_fwd", "mha_bwd", "im2col", "head", "adam", "dec_conv"]
    for i in range(n_fams):
        fams.append({"family": names[i % len(names)], "key": f"C{128 << (i % 4)}_rows{2007040 >> (i % 4)}_a_rather_long_shape_key_{i}", "launches": 36 * 8, "sampled": 12,
                     "sampled_ms": 12 * (0.02 + 0.005 * i), "sampled_flops": 12 * 1e9 * (i + 1), "sampled_bytes": 12 * 5e8 * (i + 1)})
    return gp, fams


def ctx(**kw):
    c = {"batch": 32, "world": 1, "n_steps": 20, "warmup": 5, "steps_arg": 20, "dt": 2.26, "dt_eager": 2.39, "prof_steps": 20, "workload": "swin_b",
         "workload_desc": bench.WORKLOADS["swin_b"][1], "gflop_per_clip": 1587.5, "fp8": False, "residual": "fp32",
         "step_how": "2 micro-batch graphs on 2 streams + join", "pick": None, "overlap": True, "final_loss": 3.3712,
         "options": {f"opt{i}": True for i in range(40)}}
    c.update(kw)
    return c


def test_line_fits_and_carries_roofline_and_cpu_baseline():
    gp, fams = synthetic_profile()
    line, detail = bench.build_report(gp, fams, ctx())
    line["cpu_baseline"] = {"value": 0.2888, "unit": "clips/s", "cores": 16, "kind": "port", "sample": "x" * 220}
    line["detail"] = "gpurun_out/bench_detail.json"
    s = bench.finalize_line(line)
    assert len(s) <= bench.LINE_MAX == 6144 and "\n" not in s
    d = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] == round(r["achieved"] / r["peak"], 4) and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert d["cpu_baseline"]["cores"] == 16 and d["cpu_baseline"]["kind"] == "port"
    assert abs(d["value"] - 32 * 20 / 2.26) < 1e-2 and d["vs_baseline"] is None
    # the tables the line no longer carries are in the detail object
    assert len(detail["roofline_classes"]) == 60 and len(detail["roofline_family_classes"]) == 60 and detail["options"]


def test_line_fits_with_a_pathological_profile_and_long_step_description():
    gp, fams = synthetic_profile(200, 400)
    for i, f in enumerate(fams):
        f["family"] = f"family_with_a_long_name_{i}"            # 400 distinct families: the summary must be cut, not the headline
    line, _ = bench.build_report(gp, fams, ctx(world=8, dt_eager=None, step_how="eager, one stream (" + "y" * 900 + ")",
                                               pick={"eager_ms_per_step": 119.1, "replay_ms_per_step": 113.2}))
    line["cpu_baseline"] = None
    s = bench.finalize_line(line)
    assert len(s) <= 6144
    d = json.loads(s)
    assert d["roofline"]["frac"] > 0 and d["n_gpus"] == 8 and d["config"]["global_batch"] == 256


def test_roofline_is_the_dominant_kernel_aggregated_over_its_classes():
    gp = [{"kernel": "gemm_nt_8ph_kernel", "N": 512, "K": 2048, "epi": "none", "launches": 10, "sampled": 10, "sampled_ms": 2.0,
           "sampled_flops": 10 * 2e11, "sampled_bytes": 10 * 4e8},
          {"kernel": "gemm_nt_8ph_kernel", "N": 512, "K": 512, "epi": "none", "launches": 10, "sampled": 10, "sampled_ms": 1.0,
           "sampled_flops": 10 * 1e11, "sampled_bytes": 10 * 3e8},
          {"kernel": "gemm_nt_8phm_kernel", "N": 2048, "K": 512, "epi": "bap8", "launches": 10, "sampled": 10, "sampled_ms": 2.5,
           "sampled_flops": 10 * 2.6e11, "sampled_bytes": 10 * 9e8}]
    line, _ = bench.build_report(gp, [], ctx(prof_steps=1))
    r = line["roofline"]
    assert r["kernel"] == "gemm_nt_8ph_kernel" and r["launches_per_step"] == 20.0
    assert abs(r["achieved"] - (10 * 2e11 + 10 * 1e11) / 3.0e-3 / 1e12) < 0.5 and r["bound"] == "mfma"
    assert line["roofline_class"]["kernel"] == "gemm_nt_8phm_kernel"       # the largest single class keeps its own object
