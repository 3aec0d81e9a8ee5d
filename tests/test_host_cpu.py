"""CPU (-m "not gpu"): host-side logic of the product package -- C-ABI library loads and exports every declared symbol,
nn.Module surface / state_dict contract vs the reference's structure, parameter partition, addressing maps vs the oracle,
error behaviour, and the no-CPU-fallback guarantee.  No kernel is launched here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from golden_util import GOLD, ROOT

import oracle.swin as OS


def test_library_loads_and_exports_every_declared_symbol(stg):
    from stgcma import _lib
    lib = _lib.lib()
    assert lib.stg_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "stgcma.h")).read()
    declared = set(re.findall(r"\b(stg_[a-z0-9_]+)\s*\(", header))
    declared -= {"stg_attn_bwd_prep"}          # mentioned in a comment only
    assert declared, "no declarations parsed"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/stgcma.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == declared


def test_error_reporting_without_a_gpu(stg):
    """Argument validation happens on the host before any launch: a NULL argument struct is rejected with a message."""
    from stgcma import _lib
    lib = _lib.lib()
    rc = lib.stg_gemm_nt(None, None)
    assert rc < 0 and b"null" in lib.stg_last_error()
    a = _lib.GemmArgs()
    a.A, a.W, a.C = 16, 16, 16
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = 4, 8, 12, 12, 12, 8     # K not a multiple of 8
    rc = lib.stg_gemm_nt(ctypes.byref(a), None)
    assert rc < 0 and b"multiple of 8" in lib.stg_last_error()


@pytest.mark.parametrize("tag,ctor", [
    ("swin_b_fusion", dict(label_dim=29, patch_size=[1, 4, 4], num_frames=10, embed_dim=128, depths=[2, 2, 18, 2],
                           num_heads=[4, 8, 16, 32], window_size=7, pretrained=None, ftmode="fusion",
                           adapter_mlp_ratio=[.125, .125, .0625, .0625])),
    ("swin_l_fusion", dict(label_dim=29, patch_size=[1, 4, 4], img_size=224, num_frames=10, embed_dim=192, depths=[2, 2, 18, 2],
                           num_heads=[6, 12, 24, 48], window_size=7, pretrained=None, ftmode="fusion",
                           adapter_mlp_ratio=[.5, .25, .125, .0625])),
])
def test_state_dict_contract_matches_reference(stg, tag, ctor):
    """Keys, shapes, dtypes AND order of state_dict() equal the reference's; so do total / trainable / head counts."""
    from stgcma.model import Swin_AVE as S
    from stgcma import recipe
    with open(os.path.join(GOLD, "structure.json")) as f:
        ref = json.load(f)[tag]
    m = S.SwinTransformer2D_Adapter_New(**ctor)
    mine = [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in m.state_dict().items()]
    assert mine == [tuple(x) if False else (x[0], x[1], x[2]) for x in ref["keys"]]
    adapt, head = recipe.apply_freeze(m)
    assert sum(p.numel() for p in m.parameters()) == ref["n_total"]
    assert sum(p.numel() for p in adapt) + sum(p.numel() for p in head) == ref["n_trainable"]
    assert sum(p.numel() for p in head) == ref["n_head"]
    # frozen by name: both patch embeddings and the final norm
    d = dict(m.named_parameters())
    for n in ("patch_embed.proj.weight", "patch_embed_audio.proj.weight", "norm.weight", "layers.0.blocks.0.attn.qkv.weight",
              "layers.0.blocks.0.attn.relative_position_bias_table"):
        assert not d[n].requires_grad
    for n in ("layers.0.blocks.0.attn.temporal_position_bias_table_audio", "layers.2.blocks.5.gate_v",
              "layers.3.blocks.1.S_Adapter2_Audio.D_fc1.bias", "mlp_head.2.bias"):
        assert d[n].requires_grad


def test_zero_init_of_adapters_and_gates(stg):
    """D_fc2 of every adapter and both gates start at zero (Swin_AVE.py:1422-1468, :365-366)."""
    from stgcma.model import Swin_AVE as S
    m = S.SwinTransformer2D_Adapter_New(label_dim=5, embed_dim=32, depths=[2, 2], num_heads=[1, 2], num_frames=2,
                                        ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25])
    for n, p in m.named_parameters():
        if "D_fc2" in n or "gate_" in n:
            assert float(p.abs().max()) == 0, n
        if n.endswith("norm1.weight"):
            assert torch.equal(p, torch.ones_like(p))


def test_ctor_and_forward_error_behaviour(stg):
    from stgcma.model import Swin_AVE as S
    with pytest.raises(TypeError, match="ftmode is not expected"):
        S.SwinTransformer2D_Adapter_New(label_dim=5, ftmode="bogus")
    with pytest.raises(TypeError, match="pretrained must be a str or None"):
        S.SwinTransformer2D_Adapter_New(label_dim=5, embed_dim=32, depths=[2], num_heads=[1], pretrained=3, ftmode="fusion",
                                        adapter_mlp_ratio=[0.5])
    m = S.SwinTransformer2D_Adapter_New(label_dim=5, embed_dim=32, depths=[2, 2], num_heads=[1, 2], num_frames=2,
                                        ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25])
    with pytest.raises(TypeError):
        m(torch.zeros(1, 2, 224, 224), torch.zeros(1, 3, 2, 224, 224), "videoonly")     # mode != ftmode
    # the product path has no CPU fallback: CPU tensors are refused, loudly
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 224, 224), torch.zeros(1, 3, 2, 224, 224), "fusion")


def test_missing_extension_fails_loudly(stg, monkeypatch):
    from stgcma import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libstgcma_hip.so")
    with pytest.raises(_lib.StgLibraryMissing):
        _lib.lib()


@pytest.mark.parametrize("H,ws,shift", [(56, 7, 0), (56, 7, 3), (28, 7, 3), (14, 7, 3), (7, 7, 0)])
def test_addressing_maps_equal_oracle(stg, H, ws, shift):
    """The maps the attention kernels address through == the oracle's restatement of roll + window_partition, and the
    shift mask == the reference's attn_mask (pinned in the golden fixtures through the oracle test)."""
    from stgcma import ops
    m = ops.window_token_map(H, H, ws, shift)
    assert torch.equal(m.long().view(-1, ws * ws), OS.window_token_map(H, H, ws, shift))
    assert m.sort().values.tolist() == list(range(H * H))          # a permutation: scatter back hits every token once
    if shift:
        assert torch.equal(ops.shift_mask(H, H, ws, shift), OS.shift_attn_mask(H, H, ws, shift))
    T = 10
    tm = ops.temporal_token_map(H * H, T).view(H * H, T)
    n = torch.arange(H * H)
    for t in (0, 3, 9):
        assert torch.equal(tm[:, t].long(), t * H * H + n)


def test_recipe_partition_matches_reference_filter(stg):
    from stgcma import recipe
    assert recipe.is_trainable("module.layers.0.blocks.0.T_Adapter.D_fc1.weight")
    assert recipe.is_trainable("layers.1.blocks.0.attn.temporal_position_bias_table")
    assert recipe.is_trainable("mlp_head.0.weight") and recipe.is_trainable("transformer.resblocks.3.gate_a")
    assert recipe.is_trainable("ln_post.weight") and recipe.is_trainable("temporal_embedding_audio")
    for frozen in ("patch_embed_audio.proj.weight", "conv1_audio.weight", "positional_embedding_audio", "norm.bias",
                   "layers.0.blocks.0.attn.relative_position_bias_table", "layers.0.blocks.0.mlp.fc1.weight"):
        assert not recipe.is_trainable(frozen), frozen


def test_grad_arena_layout(stg):
    from stgcma import ops
    P = {"a": torch.zeros(3, 5), "b": torch.zeros(7), "c": torch.zeros(2, 2)}
    need = {"a": True, "b": False, "c": True}
    ar = ops.GradArena(["a", "b", "c"], P, need, torch.device("cpu"))
    va, vc = ar.view("a", P["a"]), ar.view("c", P["c"])
    assert va.shape == (3, 5) and vc.shape == (2, 2) and ar.flat.numel() == 16 + 4
    va.fill_(1.0); vc.fill_(2.0)
    assert float(ar.flat.sum()) == 15 + 8
    assert va.data_ptr() % 16 == ar.flat.data_ptr() % 16 and (vc.data_ptr() - ar.flat.data_ptr()) % 16 == 0


def test_bench_cpu_baseline_leg_runs_on_host(stg):
    """bench.py's cpu_baseline leg (the oracle timed on host cores) must work from a model's state_dict, integer buffers
    included -- it only ever runs on the GPU box otherwise."""
    import bench
    from stgcma.model import Swin_AVE as S
    torch.manual_seed(0)
    m = S.SwinTransformer2D_Adapter_New(**bench.SWIN_B)
    from stgcma.recipe import is_trainable
    for n, p in m.named_parameters():
        p.requires_grad = is_trainable(n)
    r = bench.cpu_baseline_measure(torch, m, max_passes=1)
    assert r["kind"] == "port" and r["unit"] == "clips/s" and r["value"] > 0 and r["cores"] >= 1


@pytest.mark.parametrize("tag,mod,cls", [("avs_tiny_backbone", "Swin_AVSModel", "SwinTransformer2D_Adapter_AVS"),
                                         ("avqa_tiny_backbone", "Swin_AVQAModel_V1", "SwinTransformer2D_Adapter_AVQA")])
def test_avs_avqa_backbone_state_dict_is_the_reference_subset(stg, tag, mod, cls):
    """The AVS / AVQA mirrors hold exactly the backbone tensors of the reference classes (patch embeds, layers, norm) under
    the reference's names and shapes; the AVS mirror also carries the dense decoder (avstask_*), the AVQA mirror the QA head (avqatask_*)."""
    import importlib
    from golden_util import load_case
    z, cfg, shapes, names = load_case(tag)
    M = importlib.import_module("stgcma.model." + mod)
    m = getattr(M, cls)(pretrained=None, num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                        num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"])
    sd = m.state_dict()
    backbone = ("patch_embed.", "patch_embed_audio.", "layers.", "norm.")
    mine = [(k, tuple(v.shape)) for k, v in sd.items() if v.is_floating_point() and not k.endswith("attn_mask") and k.startswith(backbone)]
    assert mine == [(k, tuple(s)) for k, s in shapes]
    from stgcma.recipe import is_trainable
    assert [n for n, _ in m.named_parameters() if is_trainable(n) and n.startswith(backbone)] == names
    rest = [k for k in sd if not k.startswith(backbone)]
    pre = "avstask_" if mod == "Swin_AVSModel" else "avqatask_"   # the dense decoder / the QA head under the reference's names
    assert rest and all(k.startswith(pre) for k in rest)


def test_avs_full_state_dict_is_the_reference(stg):
    """backbone + dense decoder: float keys (BatchNorm running statistics included), order and shapes of the mirror == the
    reference's SwinTransformer2D_Adapter_AVS_Base (golden avs_full_tiny); the AVS loop's name filter (traintest_adapt_avs.py:55)
    selects the same trainable tensors."""
    from golden_util import load_case
    from stgcma.model import Swin_AVSModel
    from stgcma.recipe import is_trainable
    z, cfg, shapes, names = load_case("avs_full_tiny")
    m = Swin_AVSModel.SwinTransformer2D_Adapter_AVS_Base(pretrained=None, num_frames=5, embed_dim=cfg["embed_dim"], depths=cfg["depths"],
                                                    num_heads=cfg["num_heads"], ftmode="fusion", adapter_mlp_ratio=cfg["adapter_mlp_ratio"],
                                                    channel=256, opt=None, config=None, vis_dim=[64, 128, 320, 512],
                                                    tpavi_stages=[0, 1, 2, 3], tpavi_vv_flag=False, tpavi_va_flag=True)
    sd = m.state_dict()
    mine = [(k, tuple(v.shape)) for k, v in sd.items() if v.is_floating_point() and not k.endswith("attn_mask")]
    assert mine == [(k, tuple(s)) for k, s in shapes]
    assert [n for n, _ in m.named_parameters() if is_trainable(n)] == names
    for i in range(4):                                       # reference init: TPAVI's BatchNorm scale and shift start at zero
        bn = getattr(m, f"avstask_tpavi_b{i + 1}").W_z[1]
        assert float(bn.weight.abs().max()) == 0 and float(bn.bias.abs().max()) == 0
    with pytest.raises(TypeError, match="ftmode is not expected"):
        m(None, None, "videoonly")


def test_avqa_full_state_dict_is_the_reference(stg):
    """backbone + QA head: float keys, order and shapes of the mirror == the reference model's (golden avqa_full_tiny), and the
    AVQA loop's name filter (traintest_adapt_avqa.py:72) selects the same trainable tensors."""
    from golden_util import load_case
    from stgcma.model import Swin_AVQAModel_V1
    from stgcma.recipe import is_trainable
    z, cfg, shapes, names = load_case("avqa_full_tiny")
    m = Swin_AVQAModel_V1.SwinTransformer2D_Adapter_AVQA(pretrained=None, num_frames=cfg["num_frames"], embed_dim=cfg["embed_dim"],
                                                 depths=cfg["depths"], num_heads=cfg["num_heads"], ftmode="fusion",
                                                 adapter_mlp_ratio=cfg["adapter_mlp_ratio"])
    sd = m.state_dict()
    mine = [(k, tuple(v.shape)) for k, v in sd.items() if v.is_floating_point() and not k.endswith("attn_mask")]
    assert mine == [(k, tuple(s)) for k, s in shapes]
    assert [n for n, _ in m.named_parameters() if is_trainable(n)] == names
    with pytest.raises(TypeError, match="ftmode is not expected"):
        m(None, None, None, None, "videoonly")


def test_recipe_cosine_scheduler_and_optimizer_groups(stg):
    """the product's LR tables equal the reference's (golden from utilities/scheduler.py) and the optimizer has the
    reference's two groups."""
    import os
    from golden_util import GOLD
    from stgcma import recipe
    from stgcma.model import Swin_AVE as S
    z = np.load(os.path.join(GOLD, "cosine_scheduler.npz"))
    np.testing.assert_allclose(recipe.cosine_scheduler(5e-5, 2e-6, 20, 3339, warmup_epochs=2), z["t1"], rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(recipe.cosine_scheduler(1e-4, 2e-6, 3, 7, warmup_epochs=1), z["t3"], rtol=1e-12, atol=1e-18)
    m = S.SwinTransformer2D_Adapter_New(label_dim=29, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], num_frames=2,
                                        ftmode="fusion", adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
    opt = recipe.build_optimizer(m, lr=1e-4, head_lr=10.0)
    g0, g1 = opt.param_groups
    assert g0["lr"] == 1e-4 and abs(g1["lr"] - 1e-3) < 1e-12 and g0["betas"] == (0.95, 0.999) and g0["weight_decay"] == 5e-7
    assert sum(p.numel() for p in g1["params"]) == sum(p.numel() for n, p in m.named_parameters() if n.startswith("mlp_head."))
    assert all(p.requires_grad for g in opt.param_groups for p in g["params"])
    assert not m.layers[0].blocks[0].attn.qkv.weight.requires_grad


def test_configure_is_the_one_switchboard():
    """every Python-side switch goes through stgcma.configure(); the module attributes the hot path reads follow it."""
    import importlib
    import pytest
    import torch
    import stgcma
    ops = importlib.import_module("stg-cma_amd.ops")
    ops_vit = importlib.import_module("stg-cma_amd.ops_vit")
    o0 = stgcma.options()
    assert o0["residual"] == "float32" and o0["upln"] is True and o0["fp8"] is False
    try:
        stgcma.configure(upln=False, residual="bf16", mlp_dact="bf16")
        assert ops.USE_UPLN is False and ops.RESIDUAL_DTYPE == torch.bfloat16 and ops_vit.RESIDUAL_DTYPE == torch.bfloat16 and ops.MLP_DACT is True
        assert stgcma.options()["residual"] == "bfloat16"
        with pytest.raises(KeyError):
            stgcma.configure(no_such_switch=1)
    finally:
        stgcma.configure(upln=True, residual="fp32", mlp_dact="u8")
    assert ops.USE_UPLN is True and ops.RESIDUAL_DTYPE == torch.float32 and ops.MLP_DACT == "u8"


def test_bench_self_launch_builds_the_documented_command(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE: a CHILD process through torch.distributed.run on 127.0.0.1 with the same arguments; the
    parent relays its exit code (and has not imported torch)."""
    import subprocess
    import sys
    import pytest
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(4)
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
