"""The drop-in boundary the way a user of the reference meets it: a fresh interpreter, a foreign working
directory laid out like the reference's task folders (a `model/` directory WITHOUT __init__.py next to the runner), PYTHONPATH
exactly as documented, and the import lines of the reference runners -- AVE/run_adapt_ave29.py:12,156,
AVS/run_adapt_avs.py:14-16,146-185, AVQA/run_adapt_avqa.py:18-20,288-301, AVQA/test.py:8.  No GPU: only imports and constructors."""
import os
import subprocess
import sys
import textwrap

from golden_util import ROOT

PKG = os.path.join(ROOT, "stg-cma_amd")


def _run(code, cwd, pythonpath):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = pythonpath
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + "\n" + r.stderr
    return r.stdout


def _fake_task_dir(tmp_path):
    d = tmp_path / "AVX"
    (d / "model").mkdir(parents=True)                    # the reference's model/ folders carry no __init__.py
    (d / "model" / "Swin_AVE.py").write_text("raise ImportError('the reference file must be shadowed by the drop-in package')\n")
    return str(d)


def test_documented_pythonpath_imports_and_constructs(tmp_path):
    out = _run("""
        import model                                              # AVE/run_adapt_ave29.py:12
        import models                                             # AVS/run_adapt_avs.py:14 (dead import of the runners)
        import model.Swin_AVSModel as AVSModel                    # AVS/run_adapt_avs.py:15
        import model.Swin_AVSModel_Base as AVSModelBase           # AVS/run_adapt_avs.py:16
        import model.Swin_AVQAModel_V1 as AVQAModel               # AVQA/run_adapt_avqa.py:20
        import model.Swin_AVQAModel as AVQAModel512               # AVQA/test.py:8
        m = model.Swin_AVE.SwinTransformer2D_Adapter_New(label_dim=29, patch_size=[1, 4, 4], num_frames=10, embed_dim=32,
                depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=7, pretrained=None, ftmode='fusion',
                adapter_mlp_ratio=[0.125, 0.125, 0.0625, 0.0625])
        c = model.CLIP_AVE.MM_CLIP_AVE(label_dim=29, num_video_frames=10, audio_length=1024, layers=2, heads=8, embed_dim=768,
                patch_size=16, input_resolution=224, pretrained=None, ftmode='fusion', drop_path_rate=0.2, num_tadapter=1,
                adapter_scale=0.5)
        kw = dict(patch_size=[1, 4, 4], img_size=224, num_frames=5, depths=[2, 2, 2, 2], window_size=7, pretrained=None,
                  ftmode='fusion', channel=256, opt=None, config=None, vis_dim=[64, 128, 320, 512], tpavi_stages=[0, 1, 2, 3],
                  tpavi_vv_flag=False, tpavi_va_flag=True)
        b = AVSModelBase.SwinTransformer2D_Adapter_AVS_Base(embed_dim=32, num_heads=[1, 2, 4, 8],
                adapter_mlp_ratio=[0.25, 0.25, 0.125, 0.125], **kw)
        l = AVSModel.SwinTransformer2D_Adapter_AVS(embed_dim=48, num_heads=[1, 2, 4, 8],
                adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625], **kw)
        q = AVQAModel.SwinTransformer2D_Adapter_AVQA(patch_size=[1, 4, 4], img_size=224, num_frames=10, embed_dim=192,
                depths=[2, 2, 2, 2], num_heads=[6, 12, 24, 48], window_size=7, pretrained=None, grounding_pretrained=None,
                ftmode='fusion', adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
        q5 = AVQAModel512.SwinTransformer2D_Adapter_AVQA(patch_size=[1, 4, 4], img_size=224, num_frames=10, embed_dim=192,
                depths=[2, 2, 2, 2], num_heads=[6, 12, 24, 48], window_size=7, pretrained=None, grounding_pretrained=None,
                ftmode='fusion', adapter_mlp_ratio=[0.5, 0.25, 0.125, 0.0625])
        assert q.avqatask_fc_ans.in_features == 1536 and q5.avqatask_fc_ans.in_features == 512
        assert q5.avqatask_yb_fc_a.out_features == 128 and not hasattr(q, 'avqatask_yb_fc_v')
        # one class object per class, whichever way it is imported
        import stgcma
        from stgcma.model import Swin_AVE, Swin_AVSModel_Base
        assert Swin_AVE is model.Swin_AVE and Swin_AVSModel_Base is AVSModelBase
        assert type(m).__module__.endswith('model.Swin_AVE')
        # the reference loop's name filter finds the task heads (traintest_adapt_avs.py:55, _avqa.py:72)
        assert any(n.startswith('avstask_') for n, _ in b.named_parameters())
        assert any(n.startswith('avqatask_') for n, _ in q.named_parameters())
        # without a GPU the forward must fail loudly, not fall back
        import torch
        try:
            m(torch.zeros(1, 10, 224, 224), torch.zeros(1, 3, 10, 224, 224), 'fusion')
        except RuntimeError as e:
            assert 'MI355X' in str(e) or 'GPU' in str(e), e
        else:
            raise AssertionError('CPU forward did not raise')
        print('DROPIN-OK')
        """, _fake_task_dir(tmp_path), PKG + os.pathsep + ROOT)
    assert "DROPIN-OK" in out


def test_package_dir_alone_on_pythonpath(tmp_path):
    """Only stg-cma_amd/ on PYTHONPATH (the repository root is found from the package's own location)."""
    out = _run("""
        import model.Swin_AVQAModel_V1 as AVQAModel
        import model
        assert model.Swin_AVQAModel_V1 is AVQAModel
        print(AVQAModel.SwinTransformer2D_Adapter_AVQA.__name__)
        """, _fake_task_dir(tmp_path), PKG)
    assert "SwinTransformer2D_Adapter_AVQA" in out
