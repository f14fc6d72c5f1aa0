"""``dropin.convert()`` on the REFERENCE's own BDModel / DepthModel instances (build container only: needs
/root/reference, which never travels to the GPU box — skipped there).  The reference is imported with the same stub
modules the golden generator uses; no kernel runs (CPU).  The checks run in a child interpreter started with
PYTORCH_JIT=0: the reference's geometry helpers are jit.ScriptModules that reference stubbed kornia symbols, and
TorchScript can only be switched off before ``import torch``."""
import contextlib
import io
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CONFIGS = [("BDModel", "mlp_feature_volume", 7, "unet_pp", False), ("BDModel", "simple_cost_volume", 2, "skip", False),
           ("BDModel", "mlp_feature_volume", 7, "unet_pp", True), ("DepthModel", "mlp_feature_volume", 2, "unet_pp", False)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is only present in the build container")
def test_convert_reference_models_keeps_state_dicts():
    env = dict(os.environ, PYTORCH_JIT="0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("converted ok") == len(CONFIGS), r.stdout
    # the reference's own constructors reject matching_scale != 1 (IndexError in the encoder / decoder channel lists,
    # bd_model.py:75-83) and 9 / 12 source views or 32 matching channels construct and convert: the config surface the drop-ins
    # must (and need not) cover
    assert r.stdout.count("reference rejects matching_scale") == 2, r.stdout
    assert r.stdout.count("wide config ok") == 3, r.stdout


def _child():
    import torch

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_golden as g

    g.import_reference()
    import implicit_depth_amd.synthetic as syn
    import antialiased_cnns
    import timm
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import dropin
    from implicit_depth_amd import networks as net

    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        g._stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    k = sys.modules["kornia"]
    k.filters.sobel = None
    for name in ("losses", "geometry", "utils"):
        setattr(k, name, g._stub("kornia." + name))
    timm.create_model = lambda *a, **kw: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **kw: syn.StubResnetStem())
    torch.nn.Module.save_hyperparameters = lambda self, *a, **kw: None
    torch.nn.Module.cuda = lambda self, *a, **kw: self  # BDModel(use_prior) calls .cuda() in its ctor
    from experiment_modules.bd_model import BDModel
    from experiment_modules.depth_model import DepthModel
    from options import Options

    for cls, fvt, K, decoder, prior in CONFIGS:
        o = Options()
        o.image_width, o.image_height = 128, 96
        o.matching_num_depth_bins = 16
        o.feature_volume_type = fvt
        o.model_num_views = K + 1
        o.binary_loss_positive_weight = 1.0
        o.bd_edge_regularision = False
        o.use_prior = prior
        o.depth_decoder_name = decoder
        with contextlib.redirect_stdout(io.StringIO()):
            m = {"BDModel": BDModel, "DepthModel": DepthModel}[cls](o)
        syn.fill_state_dict(m, seed=5)
        m.eval()
        before = {k: v.clone() for k, v in m.state_dict().items()}
        ref_cv = type(m.cost_volume).__module__
        assert ref_cv.startswith("modules."), ref_cv  # really the reference's class
        dropin.convert(m)
        after = m.state_dict()
        assert list(after) == list(before)  # same keys, same order: checkpoints load unchanged
        for k in before:
            assert torch.equal(after[k], before[k]), k
        assert isinstance(m.cost_volume, cv.CostVolumeManager) and isinstance(m.cost_volume_net, net.CVEncoder)
        assert isinstance(m.depth_decoder, (net._DecoderPP, net.SkipDecoder))
        assert (decoder == "skip") == isinstance(m.depth_decoder, net.SkipDecoder)
        if cls == "BDModel":
            assert isinstance(m.binary_mlp, net.BinaryMLPNetwork) and m.binary_mlp.use_prior == prior
        dropin.convert(m)  # idempotent
        assert list(m.state_dict()) == list(before)
        hot = dropin.hot_path_of(m)
        assert hot.matching_model is m.matching_model  # the reference encoder: its net[5] / net[8] are what the head plan reads
        assert (hot.min_depth, hot.max_depth) == (m.run_opts.min_matching_depth, m.run_opts.max_matching_depth)
        print("converted ok", cls, fvt, K, decoder, prior, len(before), "tensors")

    for ms in (0, 2):
        o = Options()
        o.image_width, o.image_height, o.matching_num_depth_bins, o.binary_loss_positive_weight, o.matching_scale = 128, 96, 16, 1.0, ms
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                BDModel(o)
        except IndexError as e:
            print("reference rejects matching_scale", ms, "->", repr(e))
    for K, C in ((9, 16), (12, 16), (7, 32)):
        o = Options()
        o.image_width, o.image_height, o.matching_num_depth_bins, o.binary_loss_positive_weight = 128, 96, 16, 1.0
        o.feature_volume_type, o.model_num_views, o.matching_feature_dims, o.bd_edge_regularision = "mlp_feature_volume", K + 1, C, False
        with contextlib.redirect_stdout(io.StringIO()):
            m = DepthModel(o)  # the model that wires num_source_views / matching_dim_size through (depth_model.py:206-212)
        syn.fill_state_dict(m, seed=6)
        before = {k: v.clone() for k, v in m.state_dict().items()}
        dropin.convert(m)
        assert isinstance(m.cost_volume, cv.FeatureVolumeManager)
        assert (m.cost_volume.num_source_views, m.cost_volume.matching_dim_size) == (K, C), (m.cost_volume.num_source_views, m.cost_volume.matching_dim_size)
        assert list(m.state_dict()) == list(before) and all(torch.equal(m.state_dict()[k], before[k]) for k in before)
        print("wide config ok", K, C)


if __name__ == "__main__":
    _child()
