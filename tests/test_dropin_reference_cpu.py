"""``dropin.convert()`` on the REFERENCE's own BDModel / DepthModel instances (build container only: needs
/root/reference, which never travels to the GPU box — skipped there).  The reference is imported with the same stub
modules the golden generator uses; no kernel runs (CPU)."""
import contextlib
import io
import os
import sys

import pytest
import torch

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is only present in the build container")


@pytest.fixture(scope="module")
def ref_models():
    # utils/generic_utils.py TorchScript-compiles a helper that references a stubbed kornia symbol at import time;
    # torch is already imported here, so PYTORCH_JIT=0 (what the golden generator sets) comes too late
    old_script = torch.jit.script
    torch.jit.script = lambda fn=None, *a, **kw: fn
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_golden as g

    g.import_reference()
    import implicit_depth_amd.synthetic as syn
    import antialiased_cnns
    import timm

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
    old_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **kw: self  # BDModel(use_prior) calls .cuda() in its ctor
    from experiment_modules.bd_model import BDModel
    from experiment_modules.depth_model import DepthModel
    from options import Options

    def make(cls, fvt, K, decoder="unet_pp", use_prior=False):
        o = Options()
        o.image_width, o.image_height = 128, 96
        o.matching_num_depth_bins = 16
        o.feature_volume_type = fvt
        o.model_num_views = K + 1
        o.binary_loss_positive_weight = 1.0
        o.bd_edge_regularision = False
        o.use_prior = use_prior
        o.depth_decoder_name = decoder
        with contextlib.redirect_stdout(io.StringIO()):
            m = cls(o)
        syn.fill_state_dict(m, seed=5)
        return m.eval()

    yield {"make": make, "BDModel": BDModel, "DepthModel": DepthModel}  # DepthModel's ctor scripts a kornia-based helper too
    torch.jit.script = old_script
    torch.nn.Module.cuda = old_cuda
    del torch.nn.Module.save_hyperparameters


@pytest.mark.parametrize("cfg", [("BDModel", "mlp_feature_volume", 7, "unet_pp", False), ("BDModel", "simple_cost_volume", 2, "skip", False),
                                 ("BDModel", "mlp_feature_volume", 7, "unet_pp", True), ("DepthModel", "mlp_feature_volume", 2, "unet_pp", False)])
def test_convert_reference_model_keeps_state_dict(ref_models, cfg):
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import dropin
    from implicit_depth_amd import networks as net

    cls, fvt, K, decoder, prior = cfg
    try:
        m = ref_models["make"](ref_models[cls], fvt, K, decoder, prior)
    except AttributeError as e:  # an Options field this reference revision lacks for that model
        pytest.skip(f"reference ctor: {e}")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ref_types = {n: type(getattr(m, n)).__name__ for n in ("cost_volume", "cost_volume_net", "depth_decoder")}
    dropin.convert(m)
    after = m.state_dict()
    assert list(after) == list(before)  # same keys, same order: checkpoints load unchanged
    for k in before:
        assert torch.equal(after[k], before[k]), k
    assert isinstance(m.cost_volume, cv.CostVolumeManager) and type(m.cost_volume).__module__.startswith("implicit")
    assert isinstance(m.cost_volume_net, net.CVEncoder)
    assert isinstance(m.depth_decoder, (net._DecoderPP, net.SkipDecoder))
    assert ref_types["cost_volume"] in ("FeatureVolumeManager", "CostVolumeManager")
    if cls == "BDModel":
        assert isinstance(m.binary_mlp, net.BinaryMLPNetwork) and m.binary_mlp.use_prior == prior
    dropin.convert(m)  # idempotent
    assert list(m.state_dict()) == list(before)
    hot = dropin.hot_path_of(m)
    assert hot.matching_model is m.matching_model  # the reference encoder: its net[5] / net[8] are what the head plan reads
    assert (hot.min_depth, hot.max_depth) == (m.run_opts.min_matching_depth, m.run_opts.max_matching_depth)
