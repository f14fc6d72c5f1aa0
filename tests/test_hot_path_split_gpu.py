"""End-to-end hot path with the split-precision conv kernels against the REFERENCE goldens
(G5: BDModel.forward dot / mlp, G9: DepthModel.forward) — same tolerances as the fp32-MFMA path
(tests/test_bdmodel_gpu.py), plus a direct comparison of the two arithmetic modes at bench shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from implicit_depth_amd import nhwc

import test_bdmodel_gpu as base


@pytest.fixture(params=["bf16x6", "f16x3"])
def split_default(request):
    old, oldmin = nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS
    nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS = request.param, 1
    yield request.param
    nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS = old, oldmin


@pytest.mark.parametrize("volume", ["dot", "mlp"])
def test_bdmodel_golden_with_split_convs(volume, split_default):
    base.test_hot_path_reproduces_reference_bdmodel_forward(volume)


def test_depthmodel_golden_with_split_convs(split_default):
    base.test_hot_path_reproduces_reference_depthmodel_forward()


@pytest.mark.parametrize("math", ["bf16x6", "f16x3"])
def test_split_hot_path_matches_fp32_hot_path_at_bench_shape(math):
    """512x384 frames, K=7 MLP feature volume, D=64 (the bench workload at B=2): logits of the
    split-precision plan vs the fp32-MFMA plan."""
    import argparse

    from implicit_depth_amd.pipeline import HotPathWorkload

    outs = {}
    for m in ("fp32", math):
        args = argparse.Namespace(batch=2, views=7, planes=64, height=384, width=512, volume="mlp", conv_math=m)
        wl = HotPathWorkload(args, torch.device("cuda:0"), 0)
        wl.step()
        torch.cuda.synchronize()
        plan = next(iter(wl.model._plans.values()))["plan"]
        codes = {op.tile_m for op in plan.ops if op.kind == nhwc.OP_CONV}
        assert (nhwc.SPLIT_CODE[math] in codes) == (m != "fp32")
        outs[m] = wl.out["pred_0"].clone()
    a, b = outs["fp32"], outs[math]
    assert float((a - b).abs().max() / a.abs().max()) < 2e-5
