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


# ---- every conv-stage golden / oracle case of tests/test_conv_gpu.py and the pipeline cases of
# ---- tests/test_pipeline_gpu.py, re-run with the split-precision kernels selected wherever eligible
import test_conv_gpu as conv_base
import test_pipeline_gpu as pipe_base


@pytest.fixture(params=["bf16x6", "f16x3"])
def split_everything(request):
    from implicit_depth_amd import cost_volume as cvmod

    old = nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS, cvmod.DEFAULT_MLP_MATH
    nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS = request.param, 1
    cvmod.DEFAULT_MLP_MATH = "f16x3" if request.param == "f16x3" else "fp32"
    yield request.param
    nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS, cvmod.DEFAULT_MLP_MATH = old


@pytest.mark.parametrize("tag", ["id", "proj", "down"])
def test_basic_block_golden_split(tag, split_everything):
    conv_base.test_basic_block_golden(tag)


def test_encoder_decoder_goldens_split(split_everything):
    conv_base.test_cvencoder_golden()
    conv_base.test_decoders_golden("bd")
    conv_base.test_decoders_golden("depth")
    conv_base.test_cvencoder_decoder_batched_vs_oracle()


@pytest.mark.parametrize("reg", [False, True])
def test_skip_decoder_and_matching_head_goldens_split(reg, split_everything):
    conv_base.test_skip_decoder_golden(reg)
    conv_base.test_matching_head_golden_and_layouts()


def test_pipeline_cases_split(split_everything):
    pipe_base.test_bd_hot_path_matches_oracle((1, 2, 24, 32, 16, 3))
    pipe_base.test_bd_hot_path_matches_oracle((2, 7, 16, 24, 64, 2))
    pipe_base.test_depth_model_hot_path_matches_oracle()
    pipe_base.test_prior_channel_path()
    pipe_base.test_temporal_sequence_with_prior_d96()
