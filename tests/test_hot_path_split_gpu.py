"""End-to-end hot path with the split-precision conv kernels against the REFERENCE goldens
(G5: BDModel.forward dot / mlp, G9: DepthModel.forward) — same tolerances as the fp32-MFMA path
(tests/test_bdmodel_gpu.py), plus a direct comparison of the two arithmetic modes at bench shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from implicit_depth_amd import nhwc

import test_bdmodel_gpu as base


@pytest.fixture(params=["f16x3"])
def split_default(request):
    old, oldmin = nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS
    nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS = request.param, 1
    yield request.param
    nhwc.DEFAULT_MATH, nhwc.SPLIT_MIN_BLOCKS = old, oldmin


@pytest.mark.parametrize("volume", ["dot", "mlp"])
def test_bdmodel_golden_with_split_convs(volume, split_default):
    base.test_hot_path_reproduces_reference_bdmodel_forward(volume)


@pytest.mark.parametrize("volume", ["mlp", "dot"])
def test_full_size_bdmodel_golden_with_split_kernels(volume, split_everything):
    base.test_full_size_bdmodel_forward_golden(volume, "default")


def test_full_size_temporal_golden_with_split_kernels(split_everything):
    base.test_full_size_temporal_prior_golden()


def test_full_size_depthmodel_golden_with_split_kernels(split_everything):
    base.test_full_size_depthmodel_forward_golden("default")


def test_depthmodel_golden_with_split_convs(split_default):
    base.test_hot_path_reproduces_reference_depthmodel_forward()


@pytest.mark.parametrize("math", ["f16x3"])
def test_split_hot_path_matches_fp32_hot_path_at_bench_shape(math):
    """512x384 frames, K=7 MLP feature volume, D=64 (the bench workload at B=2): logits of the
    split-precision plan vs the fp32-MFMA plan."""
    import argparse

    from bench import HotPathWorkload

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


@pytest.fixture(params=["f16x3"])
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
    # (the opt-in mode is not the headline: one oracle case per pipeline shape; the fp32 suite runs the full set)
    pipe_base.test_bd_hot_path_matches_oracle((1, 2, 24, 32, 16, 3))
    pipe_base.test_depth_model_hot_path_matches_oracle()
    pipe_base.test_prior_channel_path()


def test_dropin_convert_with_math_selects_split_kernels_per_model():
    """dropin.hot_path_of(model, math="f16x3") on the reference-shaped holder of the G5 golden."""
    from implicit_depth_amd import cost_volume as cvmod
    from implicit_depth_amd.dropin import hot_path_of

    g = base.load_golden("g5_bdmodel_mlp")
    K = int(g["K"])
    h = base._holder(K, "mlp")
    h.cuda()
    old = nhwc.SPLIT_MIN_BLOCKS
    nhwc.SPLIT_MIN_BLOCKS = 1
    try:
        hot = hot_path_of(h, math="f16x3")
        assert hot.conv_math == "f16x3" and h.cost_volume.mlp_math == "f16x3" and h.binary_mlp.mlp_math == "f16x3"
        assert nhwc.DEFAULT_MATH == "fp32" and cvmod.DEFAULT_MLP_MATH == "fp32"  # per-model, not process-wide
        cur, src = base.syn.frame_tuple(1, K, 96, 128, seed=31, P=3)
        cur = {k: v.cuda() for k, v in cur.items()}
        src = {k: v.cuda() for k, v in src.items()}
        t = lambda name: torch.as_tensor(g[name]).cuda()
        out = hot(t("matching_cur"), t("matching_src"), [t(f"enc{i}") for i in range(5)],
                  src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1), cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"],
                  src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], return_mask=True)
        plan = next(iter(hot._plans.values()))["plan"]
        assert any(op.kind == nhwc.OP_CONV and op.tile_m == nhwc.SPLIT_CODE["f16x3"] for op in plan.ops)
        assert base.rel_err(out["pred_0"].cpu(), g["pred_0"]) < base.TOL
    finally:
        nhwc.SPLIT_MIN_BLOCKS = old


@pytest.mark.parametrize("shape", [(1, 16, 16, 9, 7, 1), (3, 64, 64, 33, 47, 1), (2, 112, 64, 24, 32, 1), (1, 64, 128, 31, 45, 2)])
def test_basic_block_ragged_shapes_split(shape, split_everything):
    conv_base.test_basic_block_vs_oracle(shape)


@pytest.mark.parametrize("math", ["fp32", "f16x3"])
def test_batch_invariance_at_bench_size(math):
    """Size-independent property at the bench's own size: frame i of the 32-frame batch (16-row tiles, level-grouped
    launches, one workgroup round per layer) must equal the same frame run alone (8- / 4-row tiles, split-K on the small
    maps) up to summation order."""
    import argparse

    from bench import HotPathWorkload

    mk = lambda B: argparse.Namespace(batch=B, views=7, planes=64, height=384, width=512, volume="mlp", conv_math=math,
                                      mlp_math="f16x3" if math == "f16x3" else "fp32")
    big = HotPathWorkload(mk(32), torch.device("cuda:0"), 0)
    big.step()
    torch.cuda.synchronize()
    ref = big.out["pred_0"]
    low = big.out["lowest_cost_bhw"]
    one = HotPathWorkload(mk(1), torch.device("cuda:0"), 0)
    for i in (0, 17, 31):
        one.d = {k: (v[i:i + 1].contiguous() if v.dim() > 0 and v.shape[0] == 32 else v) for k, v in big.d.items()}
        one.pyr = [t[i:i + 1].contiguous() for t in big.pyr]
        one.rd = big.rd[i:i + 1].contiguous()
        one.l1 = big.l1[i:i + 1].contiguous()
        one.step()
        torch.cuda.synchronize()
        a, b = one.out["pred_0"][0], ref[i]
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5, i
        assert ((one.out["lowest_cost_bhw"][0] - low[i]).abs() > 1e-5).float().mean().item() < 5e-3
