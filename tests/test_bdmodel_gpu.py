"""End-to-end check against the REFERENCE's BDModel.forward (golden G5, stub backbones): the
hot path is fed the matching / encoder features the reference model produced and must
reproduce its ``pred_0``, ``lowest_cost_bhw`` and ``overall_mask_bhw`` — this pins the forward
orchestration (relative poses bd_model.py:196-204, volume -> encoder -> decoder -> per-plane MLP)."""
import pytest
import torch
from torch import nn

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _holder(K, volume):
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net

    h = nn.Module()
    H, W, D = 24, 32, 16
    h.cost_volume = cv.FeatureVolumeManager(H, W, D, num_source_views=K) if volume == "mlp" else cv.CostVolumeManager(H, W, D)
    h.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    h.depth_decoder = net.BDDecoderPP([24] + h.cost_volume_net.num_ch_enc)
    h.binary_mlp = net.BinaryMLPNetwork(h.depth_decoder.num_ch_dec, mlp_size=128, use_prior=False)
    syn.fill_state_dict(h, seed=30)  # name-keyed: same tensors the reference BDModel received
    return h


@pytest.mark.parametrize("volume", ["dot", "mlp"])
def test_hot_path_reproduces_reference_bdmodel_forward(volume):
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden(f"g5_bdmodel_{volume}")
    K = int(g["K"])
    h = _holder(K, volume)
    hot_keys = sorted(k for k in h.state_dict() if not k.startswith("cost_volume.linear") or True)
    assert sorted(h.state_dict()) == list(g["keys"])
    h.cuda()
    cur, src = syn.frame_tuple(1, K, 96, 128, seed=31, P=3)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    src_cam_T_cur_cam = src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1)
    cur_cam_T_src_cam = cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"]
    hot = hot_path_of(h)
    t = lambda name: torch.as_tensor(g[name]).cuda()
    out = hot(t("matching_cur"), t("matching_src"), [t(f"enc{i}") for i in range(5)], src_cam_T_cur_cam, cur_cam_T_src_cam,
              src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], return_mask=True)
    assert rel_err(out["pred_0"].cpu(), g["pred_0"]) < TOL
    assert ((out["lowest_cost_bhw"].cpu() - torch.as_tensor(g["lowest_cost"])).abs() > 1e-5).float().mean().item() < 5e-3
    if volume == "mlp":
        assert (out["overall_mask_bhw"].cpu() != torch.as_tensor(g["overall_mask"])).float().mean().item() < 2e-3
    else:
        assert out["overall_mask_bhw"] is None


def test_hot_path_reproduces_reference_depthmodel_forward():
    """DepthModel.forward (SimpleRecon regression baseline, depth_model.py:280-440), golden G9:
    MLP feature volume built with num_source_views=K (K=2 here), DepthDecoderPP heads, exp()."""
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g9_depthmodel")
    K = int(g["K"])
    h = nn.Module()
    H, W, D = 24, 32, 16
    h.cost_volume = cv.FeatureVolumeManager(H, W, D, num_source_views=K)
    h.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    h.depth_decoder = net.DepthDecoderPP([24] + h.cost_volume_net.num_ch_enc)
    syn.fill_state_dict(h, seed=33)
    assert sorted(h.state_dict()) == list(g["keys"])
    h.cuda()
    cur, src = syn.frame_tuple(1, K, 96, 128, seed=34, P=1)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    src_cam_T_cur_cam = src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1)
    cur_cam_T_src_cam = cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"]
    hot = hot_path_of(h)
    t = lambda name: torch.as_tensor(g[name]).cuda()
    out = hot(t("matching_cur"), t("matching_src"), [t(f"enc{i}") for i in range(5)], src_cam_T_cur_cam, cur_cam_T_src_cam,
              src["K_s1_b44"], cur["invK_s1_b44"], return_mask=True)
    for i in range(4):
        assert rel_err(out[f"log_depth_pred_s{i}_b1hw"].cpu(), g[f"log_depth_pred_s{i}_b1hw"]) < TOL
        assert rel_err(out[f"depth_pred_s{i}_b1hw"].cpu(), g[f"depth_pred_s{i}_b1hw"]) < 5 * TOL  # exp() amplifies
    assert ((out["lowest_cost_bhw"].cpu() - torch.as_tensor(g["lowest_cost_bhw"])).abs() > 1e-5).float().mean().item() < 5e-3
    assert (out["overall_mask_bhw"].cpu() != torch.as_tensor(g["overall_mask_bhw"])).float().mean().item() < 2e-3


@pytest.fixture
def conv_plan(request):
    """"default": the kernel selection a one-frame plan gets by itself (no layer reaches the F(4x4) tile threshold).  "wino4": the thresholds
    of nhwc.wino4_eligible lowered so that EVERY eligible layer runs conv3x3_wino4_k - the kernel selection of the B = 32 bench plan, compared
    with the reference's full-size goldens directly instead of through a batch-invariance hop."""
    from implicit_depth_amd import nhwc

    old = (nhwc.WINOGRAD4, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL)
    if request.param == "wino4":
        nhwc.WINOGRAD4, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL = True, 1, 0.0
    yield request.param
    nhwc.WINOGRAD4, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL = old


def _assert_plan_kernels(hot, conv_plan, min_wino4):
    from implicit_depth_amd import nhwc

    ops = [op for ent in hot._plans.values() for op in ent["plan"].ops if op.kind == nhwc.OP_CONV]
    if any(op.tile_m in nhwc.SPLIT_CODE.values() for op in ops):
        return  # (called from tests/test_hot_path_split_gpu.py with the split-precision kernels selected)
    n4 = sum(op.tile_m == nhwc.TILE_WINO4 for op in ops)
    print(f"conv plan {conv_plan}: {n4} of {len(ops)} convs on conv3x3_wino4_k")
    assert (n4 >= min_wino4) if conv_plan == "wino4" else (n4 == 0)


@pytest.mark.parametrize("volume,conv_plan", [("mlp", "default"), ("dot", "default"), ("mlp", "wino4")], indirect=["conv_plan"])
def test_full_size_bdmodel_forward_golden(volume, conv_plan):
    """BASELINE.json's size: the reference's BDModel.forward on a 512x384 tuple — mlp_feature_volume K=7
    (reference-native) and simple_cost_volume K=8 (BASELINE's literal "8 views") — D=64, 8 query planes;
    goldens g5_full_*: checksums + strided slices, backbone features regenerated from the seeds the reference
    run used.  conv_plan "wino4": the same comparison with every eligible conv on the F(4x4) kernel."""
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden(f"g5_full_bdmodel_{volume}")
    K, Hi, Wi, D, P = [int(v) for v in g["dims"]]
    h = nn.Module()
    h.cost_volume = (cv.FeatureVolumeManager(Hi // 4, Wi // 4, D, num_source_views=K) if volume == "mlp"
                     else cv.CostVolumeManager(Hi // 4, Wi // 4, D))
    h.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    h.depth_decoder = net.BDDecoderPP([24] + h.cost_volume_net.num_ch_enc)
    h.binary_mlp = net.BinaryMLPNetwork(h.depth_decoder.num_ch_dec, mlp_size=128, use_prior=False)
    syn.fill_state_dict(h, seed=30)
    assert sorted(h.state_dict()) == list(g["keys"])
    h.cuda()
    cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=31, P=P)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 71, "mc").cuda()
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 72, "ms").cuda()
    pyr = [t.cuda() for t in syn.encoder_pyramid(1, Hi, Wi, seed=73)]
    hot = hot_path_of(h)
    out = hot(mc, ms, pyr, src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1),
              cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"], src["K_s1_b44"], cur["invK_s1_b44"],
              rendered_depth=cur["rendered_depth"], return_mask=True)
    _assert_plan_kernels(hot, conv_plan, 80)
    pred, low = out["pred_0"].cpu(), out["lowest_cost_bhw"].cpu()
    assert rel_err(pred[:, :, ::6, ::8], g["pred_slice"]) < TOL
    if volume == "mlp":
        from conftest import block_err

        be = block_err(pred, load_golden("g_full_blocks")["g5_full_bdmodel_mlp_pred_1x8x8"], 1, 8, 8)  # every 8x8 block of every plane
        assert be < 5e-5, be
    s = pred.double()
    import numpy as np

    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["pred_chk"][1:], rtol=2e-4)
    assert ((low[:, ::3, ::4] - torch.as_tensor(g["lowest_slice"])).abs() > 1e-5).float().mean().item() < 5e-3
    if volume == "mlp":
        assert (out["overall_mask_bhw"].cpu()[:, ::3, ::4] != torch.as_tensor(g["mask_slice"])).float().mean().item() < 2e-3
    else:
        assert out["overall_mask_bhw"] is None


def test_full_size_temporal_prior_golden():
    """BASELINE.json config 5 at full size: 512x384 8-frame tuple, 96 depth planes, prior-enabled occlusion MLP with
    the previous prediction warped by sample_prior — against the reference's BDModel.forward (golden g5_full_temporal)."""
    import numpy as np

    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g5_full_temporal_d96")
    K, Hi, Wi, D, P = [int(v) for v in g["dims"]]
    h = nn.Module()
    h.cost_volume = cv.FeatureVolumeManager(Hi // 4, Wi // 4, D, num_source_views=K)
    h.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    h.depth_decoder = net.BDDecoderPP([24] + h.cost_volume_net.num_ch_enc)
    h.binary_mlp = net.BinaryMLPNetwork(h.depth_decoder.num_ch_dec, mlp_size=128, use_prior=True)
    syn.fill_state_dict(h, seed=30)
    assert sorted(h.state_dict()) == list(g["keys"])
    h.cuda()
    cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=31, P=P)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    prior_inputs = {"prior_prediction": torch.sigmoid(syn.randn((1, 1, Hi // 2, Wi // 2), 74, "prior")).cuda(),
                    "prior_cam_T_world": torch.linalg.inv(syn.source_pose(1).float())[None].cuda(),
                    "world_T_cam_b44": cur["world_T_cam_b44"], "K_s0_b44": cur["K_s0_b44"], "invK_s0_b44": cur["invK_s0_b44"]}
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 71, "mc").cuda()
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 72, "ms").cuda()
    pyr = [t.cuda() for t in syn.encoder_pyramid(1, Hi, Wi, seed=73)]
    hot = hot_path_of(h)
    out = hot(mc, ms, pyr, src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1),
              cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"], src["K_s1_b44"], cur["invK_s1_b44"],
              rendered_depth=cur["rendered_depth"], prior_inputs=prior_inputs, return_mask=True)
    pm = out["prior_mask"].cpu()
    assert ((pm[:, :, ::6, ::8] - torch.as_tensor(g["prior_mask_slice"])).abs() > 1e-6).float().mean().item() < 2e-3  # nearest sampling
    pred = out["pred_0"].cpu()
    # a flipped nearest-neighbour prior sample changes one pixel's logit: compare the bulk tightly, allow rare outliers
    d = (pred[:, :, ::6, ::8] - torch.as_tensor(g["pred_slice"])).abs() / torch.as_tensor(g["pred_slice"]).abs().max()
    assert (d > TOL).float().mean().item() < 2e-3
    s = pred.double()
    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["pred_chk"][1:], rtol=5e-4)
    assert ((out["lowest_cost_bhw"].cpu()[:, ::3, ::4] - torch.as_tensor(g["lowest_slice"])).abs() > 1e-5).float().mean().item() < 5e-3


@pytest.mark.parametrize("conv_plan", ["default", "wino4"], indirect=True)
def test_full_size_depthmodel_forward_golden(conv_plan):
    """The reference's DepthModel.forward at 512x384 (mlp_feature_volume K=7, D=64, DepthDecoderPP heads + exp); conv_plan "wino4": every
    eligible conv on the F(4x4) kernel."""
    import numpy as np

    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g9_full_depthmodel")
    K, Hi, Wi, D = [int(v) for v in g["dims"]]
    h = nn.Module()
    h.cost_volume = cv.FeatureVolumeManager(Hi // 4, Wi // 4, D, num_source_views=K)
    h.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    h.depth_decoder = net.DepthDecoderPP([24] + h.cost_volume_net.num_ch_enc)
    syn.fill_state_dict(h, seed=33)
    assert sorted(h.state_dict()) == list(g["keys"])
    h.cuda()
    cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=34, P=1)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 75, "mc").cuda()
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 76, "ms").cuda()
    pyr = [t.cuda() for t in syn.encoder_pyramid(1, Hi, Wi, seed=77)]
    hot = hot_path_of(h)
    out = hot(mc, ms, pyr, src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1),
              cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"], src["K_s1_b44"], cur["invK_s1_b44"], return_mask=True)
    _assert_plan_kernels(hot, conv_plan, 80)
    for i in range(4):
        sl = (slice(None), slice(None), slice(None, None, 3), slice(None, None, 4)) if i >= 2 else (slice(None), slice(None), slice(None, None, 6), slice(None, None, 8))
        for nm, tol in ((f"log_depth_pred_s{i}_b1hw", TOL), (f"depth_pred_s{i}_b1hw", 5 * TOL)):  # exp() amplifies
            y = out[nm].cpu()
            assert rel_err(y[sl], g[nm + "_slice"]) < tol, nm
            s = y.double()
            np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g[nm + "_chk"][1:], rtol=1e-3)
    assert ((out["lowest_cost_bhw"].cpu()[:, ::3, ::4] - torch.as_tensor(g["lowest_slice"])).abs() > 1e-5).float().mean().item() < 5e-3
    assert (out["overall_mask_bhw"].cpu()[:, ::3, ::4] != torch.as_tensor(g["mask_slice"])).float().mean().item() < 2e-3


# ---- forward-level drop-in: from RAW IMAGES, through stand-in backbones with the weights the reference run used ------------
class _RunOpts:
    matching_scale = 1
    min_matching_depth = 0.25
    max_matching_depth = 5.0
    use_prior = False


def _standin_model(K, volume, decoder="bd"):
    """An object with the attribute tree of the reference BDModel / DepthModel (same names -> syn.fill_state_dict gives
    the tensors gen_golden.py gave the reference model): stub image encoder + stub ResNet stem (as in the golden run),
    drop-in hot-path modules."""
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net

    m = nn.Module()
    m.encoder = syn.StubImageEncoder()
    H, W, D = 24, 32, 16
    m.cost_volume = cv.FeatureVolumeManager(H, W, D, num_source_views=K) if volume == "mlp" else cv.CostVolumeManager(H, W, D)
    stem = syn.StubResnetStem()
    m.matching_model = net.ResnetMatchingEncoder([stem.conv1, stem.bn1, stem.relu, stem.maxpool, stem.layer1], 16)
    m.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    if decoder == "bd":
        m.depth_decoder = net.BDDecoderPP([24] + m.cost_volume_net.num_ch_enc)
        m.binary_mlp = net.BinaryMLPNetwork(m.depth_decoder.num_ch_dec, mlp_size=128, use_prior=False)
    else:
        m.depth_decoder = net.DepthDecoderPP([24] + m.cost_volume_net.num_ch_enc)
    m.run_opts = _RunOpts()
    m.thresholder = None
    return m


@pytest.mark.parametrize("unbatched", [True, False])
@pytest.mark.parametrize("volume", ["dot", "mlp"])
def test_fused_forward_from_raw_images_matches_reference_bdmodel(volume, unbatched):
    """dropin.fused_forward(model)("test", cur_data, src_data, ...) — BDModel.forward's own signature — starting at the
    images: stub backbones in torch, everything else in one HotPath pass; golden G5 = the reference's BDModel.forward on
    the same tuple with the same (name-keyed) weights."""
    from implicit_depth_amd.dropin import fused_forward

    g = load_golden(f"g5_bdmodel_{volume}")
    K = int(g["K"])
    m = _standin_model(K, volume)
    syn.fill_state_dict(m, seed=30)
    m.cuda().eval()
    cur, src = syn.frame_tuple(1, K, 96, 128, seed=31, P=3)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    fwd = fused_forward(m)
    out = fwd("test", cur, src, unbatched_matching_encoder_forward=unbatched, return_mask=True)
    assert rel_err(out["pred_0"].cpu(), g["pred_0"]) < TOL
    assert ((out["lowest_cost_bhw"].cpu() - torch.as_tensor(g["lowest_cost"])).abs() > 1e-5).float().mean().item() < 5e-3
    if volume == "mlp":
        assert (out["overall_mask_bhw"].cpu() != torch.as_tensor(g["overall_mask"])).float().mean().item() < 2e-3
        o2 = fwd("test", cur, src, unbatched_matching_encoder_forward=unbatched, infer_depth=True)
        from hot_helpers import search_agrees

        search_agrees(o2["search_depths"], g["search_depths"], g["search_margin"])
    with pytest.raises(Exception):
        fwd("train", cur, src)


def test_fused_forward_from_raw_images_matches_reference_depthmodel():
    from implicit_depth_amd.dropin import fused_forward

    g = load_golden("g9_depthmodel")
    K = int(g["K"])
    m = _standin_model(K, "mlp", decoder="depth")
    syn.fill_state_dict(m, seed=33)
    m.cuda().eval()
    cur, src = syn.frame_tuple(1, K, 96, 128, seed=34, P=1)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    out = fused_forward(m)("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
    for i in range(4):
        assert rel_err(out[f"log_depth_pred_s{i}_b1hw"].cpu(), g[f"log_depth_pred_s{i}_b1hw"]) < TOL
        assert rel_err(out[f"depth_pred_s{i}_b1hw"].cpu(), g[f"depth_pred_s{i}_b1hw"]) < 5 * TOL
    assert sorted(k for k in out if "depth_pred" in k) == sorted(k for k in g if "depth_pred" in k)


@pytest.mark.parametrize("volume", ["dot", "mlp"])
def test_bench_call_shape_workloads_match_reference_bdmodel(volume):
    """bench.py's two call-shape workloads on the golden's tuple and weights: the module-swap forward (dropin.convert + the reference's own
    call sequence bd_model.py:175-311, module by module, 8 BinaryMLPNetwork calls on the permuted NCHW view) and dropin.fused_forward must
    both reproduce the reference's BDModel.forward (golden G5)."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from implicit_depth_amd.dropin import convert, fused_forward

    g = load_golden(f"g5_bdmodel_{volume}")
    K = int(g["K"])
    m = bench._standin_bdmodel(K, 24, 32, 16, volume, golden_weights=True).cuda().eval()
    cur, src = syn.frame_tuple(1, K, 96, 128, seed=31, P=3)
    cur = {k: v.cuda() for k, v in cur.items()}
    src = {k: v.cuda() for k, v in src.items()}
    with torch.inference_mode():
        swap = bench._reference_shaped_forward(convert(m), cur, src, return_mask=True)
        fused = fused_forward(m)("test", cur, src, return_mask=True)
    for out in (swap, fused):
        assert rel_err(out["pred_0"].cpu(), g["pred_0"]) < TOL
        assert ((out["lowest_cost_bhw"].cpu() - torch.as_tensor(g["lowest_cost"])).abs() > 1e-5).float().mean().item() < 5e-3
        if volume == "mlp":
            assert (out["overall_mask_bhw"].cpu() != torch.as_tensor(g["overall_mask"])).float().mean().item() < 2e-3
    assert rel_err(swap["pred_0"].cpu(), fused["pred_0"].cpu()) < TOL
