"""The whole in-scope path — matching backbone's layer1 map -> encoder head -> cost volume -> CVEncoder -> UNet++ ->
occlusion MLP / binary depth search — against the REFERENCE's BDModel.forward (goldens G5 small + full size, the
16-frame temporal sequence), with the head running inside HotPath's plan."""
import numpy as np
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err
from hot_helpers import holder, hot_keys, rel_poses, search_agrees, to_cuda

pytestmark = pytest.mark.gpu


def _thresholder():
    from implicit_depth_amd.metrics import Thresholder

    return Thresholder(torch.tensor([1.5 + 0.5 * i for i in range(8)]), torch.tensor([0.3, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7]))


@pytest.mark.parametrize("layout", ["nchw", "channels_last"])
@pytest.mark.parametrize("volume", ["dot", "mlp"])
def test_small_bdmodel_forward_from_layer1_map(volume, layout):
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden(f"g5_bdmodel_{volume}")
    K = int(g["K"])
    h = holder(K, volume, 24, 32, 16)
    assert hot_keys(h) == list(g["keys"])
    h.cuda()
    cur, src = (to_cuda(d) for d in syn.frame_tuple(1, K, 96, 128, seed=31, P=3))
    E, P = rel_poses(cur, src)
    hot = hot_path_of(h)
    assert hot.matching_model is h.matching_model
    l1 = torch.as_tensor(g["layer1"]).cuda()
    if layout == "channels_last":
        l1 = l1.view(-1, *l1.shape[2:]).contiguous(memory_format=torch.channels_last).view(l1.shape)
        assert not l1.view(-1, *l1.shape[2:]).is_contiguous()
    enc = [torch.as_tensor(g[f"enc{i}"]).cuda() for i in range(5)]
    out = hot(None, None, enc, E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], return_mask=True,
              matching_layer1=l1, return_matching_feats=True)
    assert rel_err(out["matching_cur_feats"].cpu(), g["matching_cur"]) < TOL
    assert rel_err(out["matching_src_feats"].cpu(), g["matching_src"]) < TOL
    assert rel_err(out["pred_0"].cpu(), g["pred_0"]) < TOL
    assert ((out["lowest_cost_bhw"].cpu() - torch.as_tensor(g["lowest_cost"])).abs() > 1e-5).float().mean().item() < 5e-3
    if volume == "mlp":
        assert (out["overall_mask_bhw"].cpu() != torch.as_tensor(g["overall_mask"])).float().mean().item() < 2e-3
    # second call replays the cached plan (pointer patching only)
    out2 = hot(None, None, enc, E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], matching_layer1=l1)
    assert torch.equal(out2["pred_0"], out["pred_0"])


@pytest.mark.parametrize("thr", [False, True])
def test_small_infer_depth_matches_reference(thr):
    """BDModel.forward(..., infer_depth=True) (bd_model.py:273-292), without / with the per-depth Thresholder
    (binary_metrics_utils.py:42-52): HotPath(infer_depth=True) against the reference's search_depths and last logits."""
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g5_bdmodel_mlp")
    K = int(g["K"])
    h = holder(K, "mlp", 24, 32, 16).cuda()
    cur, src = (to_cuda(d) for d in syn.frame_tuple(1, K, 96, 128, seed=31, P=3))
    E, P = rel_poses(cur, src)
    hot = hot_path_of(h)
    hot.thresholder = _thresholder() if thr else None
    t = lambda name: torch.as_tensor(g[name]).cuda()
    out = hot(t("matching_cur"), t("matching_src"), [t(f"enc{i}") for i in range(5)], E, P, src["K_s1_b44"], cur["invK_s1_b44"],
              rendered_depth=cur["rendered_depth"], infer_depth=True)
    tag = "_thr" if thr else ""
    agree = search_agrees(out["search_depths"], g["search_depths" + tag], g["search_margin" + tag])
    assert rel_err(out["pred_0"].cpu()[agree], torch.as_tensor(g["search_pred" + tag])[agree]) < TOL
    assert tuple(out["search_depths"].shape) == (1, 1, 48, 64)


@pytest.mark.parametrize("volume", ["mlp", "dot"])
def test_full_size_forward_from_layer1_map(volume):
    """BASELINE.json's size, starting at the layer1 map: 512x384, K=7 MLP volume / K=8 dot volume, D=64, 8 planes."""
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden(f"g5_full_bdmodel_{volume}")
    K, Hi, Wi, D, Pq = [int(v) for v in g["dims"]]
    h = holder(K, volume, Hi // 4, Wi // 4, D)
    assert hot_keys(h) == list(g["keys"]) and sorted(k for k in h.state_dict() if k.startswith("matching_model")) == list(g["head_keys"])
    h.cuda()
    cur, src = (to_cuda(d) for d in syn.frame_tuple(1, K, Hi, Wi, seed=31, P=Pq))
    E, P = rel_poses(cur, src)
    l1 = syn.layer1_maps(1, K, Hi // 4, Wi // 4, seed=78).cuda()
    pyr = [t.cuda() for t in syn.encoder_pyramid(1, Hi, Wi, seed=73)]
    hot = hot_path_of(h)
    out = hot(None, None, pyr, E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], return_mask=True,
              matching_layer1=l1, return_matching_feats=True)
    feats = torch.cat([out["matching_cur_feats"][:, None], out["matching_src_feats"]], 1).cpu()
    assert rel_err(feats[:, :, :, ::6, ::8], g["head_feats_slice"]) < TOL
    pred, low = out["pred_0"].cpu(), out["lowest_cost_bhw"].cpu()
    assert rel_err(pred[:, :, ::6, ::8], g["head_pred_slice"]) < TOL
    if volume == "mlp":
        from conftest import block_err

        be = block_err(pred, load_golden("g_full_blocks")["g5_full_bdmodel_mlp_head_pred_1x8x8"], 1, 8, 8)  # every 8x8 block of every plane
        assert be < 5e-5, be
    s = pred.double()
    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["head_pred_chk"][1:], rtol=2e-4)
    assert ((low[:, ::3, ::4] - torch.as_tensor(g["head_lowest_slice"])).abs() > 1e-5).float().mean().item() < 5e-3
    if volume == "mlp":
        assert (out["overall_mask_bhw"].cpu()[:, ::3, ::4] != torch.as_tensor(g["head_mask_slice"])).float().mean().item() < 2e-3
        for thr in (False, True):  # the binary depth search at full size
            hot.thresholder = _thresholder() if thr else None
            o = hot(None, None, pyr, E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], matching_layer1=l1,
                    infer_depth=True)
            tag = "_thr" if thr else ""
            sd = o["search_depths"].cpu()
            agree = search_agrees(sd[:, :, ::6, ::8], g["head_search_depths" + tag], g["head_search_margin" + tag])
            assert rel_err(o["pred_0"].cpu()[:, :, ::6, ::8][agree], torch.as_tensor(g["head_search_pred" + tag])[agree]) < TOL
            np.testing.assert_allclose(sd.double().sum().item(), g["head_search_depths" + tag + "_chk"][0], rtol=1e-3)


def _temporal_holder(K, Hi, Wi, D):
    h = holder(K, "mlp", Hi // 4, Wi // 4, D, use_prior=True)
    return h.cuda()


def test_full_size_infer_depth_with_prior():
    """The binary depth search on the temporal model (prior channel = previous prediction warped by sample_prior)."""
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g5_full_temporal_d96")
    K, Hi, Wi, D, Pq = [int(v) for v in g["dims"]]
    h = _temporal_holder(K, Hi, Wi, D)
    cur, src = (to_cuda(d) for d in syn.frame_tuple(1, K, Hi, Wi, seed=31, P=Pq))
    E, P = rel_poses(cur, src)
    prior_inputs = {"prior_prediction": torch.sigmoid(syn.randn((1, 1, Hi // 2, Wi // 2), 74, "prior")).cuda(),
                    "prior_cam_T_world": torch.linalg.inv(syn.source_pose(1).float())[None].cuda(),
                    "world_T_cam_b44": cur["world_T_cam_b44"], "K_s0_b44": cur["K_s0_b44"], "invK_s0_b44": cur["invK_s0_b44"]}
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 71, "mc").cuda()
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 72, "ms").cuda()
    pyr = [t.cuda() for t in syn.encoder_pyramid(1, Hi, Wi, seed=73)]
    hot = hot_path_of(h)
    for thr in (False, True):
        hot.thresholder = _thresholder() if thr else None
        o = hot(mc, ms, pyr, E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], prior_inputs=prior_inputs,
                infer_depth=True)
        tag = "_thr" if thr else ""
        # a flipped nearest-neighbour prior sample changes that pixel's whole search: allow the same rare outliers as
        # test_full_size_temporal_prior_golden
        sd = o["search_depths"].cpu()[:, :, ::6, ::8]
        ref, margin = torch.as_tensor(g["search_depths" + tag]), torch.as_tensor(g["search_margin" + tag])
        bad = ((sd - ref).abs() >= 1e-6) & (margin >= 1e-4)
        assert bad.float().mean().item() < 2e-3


def test_temporal_sequence_of_16_frames_with_carried_prior():
    """BASELINE.json config 5 as it runs (inference/inference.py:139-157): 16 frames, each receiving the previous
    frame's sigmoid(pred_0) + cam_T_world as the prior; D=96, one query plane at 2 m, from the layer1 map.  Golden =
    the reference's own loop.  A flipped nearest-neighbour prior sample changes one pixel's logit, so frames are
    compared by the fraction of pixels beyond tolerance; the checksums bound any drift over the sequence."""
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g5_temporal_seq16")
    K, Hi, Wi, D, T = [int(v) for v in g["dims"]]
    h = _temporal_holder(K, Hi, Wi, D)
    hot = hot_path_of(h)
    prev_pred = prev_cam_T_world = None
    ref = torch.as_tensor(g["pred_slice"])
    scale = ref.abs().max()
    for t in range(T):
        cur, src, l1, pyr = syn.temporal_frame(t, K, Hi, Wi, seed=31)
        cur, src = to_cuda(cur), to_cuda(src)
        E, P = rel_poses(cur, src)
        prior_inputs = None
        if prev_pred is not None:
            prior_inputs = {"prior_prediction": prev_pred, "prior_cam_T_world": prev_cam_T_world, "world_T_cam_b44": cur["world_T_cam_b44"],
                            "K_s0_b44": cur["K_s0_b44"], "invK_s0_b44": cur["invK_s0_b44"]}
        out = hot(None, None, [p.cuda() for p in pyr], E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"],
                  prior_inputs=prior_inputs, matching_layer1=l1.cuda())
        pred = out["pred_0"]
        d = (pred.cpu()[:, :, ::6, ::8] - ref[t : t + 1]).abs() / scale
        assert (d > TOL).float().mean().item() < 5e-3, (t, (d > TOL).float().mean().item())
        s = pred.double()
        np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["pred_chk"][t][1:], rtol=1e-3)
        if t > 0:
            pm = out["prior_mask"].cpu()[:, :, ::6, ::8]
            assert ((pm - torch.as_tensor(g["prior_slice"][t : t + 1])).abs() > 1e-4).float().mean().item() < 5e-3
        prev_pred, prev_cam_T_world = torch.sigmoid(pred), cur["cam_T_world_b44"]
