"""Fused MLP feature volume (csrc/feature_volume.hip) vs reference goldens and the oracle."""
import numpy as np
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err
from oracle import cost_volume as ocv

pytestmark = pytest.mark.gpu


def _manager(K, H, W, D, seed, C=16):
    from implicit_depth_amd.cost_volume import FeatureVolumeManager

    m = FeatureVolumeManager(H, W, D, mlp_channels=[202, 128, 128, 1], num_source_views=K, matching_dim_size=C)
    syn.fill_state_dict(m.mlp, seed=seed, gain=1.4)
    return m


@pytest.mark.parametrize("name", ["g2_small", "g2_b2", "g2_k2"])
def test_matches_reference_golden(name):
    g = load_golden(name)
    B, K, C, H, W, D, seed, bv, rv = [int(v) for v in g["dims"]]
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv)
    m = _manager(K, H, W, D, int(g["mlp_seed"]))
    np.testing.assert_allclose(np.stack([[t.double().sum().item(), t.double().abs().sum().item(), (t.double() ** 2).sum().item()]
                                         for t in m.mlp.state_dict().values()]), g["mlp_chk"], rtol=1e-12)
    m.cuda()
    fv, low, planes, mask = m(**{k: v.cuda() for k, v in inp.items()}, return_mask=True)
    assert fv.shape == (B, D, H, W) and planes.shape == (B, D, H, W) and mask.dtype == torch.bool
    assert rel_err(fv.cpu(), g["feature_volume"]) < TOL
    assert (mask.cpu() != torch.as_tensor(g["overall_mask"])).float().mean().item() < 2e-3
    assert ((low.cpu() - torch.as_tensor(g["lowest_cost"])).abs() > 1e-5).float().mean().item() < 5e-3
    # return_mask=False -> None, like the reference
    assert m(**{k: v.cuda() for k, v in inp.items()})[3] is None


@pytest.mark.parametrize("shape", [(1, 1, 9, 13, 3), (2, 5, 17, 23, 6), (1, 7, 24, 32, 64), (3, 4, 8, 8, 2), (1, 8, 12, 20, 5)])
def test_matches_oracle_fp64(shape):
    B, K, H, W, D = shape
    inp = syn.cost_volume_inputs(B, K, 16, H, W, seed=K, behind_view=K - 1 if K > 2 else -1, big_rotation_view=0 if K > 3 else -1)
    m = _manager(K, H, W, D, 77 + K)
    w = {k: v.double() for k, v in m.mlp.state_dict().items()}
    d = {k: v.double() for k, v in inp.items()}
    ref, rlow, _, rmask = ocv.feature_volume(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                                             0.25, 5.0, D, w, return_mask=True)
    m.cuda()
    fv, low, planes, mask = m(**{k: v.cuda() for k, v in inp.items()}, return_mask=True)
    assert rel_err(fv.cpu(), ref) < TOL
    assert (mask.cpu() != rmask).float().mean().item() < 2e-3
    assert ((low.cpu().double() - rlow).abs() > 1e-5).float().mean().item() < 5e-3


# (B, K, C, H, W, D): the reference's config surface beyond the shipped yaml — FeatureVolumeManager takes any num_source_views
# (cost_volume.py:382-435; DepthModel passes model_num_views - 1, depth_model.py:206-212) and matching_feature_dims is an
# option (options.py:138).  K > 8 or C = 32 run on fv_mlp_gen_k (W1 blocks streamed through L2, metadata of view q + 4j)
@pytest.mark.parametrize("shape", [(1, 9, 16, 12, 20, 5), (2, 12, 16, 17, 23, 6), (1, 16, 16, 9, 13, 3), (1, 7, 32, 24, 32, 8), (2, 3, 32, 10, 14, 4), (1, 10, 32, 8, 12, 3)])
def test_generic_kernel_more_views_and_wider_features_vs_oracle_fp64(shape):
    B, K, C, H, W, D = shape
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed=K + C, behind_view=K - 1, big_rotation_view=0)
    m = _manager(K, H, W, D, 91 + K, C=C)
    assert m.mlp.net[0].in_features == C * (K + 1) + 10 * K + 4
    w = {k: v.double() for k, v in m.mlp.state_dict().items()}
    d = {k: v.double() for k, v in inp.items()}
    ref, rlow, _, rmask = ocv.feature_volume(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                                             0.25, 5.0, D, w, return_mask=True)
    m.cuda()
    fv, low, planes, mask = m(**{k: v.cuda() for k, v in inp.items()}, return_mask=True)
    assert rel_err(fv.cpu(), ref) < TOL
    assert (mask.cpu() != rmask).float().mean().item() < 2e-3
    assert ((low.cpu().double() - rlow).abs() > 1e-5).float().mean().item() < 5e-3


def test_full_size_golden():
    """BASELINE.json's size (96x128 matching map, K=7, D=64): reference checksums + strided slices."""
    g = load_golden("g2_full_k7d64")
    B, K, C, H, W, D, seed = [int(v) for v in g["dims"][:7]]
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed)
    m = _manager(K, H, W, D, int(g["mlp_seed"])).cuda()
    fv, low, planes, mask = m(**{k: v.cuda() for k, v in inp.items()}, return_mask=True)
    fv, low, mask = fv.cpu(), low.cpu(), mask.cpu()
    assert rel_err(fv[:, ::4, ::6, ::8], g["fv_slice"]) < TOL
    from conftest import block_err

    be = block_err(fv, load_golden("g_full_blocks")["g2_full_k7d64_fv_8x8x8"], 8, 8, 8)  # every 8x8x8 block, not only the slice points
    assert be < 5e-6, be
    s = fv.double()
    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["fv_chk"][1:], rtol=1e-4)
    assert ((low[:, ::3, ::4] - torch.as_tensor(g["lowest_slice"])).abs() > 1e-5).float().mean().item() < 5e-3
    assert (mask[:, ::3, ::4] != torch.as_tensor(g["mask_slice"])).float().mean().item() < 2e-3


def test_wrong_view_count_is_an_error():
    from implicit_depth_amd import _lib

    m = _manager(7, 8, 8, 2, 1).cuda()
    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(1, 2, 16, 8, 8, 0).items()}
    with pytest.raises(_lib.IdhError):
        m(**inp)


def test_pipeline_with_feature_volume():
    """mlp_feature_volume config (every shipped checkpoint): HotPath == module-by-module."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.pipeline import HotPath

    B, K, H, W, D, P = 1, 7, 16, 24, 16, 2
    cv = _manager(K, H, W, D, 5)
    cve = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    mlp = net.BinaryMLPNetwork(dec.num_ch_dec)
    for i, mm in enumerate((cve, dec, mlp)):
        syn.fill_state_dict(mm, seed=60 + i)
    model = HotPath(cv, cve, dec, mlp).cuda()
    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(B, K, 16, H, W, 2).items()}
    pyr = [t.cuda() for t in syn.encoder_pyramid(B, H * 4, W * 4, seed=2)]
    rd = syn.rendered_depth_planes(B, H * 2, W * 2, P).cuda()
    out = model(inp["cur_feats"], inp["src_feats"], pyr, inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"], inp["cur_invK"],
                rendered_depth=rd, return_mask=True, return_features=True)
    # plain numbers: planes expanded in the kernel exactly as HotPath does (device tensors take the torch-computed planes)
    vol, low, _, mask = cv(**dict(inp, min_depth=0.25, max_depth=5.0), return_mask=True)
    enc = cve(vol, pyr[1:])
    feats = dec([pyr[0]] + enc)
    assert rel_err(out["feature_s0_b1hw"], feats["feature_s0_b1hw"]) < 1e-6
    assert torch.equal(out["overall_mask_bhw"], mask) and torch.equal(out["lowest_cost_bhw"], low)
