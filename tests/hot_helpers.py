"""Shared builders for the end-to-end GPU tests: a holder module with the drop-in hot-path modules under the
reference's attribute names (so ``syn.fill_state_dict`` reproduces the weights the reference model received) and the
relative-pose inputs of ``BDModel.forward`` (bd_model.py:196-204)."""
import torch
from torch import nn

import implicit_depth_amd.synthetic as syn


def holder(K, volume, H, W, D, seed=30, use_prior=False, decoder="bd", with_mlp=True, with_head=True):
    from implicit_depth_amd import cost_volume as cv
    from implicit_depth_amd import networks as net

    h = nn.Module()
    h.cost_volume = cv.FeatureVolumeManager(H, W, D, num_source_views=K) if volume == "mlp" else cv.CostVolumeManager(H, W, D)
    h.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    h.depth_decoder = (net.BDDecoderPP if decoder == "bd" else net.DepthDecoderPP)([24] + h.cost_volume_net.num_ch_enc)
    if with_mlp:
        h.binary_mlp = net.BinaryMLPNetwork(h.depth_decoder.num_ch_dec, mlp_size=128, use_prior=use_prior)
    if with_head:  # the third-party stem is the caller's business: identity stand-ins keep the reference's key names net.5 / net.8
        h.matching_model = net.ResnetMatchingEncoder([nn.Identity() for _ in range(5)], 16)
    syn.fill_state_dict(h, seed=seed)  # name-keyed: same tensors the reference model received
    return h


def hot_keys(h):
    return sorted(k for k in h.state_dict() if not k.startswith("matching_model"))


def rel_poses(cur, src):
    src_cam_T_cur_cam = src["cam_T_world_b44"] @ cur["world_T_cam_b44"].unsqueeze(1)
    cur_cam_T_src_cam = cur["cam_T_world_b44"].unsqueeze(1) @ src["world_T_cam_b44"]
    return src_cam_T_cur_cam, cur_cam_T_src_cam


def to_cuda(d):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}


def search_agrees(got_sd, ref_sd, margin, min_agree=0.9):
    """infer_depth outputs: identical decisions everywhere except knife-edge pixels (the search converges onto the
    decision boundary, so its last steps sit within fp32 noise of the threshold by construction)."""
    got_sd, ref_sd, margin = (torch.as_tensor(t).float().cpu() for t in (got_sd, ref_sd, margin))
    agree = (got_sd - ref_sd).abs() < 1e-6
    assert agree.float().mean().item() > min_agree, agree.float().mean().item()
    assert bool((agree | (margin < 1e-4)).all())
    assert (got_sd - ref_sd).abs().max().item() < 0.05
    return agree
