"""Host side of the fused per-pixel occlusion MLP (csrc/mlp.hip).

``occlusion_logits`` is the fused form of the reference's per-plane loop
(experiment_modules/bd_model.py:293-304, :412-442); ``binary_mlp_forward`` keeps the exact
``BinaryMLPNetwork.forward(list_of_BHWC, max_scale_only)`` interface
(modules/networks.py:106-115) on top of the same kernel.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib


def mlp_math_of(net) -> str:
    """Arithmetic of the per-plane 128x128 layer: ``net.mlp_math`` if set, else
    ``cost_volume.DEFAULT_MLP_MATH`` ("fp32" | "f16x3", see cost_volume.MLP_MATH_MODES)."""
    from . import cost_volume as _cv

    m = getattr(net, "mlp_math", None) or _cv.DEFAULT_MLP_MATH
    if m not in _cv.MLP_MATH_MODES:
        raise _lib.IdhError(f"unknown MLP math mode {m!r} (expected one of {_cv.MLP_MATH_MODES})")
    return m


def _prepared(seq: nn.Sequential, n_feat: int, use_prior: bool, math: str = "fp32"):
    """Fragment-ordered copies of one scale's three Linear layers, cached on the module."""
    l1, l2, l3 = seq[0], seq[2], seq[4]
    key = (math,) + tuple((p.data_ptr(), _lib.param_version(p)) for p in seq.parameters())
    c = seq.__dict__.get("_idh_mlp")
    if c is not None and c[0] == key:
        return c[1]
    w1, w2 = l1.weight.detach().contiguous(), l2.weight.detach().contiguous()
    _lib.require_cuda_f32(w1, w2)
    if w1.shape[0] != 128 or tuple(w2.shape) != (128, 128) or l3.weight.shape != (1, 128):
        raise _lib.IdhError("binary MLP kernel is specialised for mlp_size=128 (reference networks.py:88)")
    if w1.shape[1] != n_feat + (2 if use_prior else 1):
        raise _lib.IdhError(f"first Linear has {w1.shape[1]} inputs, expected depth + {n_feat} features" + (" + prior" if use_prior else ""))
    L = _lib.lib()
    dev = w1.device
    w1p = torch.empty(L.idh_packed_mlp_weight_floats(n_feat), device=dev)
    st = _lib.stream_ptr()
    _lib.check(L.idh_pack_mlp_weight(w1.data_ptr(), w1p.data_ptr(), w1.shape[1], 1, n_feat, st), "idh_pack_mlp_weight")
    if math == "f16x3":
        w2p = torch.empty(L.idh_packed_mlp_weight_f16_bytes(128) // 4, device=dev, dtype=torch.int32)
        _lib.check(L.idh_pack_mlp_weight_f16(w2.data_ptr(), w2p.data_ptr(), 128, 0, 128, st), "idh_pack_mlp_weight_f16")
    else:
        w2p = torch.empty(L.idh_packed_mlp_weight_floats(128), device=dev)
        _lib.check(L.idh_pack_mlp_weight(w2.data_ptr(), w2p.data_ptr(), 128, 0, 128, st), "idh_pack_mlp_weight")
    vecs = torch.zeros(6, 128, device=dev)
    vecs[0] = l1.bias.detach()
    vecs[1] = w1[:, 0]
    if use_prior:
        vecs[2] = w1[:, n_feat + 1]
    vecs[3] = l2.bias.detach()
    vecs[4] = l3.weight.detach()[0]
    vecs[5, 0] = l3.bias.detach()[0]
    out = (w1p, w2p, vecs.contiguous())
    seq.__dict__["_idh_mlp"] = (key, out)
    return out


def occlusion_logits(net, feat_nhwc: torch.Tensor, feat_c0: int, n_feat: int, depth_bphw: torch.Tensor,
                     prior_bphw: Optional[torch.Tensor] = None, scale: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """feat_nhwc: dense (B,H,W,CS) buffer, features are channels [feat_c0, feat_c0+n_feat).
    Returns logits (B,P,H,W).  With a prior-enabled network and ``prior_bphw=None`` the prior
    channel is the constant -1 (reference bd_model.py:433-434)."""
    _lib.require_cuda_f32(feat_nhwc, depth_bphw, prior_bphw)
    B, H, W, CS = feat_nhwc.shape
    P = depth_bphw.shape[1]
    if tuple(depth_bphw.shape) != (B, P, H, W):
        raise _lib.IdhError(f"rendered depth {tuple(depth_bphw.shape)} does not match features {(B, H, W)}")
    math = mlp_math_of(net)
    w1p, w2p, vecs = _prepared(net.mlps[f"s{scale}"], n_feat, net.use_prior, math)
    depth = depth_bphw.contiguous()
    prior = None
    if prior_bphw is not None:
        if not net.use_prior:
            raise _lib.IdhError("prior given to a network built with use_prior=False")
        prior = prior_bphw.expand(B, P, H, W).contiguous()
    if out is None:
        out = torch.empty(B, P, H, W, device=feat_nhwc.device, dtype=torch.float32)
    L = _lib.lib()
    fwd = L.idh_binary_mlp_f16x3_fwd if math == "f16x3" else L.idh_binary_mlp_fwd
    _lib.check(
        fwd(feat_nhwc.data_ptr() + 4 * feat_c0, CS, n_feat, depth.data_ptr(), _lib.ptr(prior), int(net.use_prior), -1.0,
            w1p.data_ptr(), w2p.data_ptr(), vecs.data_ptr(), B, P, H * W, out.data_ptr(), _lib.stream_ptr()),
        "idh_binary_mlp_fwd")
    return out


def infer_depth(net, feat_nhwc: torch.Tensor, feat_c0: int, n_feat: int, prior_b1hw: Optional[torch.Tensor] = None,
                iters: int = 12, lo: float = 0.5, hi: float = 8.0, threshold: float = 0.5, thresholder=None):
    """Fused form of the reference's ``infer_depth`` loop (bd_model.py:273-292): returns
    (search_depths (B,1,H,W), logits of the last evaluation (B,1,H,W)).  ``thresholder``: an object
    with ``bins`` / ``thresholds`` tensors (metrics.Thresholder, reference binary_metrics_utils.py:42-52)
    -> per-depth thresholds as in bd_model.py:282-283; None -> the constant ``threshold``."""
    _lib.require_cuda_f32(feat_nhwc, prior_b1hw)
    B, H, W, CS = feat_nhwc.shape
    math = mlp_math_of(net)
    w1p, w2p, vecs = _prepared(net.mlps["s0"], n_feat, net.use_prior, math)
    prior = prior_b1hw.contiguous() if prior_b1hw is not None else None
    sd = torch.empty(B, 1, H, W, device=feat_nhwc.device)
    logits = torch.empty(B, 1, H, W, device=feat_nhwc.device)
    bins = thr_logits = None
    if thresholder is not None:
        dev = feat_nhwc.device
        bins = thresholder.bins.to(device=dev, dtype=torch.float32).contiguous()
        thr = thresholder.thresholds.to(device=dev, dtype=torch.float32)
        if bins.dim() != 1 or thr.shape != bins.shape or not bool(((thr > 0) & (thr < 1)).all()):
            raise _lib.IdhError("thresholder needs 1-D bins / thresholds of equal length with thresholds in (0, 1)")
        thr_logits = torch.log(thr / (1 - thr)).contiguous()
    if math == "f16x3":
        _lib.check(
            _lib.lib().idh_binary_mlp_search_f16x3_fwd(feat_nhwc.data_ptr() + 4 * feat_c0, CS, n_feat, _lib.ptr(prior), int(net.use_prior), -1.0,
                                                       w1p.data_ptr(), w2p.data_ptr(), vecs.data_ptr(), B, H * W, iters, lo, hi, threshold,
                                                       _lib.ptr(bins), _lib.ptr(thr_logits), 0 if bins is None else bins.numel(),
                                                       sd.data_ptr(), logits.data_ptr(), _lib.stream_ptr()),
            "idh_binary_mlp_search_f16x3_fwd")
        return sd, logits
    if thresholder is not None:
        _lib.check(
            _lib.lib().idh_binary_mlp_search_thr_fwd(feat_nhwc.data_ptr() + 4 * feat_c0, CS, n_feat, _lib.ptr(prior), int(net.use_prior), -1.0,
                                                     w1p.data_ptr(), w2p.data_ptr(), vecs.data_ptr(), B, H * W, iters, lo, hi,
                                                     bins.data_ptr(), thr_logits.data_ptr(), bins.numel(), sd.data_ptr(), logits.data_ptr(),
                                                     _lib.stream_ptr()),
            "idh_binary_mlp_search_thr_fwd")
        return sd, logits
    _lib.check(
        _lib.lib().idh_binary_mlp_search_fwd(feat_nhwc.data_ptr() + 4 * feat_c0, CS, n_feat, _lib.ptr(prior), int(net.use_prior), -1.0,
                                             w1p.data_ptr(), w2p.data_ptr(), vecs.data_ptr(), B, H * W, iters, lo, hi, threshold,
                                             sd.data_ptr(), logits.data_ptr(), _lib.stream_ptr()),
        "idh_binary_mlp_search_fwd")
    return sd, logits


def sample_prior(rendered_depth: torch.Tensor, prior_prediction: torch.Tensor, cur_world_T_cam: torch.Tensor,
                 prior_cam_T_world: torch.Tensor, K: torch.Tensor, invK: torch.Tensor) -> torch.Tensor:
    """BDModel.sample_prior (reference bd_model.py:395-410) on the GPU kernel; returns the prior
    channel (B,P,H,W) with -1 where invalid."""
    _lib.require_cuda_f32(rendered_depth, prior_prediction, cur_world_T_cam, prior_cam_T_world, K, invK)
    B, P, H, W = rendered_depth.shape
    if prior_prediction.shape[0] != B or tuple(prior_prediction.shape[2:]) != (H, W):
        raise _lib.IdhError(f"prior prediction {tuple(prior_prediction.shape)} does not match rendered depth {tuple(rendered_depth.shape)}")
    out = torch.empty(B, P, H, W, device=rendered_depth.device, dtype=torch.float32)
    # hold the contiguous copies in locals until the launch is enqueued
    rd, pp, cw, pc, Kc, iKc = (t.contiguous() for t in (rendered_depth, prior_prediction, cur_world_T_cam, prior_cam_T_world, K, invK))
    _lib.check(
        _lib.lib().idh_sample_prior_fwd(rd.data_ptr(), pp.data_ptr(), prior_prediction.shape[1],
                                        cw.data_ptr(), pc.data_ptr(), Kc.data_ptr(),
                                        iKc.data_ptr(), B, P, H, W, out.data_ptr(), _lib.stream_ptr()),
        "idh_sample_prior_fwd")
    return out


def binary_mlp_forward(net, inputs: List[torch.Tensor], max_scale_only: bool = False) -> Dict[str, torch.Tensor]:
    """``BinaryMLPNetwork.forward(list of (..., Cin) tensors, max_scale_only)`` (reference networks.py:106-115); row = [depth | features |
    (prior)].  fp32: the rows are read IN PLACE through ``idh_binary_mlp_strided_fwd`` whatever their strides - in particular the
    ``permute(0, 2, 3, 1)`` view of an NCHW concat that ``BDModel.run_mlp_val`` passes (bd_model.py:415-439), whose channel planes the kernel
    reads as they lie - so a module-swapped model pays no (B,H,W,65) materialisation and no copy of the feature slice per query plane."""
    scales = [0] if max_scale_only else list(net.scales)
    outs = {}
    for s in scales:
        x = inputs[s]
        _lib.require_cuda_f32(x)
        cin = x.shape[-1]
        n_feat = cin - (2 if net.use_prior else 1)
        lead = x.shape[:-1]
        math = mlp_math_of(net)
        if math == "fp32" and x.dim() == 4 and x.stride(1) == x.shape[2] * x.stride(2) and min(x.stride()) > 0:
            # (B, H, W, Cin) with a uniform pixel stride: contiguous rows (pixel stride Cin, channel stride 1) or the permuted NCHW view (1, H*W)
            B, H, W, _ = x.shape
            w1p, w2p, vecs = _prepared(net.mlps[f"s{s}"], n_feat, net.use_prior, math)
            depth = x[..., 0].reshape(B, 1, H * W).contiguous()
            prior = x[..., 1 + n_feat].reshape(B, 1, H * W).contiguous() if net.use_prior else None
            y = torch.empty(B, 1, H * W, device=x.device, dtype=torch.float32)
            _lib.check(
                _lib.lib().idh_binary_mlp_strided_fwd(x.data_ptr() + 4 * x.stride(3), x.stride(0), x.stride(2), x.stride(3), n_feat, depth.data_ptr(),
                                                      _lib.ptr(prior), int(net.use_prior), -1.0, w1p.data_ptr(), w2p.data_ptr(), vecs.data_ptr(),
                                                      B, 1, H * W, y.data_ptr(), _lib.stream_ptr()),
                "idh_binary_mlp_strided_fwd")
            outs[f"pred_{s}"] = y.reshape(*lead, 1)
            continue
        rows = x.reshape(1, -1, 1, cin)  # (B=1, H=M, W=1, Cin)
        depth = rows[..., 0].reshape(1, 1, -1, 1)
        prior = rows[..., 1 + n_feat].reshape(1, 1, -1, 1) if net.use_prior else None
        if math == "fp32":  # row stride Cin = 65 / 66 floats, features from column 1: idh_binary_mlp_fwd takes any stride since ABI 105
            y = occlusion_logits(net, rows, 1, n_feat, depth, prior, scale=s)
        else:  # the frozen split-precision kernels keep the 16-byte row alignment
            y = occlusion_logits(net, rows[..., 1 : 1 + n_feat].contiguous(), 0, n_feat, depth, prior, scale=s)
        outs[f"pred_{s}"] = y.reshape(*lead, 1)
    return outs
