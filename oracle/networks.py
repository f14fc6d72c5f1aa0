"""ORACLE (test infrastructure only — never imported by the product path).

Functional CPU restatement of the conv / MLP blocks that surround the cost volume.  Every
function takes a flat ``{reference state_dict key: tensor}`` mapping (with the prefix of
the block stripped by ``sub``), so the same seeded weights drive the reference module
(golden generation), this oracle, and the HIP drop-ins.

Reference anchors (relative to /root/reference):
  BasicBlock            modules/layers.py:34-95   (bias=True, no norm, LeakyReLU 0.2)
  CVEncoder             modules/networks.py:186-215
  BDDecoderPP           modules/networks.py:20-84
  DepthDecoderPP        modules/networks.py:118-183 (1x1 heads :158-161)
  upsample              utils/generic_utils.py:94-103 (bilinear x2, align_corners=False)
  BinaryMLPNetwork      modules/networks.py:87-115  (ELU)
  run_mlp_val           experiment_modules/bd_model.py:412-449
  sample_prior          experiment_modules/bd_model.py:395-410
  matching head         modules/networks.py:279-283
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]


def sub(weights: W, prefix: str) -> W:
    p = prefix + "."
    return {k[len(p):]: v for k, v in weights.items() if k.startswith(p)}


def lrelu(x: torch.Tensor, slope: float = 0.2) -> torch.Tensor:
    return torch.where(x >= 0, x, x * slope)


def basic_block(x: torch.Tensor, w: W, stride: int = 1) -> torch.Tensor:
    """layers.py:78-95.  The residual branch is identity, a 1x1 conv (channel change, stride 1)
    or a strided 3x3 conv (stride 2) — whichever the weights say (layers.py:68-75)."""
    h = lrelu(F.conv2d(x, w["conv1.weight"], w["conv1.bias"], stride=stride, padding=1))
    h = F.conv2d(h, w["conv2.weight"], w["conv2.bias"], stride=1, padding=1)
    if "downsample.0.weight" in w:
        dw = w["downsample.0.weight"]
        pad = dw.shape[-1] // 2
        idn = F.conv2d(x, dw, w["downsample.0.bias"], stride=stride, padding=pad)
    else:
        idn = x
    return lrelu(h + idn)


def upsample2(x: torch.Tensor) -> torch.Tensor:
    """Exact x2 bilinear, align_corners=False: out[2i]=.25*in[i-1]+.75*in[i],
    out[2i+1]=.75*in[i]+.25*in[i+1] with edge clamping — written out rather than calling
    F.interpolate so the HIP kernel has an independent statement to match."""

    def up_last(t):
        lo = torch.cat([t[..., :1], t[..., :-1]], -1)
        hi = torch.cat([t[..., 1:], t[..., -1:]], -1)
        even = 0.25 * lo + 0.75 * t
        odd = 0.75 * t + 0.25 * hi
        return torch.stack([even, odd], -1).reshape(*t.shape[:-1], t.shape[-1] * 2)

    x = up_last(x)
    x = up_last(x.transpose(-1, -2)).transpose(-1, -2)
    return x


def cv_encoder(cost_volume: torch.Tensor, img_feats: List[torch.Tensor], w: W) -> List[torch.Tensor]:
    """networks.py:208-215."""
    outs = []
    x = cost_volume
    for i in range(len(img_feats)):
        x = basic_block(x, sub(w, f"convs.ds_conv_{i}"), stride=1 if i == 0 else 2)
        x = torch.cat([x, img_feats[i]], 1)
        x = basic_block(x, sub(w, f"convs.conv_{i}.0"))
        x = basic_block(x, sub(w, f"convs.conv_{i}.1"))
        outs.append(x)
    return outs


def unetpp_decoder(feats: List[torch.Tensor], w: W, depth_head: bool) -> Dict[str, torch.Tensor]:
    """networks.py:64-84 (BDDecoderPP) / :163-183 (DepthDecoderPP).

    ``output_{i}`` is a single module per scale (re-registered for each j, the last one
    wins) applied at every j; only the final j's results survive in the dict."""
    prev = list(feats)
    outputs: List[torch.Tensor] = []
    result: Dict[str, torch.Tensor] = {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            ins = [basic_block(prev[i], sub(w, f"convs.right_conv_{i}{j - 1}"))]
            ins.append(upsample2(basic_block(prev[i + 1], sub(w, f"convs.diag_conv_{i + 1}{j - 1}"))))
            if i + j != 4:
                ins.append(upsample2(basic_block(outputs[-1], sub(w, f"convs.up_conv_{i + 1}{j}"))))
            x = torch.cat(ins, 1)
            x = basic_block(x, sub(w, f"convs.in_conv_{i}{j}.0"))
            x = basic_block(x, sub(w, f"convs.in_conv_{i}{j}.conv_0"))
            outputs.append(x)
            y = x
            ow = sub(w, f"convs.output_{i}")
            if "0.conv1.weight" in ow:
                y = basic_block(y, sub(ow, "0"))
            if depth_head:
                y = F.conv2d(y, ow["1.weight"], ow["1.bias"])
                result[f"log_depth_pred_s{i}_b1hw"] = y
            else:
                result[f"feature_s{i}_b1hw"] = y
        prev = outputs[::-1]
    return result


def elu(x: torch.Tensor) -> torch.Tensor:
    return torch.where(x > 0, x, torch.expm1(x))


def binary_mlp(x_bhwc: torch.Tensor, w: W) -> torch.Tensor:
    """networks.py:98-104 for scale 0: Linear-ELU-Linear-ELU-Linear."""
    h = elu(torch.matmul(x_bhwc, w["mlps.s0.0.weight"].t()) + w["mlps.s0.0.bias"])
    h = elu(torch.matmul(h, w["mlps.s0.2.weight"].t()) + w["mlps.s0.2.bias"])
    return torch.matmul(h, w["mlps.s0.4.weight"].t()) + w["mlps.s0.4.bias"]


def occlusion_logits(
    feature_s0: torch.Tensor,
    rendered_depth_bphw: torch.Tensor,
    w: W,
    prior_bphw: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """bd_model.py:293-304 + :412-442: one BinaryMLP pass per rendered-depth plane on
    [depth, features(, prior)]; returns logits (B,P,H,W)."""
    outs = []
    for p in range(rendered_depth_bphw.shape[1]):
        parts = [rendered_depth_bphw[:, p : p + 1], feature_s0]
        if prior_bphw is not None:
            parts.append(prior_bphw[:, p : p + 1])
        x = torch.cat(parts, 1).permute(0, 2, 3, 1)
        outs.append(binary_mlp(x, w).permute(0, 3, 1, 2))
    return torch.cat(outs, 1)


def infer_depth(
    feature_s0: torch.Tensor,
    w: W,
    prior_b1hw: Optional[torch.Tensor] = None,
    bins: Optional[torch.Tensor] = None,
    thresholds: Optional[torch.Tensor] = None,
    iters: int = 12,
    lo: float = 0.5,
    hi: float = 8.0,
    return_margin: bool = False,
):
    """bd_model.py:273-292 (``infer_depth=True``): per-pixel binary search over [0.5, 8.0] m, first query
    (hi - lo) / 2 = 3.75 (:276), 12 dependent BinaryMLP evaluations; a pixel is "visible" at its query depth when
    sigmoid(logit) < threshold — 0.5, or ``thresholds[bucketize(query, bins)]`` with a Thresholder
    (utils/binary_metrics_utils.py:42-52, bd_model.py:282-283) — visible -> the query becomes the upper bound, else the
    lower bound; next query = midpoint.  Returns (search_depths, logits of the last evaluation[, smallest
    |sigmoid - threshold| seen per pixel])."""
    B, _, H, W = feature_s0.shape
    dt = feature_s0.dtype
    min_b = torch.full((B, 1, H, W), lo, dtype=dt)
    max_b = torch.full((B, 1, H, W), hi, dtype=dt)
    sd = torch.full((B, 1, H, W), (hi - lo) / 2.0, dtype=dt)
    margin = torch.full((B, 1, H, W), float("inf"), dtype=dt)
    logit = None
    for _ in range(iters):
        logit = occlusion_logits(feature_s0, sd, w, prior_b1hw)
        pred = torch.sigmoid(logit)
        thr = 0.5 if bins is None else thresholds.to(dt)[torch.bucketize(sd, bins.to(dt))]
        margin = torch.minimum(margin, (pred - thr).abs())
        vis = pred < thr
        max_b = torch.where(vis, sd, max_b)
        min_b = torch.where(vis, min_b, sd)
        sd = (max_b + min_b) / 2
    return (sd, logit, margin) if return_margin else (sd, logit)


def sample_prior(
    rendered_depth_b1hw: torch.Tensor,
    prior_prediction_b1hw: torch.Tensor,
    cur_world_T_cam: torch.Tensor,
    prior_cam_T_world: torch.Tensor,
    K_b44: torch.Tensor,
    invK_b44: torch.Tensor,
) -> torch.Tensor:
    """bd_model.py:395-410: back-project the rendered depth, project into the previous
    frame's camera, nearest-neighbour sample the previous prediction; -1 where the depth is
    not positive or the point is behind the previous camera.  grid_sample(nearest,
    zeros, align_corners=False): index = nearbyint(u - 0.5) (round-half-even), out of
    range -> 0."""
    B, _, H, W = rendered_depth_b1hw.shape
    dt = rendered_depth_b1hw.dtype
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt) + 0.5, torch.arange(W, dtype=dt) + 0.5, indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(3, -1)
    X = torch.matmul(invK_b44[:, :3, :3], pix) * rendered_depth_b1hw.reshape(B, 1, -1)
    P = torch.matmul(K_b44, torch.matmul(prior_cam_T_world, cur_world_T_cam))[:, :3]
    cam = torch.matmul(P[:, :, :3], X) + P[:, :, 3:4]
    zraw = cam[:, 2]
    z = torch.clamp_min(zraw, 1e-5)
    u, v = cam[:, 0] / z, cam[:, 1] / z
    gx = (u / W - 0.5) * 2
    gy = (v / H - 0.5) * 2
    sx = ((gx + 1) * W - 1) / 2
    sy = ((gy + 1) * H - 1) / 2
    xi, yi = torch.round(sx), torch.round(sy)  # torch.round is half-to-even like nearbyint
    ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
    idx = torch.where(ok, yi * W + xi, torch.zeros_like(xi)).long()
    val = torch.gather(prior_prediction_b1hw.reshape(B, -1), 1, idx)
    val = torch.where(ok, val, torch.zeros_like(val))
    # Project3D returns the *clamped* depth, so "cam z > 0" is always true (same quirk as the
    # cost-volume mask); only rendered_depth > 0 can invalidate a pixel.
    valid = (rendered_depth_b1hw.reshape(B, -1) > 0) & (z > 0)
    return torch.where(valid, val, torch.full_like(val, -1.0)).view(B, 1, H, W)


def instance_norm(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    m = x.mean((2, 3), keepdim=True)
    v = ((x - m) ** 2).mean((2, 3), keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


def matching_head(x_b64hw: torch.Tensor, w: W, first: int = 5) -> torch.Tensor:
    """networks.py:279-283: 1x1 conv 64->128, InstanceNorm, LeakyReLU(0.2), 3x3 conv
    128->16 with replicate padding, InstanceNorm.  ``first`` is the index of the 1x1 conv
    inside ``net`` (5 backbone entries precede it)."""
    h = F.conv2d(x_b64hw, w[f"net.{first}.weight"], w[f"net.{first}.bias"])
    h = lrelu(instance_norm(h))
    h = F.pad(h, (1, 1, 1, 1), mode="replicate")
    h = F.conv2d(h, w[f"net.{first + 3}.weight"], w[f"net.{first + 3}.bias"])
    return instance_norm(h)


def _conv_block(x: torch.Tensor, w: W) -> torch.Tensor:
    """networks_fast.py:10-28: conv3x3-ELU-conv3x3-ELU."""
    h = elu(F.conv2d(x, w["conv1.weight"], w["conv1.bias"], padding=1))
    return elu(F.conv2d(h, w["conv2.weight"], w["conv2.bias"], padding=1))


def skip_decoder(feats: List[torch.Tensor], w: W, regression: bool = False) -> Dict[str, torch.Tensor]:
    """SkipDecoder / SkipDecoderRegression (networks_fast.py:49-145): per block ConvBlock -> nearest x2
    -> cat(skip) -> ConvBlock; regression adds 1x1-ELU-1x1-ELU-1x1 heads per scale."""
    out: Dict[str, torch.Tensor] = {}
    x = feats[-1]
    for bi in range(4):
        bw = sub(w, f"block{bi + 1}")
        x = _conv_block(x, sub(bw, "pre_concat_conv"))
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)  # nearest x2
        x = torch.cat([x, feats[-2 - bi]], 1)
        x = _conv_block(x, sub(bw, "post_concat_conv"))
        out[f"feature_s{3 - bi}_b1hw"] = x
        if regression:
            hw = sub(w, f"out{bi + 1}")
            h = elu(F.conv2d(x, hw["0.weight"], hw["0.bias"]))
            h = elu(F.conv2d(h, hw["2.weight"], hw["2.bias"]))
            out[f"log_depth_pred_s{3 - bi}_b1hw"] = F.conv2d(h, hw["4.weight"], hw["4.bias"])
    return out
