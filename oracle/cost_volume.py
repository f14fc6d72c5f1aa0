"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's plane-sweep cost volumes, written from the arithmetic
spec in SURVEY.md §8(a') — explicit homography + hand-rolled bilinear gather, no
``grid_sample``, no per-plane module calls.  dtype-generic: run it in fp32 to mirror the
reference's precision, or in fp64 to obtain a "truth" both the reference and the HIP
kernels are measured against.

Pinned against the reference: ``tests/golden/gen_golden.py`` imports the reference's own
``CostVolumeManager`` / ``FeatureVolumeManager`` in the build container and stores their
outputs; ``tests/test_oracle_golden.py`` checks this file against those vectors.

Reference anchors (all relative to /root/reference):
  depth planes          modules/cost_volume.py:98-132 (log-spaced, linspace ramp :67)
  back-projection       utils/geometry_utils.py:55-63  (pixel centres at +0.5, :34-52)
  projection + z clamp  utils/geometry_utils.py:77-89  (eps = 1e-5)
  grid normalisation    modules/cost_volume.py:190     (uv = 2*pix/size - 1)
  bilinear, zeros pad   modules/cost_volume.py:192-198 (align_corners=False)
  dot + mask + sum_k    modules/cost_volume.py:302-311
  argmax -> depth       modules/cost_volume.py:319-322, 352-356
  bounds mask           modules/cost_volume.py:75-96
  MLP feature vector    modules/cost_volume.py:505-699 (channel order :681-695)
  pose distance         utils/geometry_utils.py:183-195
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


def depth_planes(min_depth: float, max_depth: float, D: int, dtype=torch.float32) -> torch.Tensor:
    """d_i = exp(log dmin + log(dmax/dmin) * i/(D-1)); cost_volume.py:98-132."""
    ramp = torch.linspace(0, 1, D, dtype=dtype)
    lo = torch.tensor(min_depth, dtype=dtype)
    hi = torch.tensor(max_depth, dtype=dtype)
    return torch.exp(torch.log(lo) + torch.log(hi / lo) * ramp)


def _pixel_rays(invK_b44: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """r[b,:,y,x] = invK[:3,:3] @ (x+.5, y+.5, 1); geometry_utils.py:34-52, :60."""
    dt = invK_b44.dtype
    ys, xs = torch.meshgrid(
        torch.arange(H, dtype=dt) + 0.5, torch.arange(W, dtype=dt) + 0.5, indexing="ij"
    )
    pix = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(3, -1)  # 3,N
    return torch.matmul(invK_b44[:, :3, :3], pix)  # B,3,N


def project_plane(
    rays_b3N: torch.Tensor, depth: torch.Tensor, P_bk34: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """World points at ``depth`` projected into each source view.

    Returns (X b3N, u bkN, v bkN, z bkN) with z clamped at 1e-5 (geometry_utils.py:86).
    """
    X = rays_b3N * depth  # B,3,N
    cam = torch.einsum("bkij,bjn->bkin", P_bk34[..., :3], X) + P_bk34[..., 3:4]  # B,K,3,N
    z = torch.clamp_min(cam[:, :, 2], 1e-5)
    return X, cam[:, :, 0] / z, cam[:, :, 1] / z, z


# When True the gather below is delegated to torch's grid_sample primitive (the op the reference
# itself calls) instead of the hand-rolled taps.  Same result (tests/test_oracle_golden.py); it
# exists so that the CPU *baseline* timed by bench.py is not handicapped by a slow restatement.
FAST_GATHER = False


def bilinear_zeros(src_bkchw: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Sample src at pixel-unit coords (u,v) exactly as grid_sample(bilinear, zeros,
    align_corners=False) does after the reference's normalisation: array coords
    sx = ((2u/W-1)+1)*W/2 - 0.5, taps outside the image contribute 0.

    src (B,K,C,H,W); u,v (B,K,N) -> (B,K,C,N)
    """
    B, K, C, H, W = src_bkchw.shape
    dt = src_bkchw.dtype
    gx = 2 * u * torch.tensor(1.0 / W, dtype=dt) - 1
    gy = 2 * v * torch.tensor(1.0 / H, dtype=dt) - 1
    if FAST_GATHER:
        import torch.nn.functional as F

        N = u.shape[-1]
        grid = torch.stack([gx, gy], -1).reshape(B * K, 1, N, 2)
        out = F.grid_sample(src_bkchw.reshape(B * K, C, H, W), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        return out.reshape(B, K, C, N)
    sx = ((gx + 1) * W - 1) / 2
    sy = ((gy + 1) * H - 1) / 2
    x0f, y0f = torch.floor(sx), torch.floor(sy)
    fx, fy = sx - x0f, sy - y0f
    flat = src_bkchw.reshape(B, K, C, H * W)
    out = torch.zeros(B, K, C, u.shape[-1], dtype=dt)
    for dy, dx, w in (
        (0, 0, (1 - fx) * (1 - fy)),
        (0, 1, fx * (1 - fy)),
        (1, 0, (1 - fx) * fy),
        (1, 1, fx * fy),
    ):
        xf, yf = x0f + dx, y0f + dy
        ok = (xf >= 0) & (xf <= W - 1) & (yf >= 0) & (yf <= H - 1)  # float test first
        xi = torch.where(ok, xf, torch.zeros_like(xf)).long()
        yi = torch.where(ok, yf, torch.zeros_like(yf)).long()
        idx = (yi * W + xi).unsqueeze(2).expand(B, K, C, -1)
        tap = torch.gather(flat, 3, idx)
        out = out + tap * torch.where(ok, w, torch.zeros_like(w)).unsqueeze(2)
    return out


def _P(src_Ks: torch.Tensor, src_extrinsics: torch.Tensor) -> torch.Tensor:
    return torch.matmul(src_Ks, src_extrinsics)[:, :, :3, :]  # geometry_utils.py:82


def cost_volume_dot(
    cur_feats: torch.Tensor,
    src_feats: torch.Tensor,
    src_extrinsics: torch.Tensor,
    src_Ks: torch.Tensor,
    cur_invK: torch.Tensor,
    min_depth: float,
    max_depth: float,
    D: int,
    planes_bdhw: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CostVolumeManager.forward restated (cost_volume.py:221-358).

    Returns (cost_volume B,D,H,W ; lowest_cost B,H,W ; planes D).  ``planes_bdhw``: the caller-supplied
    ``depth_planes_bdhw`` of cost_volume.py:280-288 (per-pixel planes, used instead of the log-spaced ones);
    then ``lowest`` gathers from it (:321, :352-356) and the third result is that tensor.
    """
    B, K, C, H, W = src_feats.shape
    dt = cur_feats.dtype
    planes = depth_planes(min_depth, max_depth, D, dt) if planes_bdhw is None else None
    rays = _pixel_rays(cur_invK, H, W)
    P = _P(src_Ks, src_extrinsics)
    cur = cur_feats.reshape(B, 1, C, H * W)
    cost = torch.empty(B, D, H * W, dtype=dt)
    for i in range(D):
        _, u, v, z = project_plane(rays, planes[i] if planes_bdhw is None else planes_bdhw[:, i].reshape(B, 1, H * W).to(dt), P)
        warped = bilinear_zeros(src_feats, u, v)
        mask = (z > 0).to(dt)  # always 1: z is already clamped
        cost[:, i] = ((warped * cur).sum(2) * mask).sum(1)
    cost = cost.view(B, D, H, W)
    if planes_bdhw is not None:
        return cost, torch.gather(planes_bdhw.to(dt), 1, torch.argmax(cost, 1, keepdim=True))[:, 0], planes_bdhw
    lowest = planes[torch.argmax(cost, 1)]
    return cost, lowest, planes


def pose_distance(pose_b44: torch.Tensor):
    """DVMVS pose distance (geometry_utils.py:183-195)."""
    R, t = pose_b44[..., :3, :3], pose_b44[..., :3, 3]
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    r_m = torch.sqrt(2 * (1 - torch.clamp_max(tr, 3.0) / 3))
    t_m = torch.sqrt((t * t).sum(-1))
    return torch.sqrt(t_m**2 + r_m**2), r_m, t_m


def _unit(x: torch.Tensor, dim: int, eps: float = 1e-12) -> torch.Tensor:
    return x / torch.clamp_min(torch.sqrt((x * x).sum(dim, keepdim=True)), eps)


def feature_vector(
    cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, depth, return_mask=False
) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """The (B, 16(K+1)+10K+4, N) MLP input of one plane, channel order of cost_volume.py:681-695."""
    B, K, C, H, W = src_feats.shape
    dt = cur_feats.dtype
    N = H * W
    rays = _pixel_rays(cur_invK, H, W)
    P = _P(src_Ks, src_extrinsics)
    X, u, v, z = project_plane(rays, depth, P)
    warped = bilinear_zeros(src_feats, u, v)  # B,K,C,N
    mask_b = z > 0
    mask = mask_b.to(dt)
    cur = cur_feats.reshape(B, C, N)
    dots = (warped * cur.unsqueeze(1)).sum(2) * mask  # B,K,N
    cur_ray = _unit(X, 1)  # B,3,N   (F.normalize, cost_volume.py:618)
    src_ray = _unit(X.unsqueeze(1) - src_poses[:, :, :3, 3].unsqueeze(-1), 2)  # B,K,3,N
    # cosine similarity of unit vectors, normalised explicitly (SURVEY.md §8a' note)
    num = (cur_ray.unsqueeze(1) * src_ray).sum(2)
    den = torch.clamp_min(torch.sqrt((cur_ray**2).sum(1)).unsqueeze(1), 1e-5) * torch.clamp_min(
        torch.sqrt((src_ray**2).sum(2)), 1e-5
    )
    angle = num / den
    pd, rm, tm = pose_distance(src_poses)  # B,K
    ex = lambda t: t.unsqueeze(-1).expand(B, K, N)
    vec = torch.cat(
        [
            warped.reshape(B, K * C, N),
            cur,
            mask,
            z,
            (depth.to(dt).expand(B, 1, N) if torch.is_tensor(depth) and depth.dim() == 3 else torch.full((B, 1, N), float(depth), dtype=dt)),
            dots,
            angle,
            cur_ray,
            src_ray.reshape(B, 3 * K, N),
            ex(pd),
            ex(rm),
            ex(tm),
        ],
        1,
    )
    overall = None
    if return_mask:
        inb = (u > 2) & (u < W - 2) & (v > 2) & (v < H - 2)
        overall = (mask_b.any(1) & inb.any(1)).view(B, H, W)
    return vec, overall


def mlp_forward(x_last: torch.Tensor, weights: Dict[str, torch.Tensor], slope: float = 0.01) -> torch.Tensor:
    """Linear/LeakyReLU stack of modules/networks.py:218-233 (final activation disabled
    at cost_volume.py:426); ``weights`` uses the reference's key names ``net.{0,2,4}.*``."""
    keys = sorted({int(k.split(".")[1]) for k in weights})
    h = x_last
    for j, li in enumerate(keys):
        h = torch.matmul(h, weights[f"net.{li}.weight"].t()) + weights[f"net.{li}.bias"]
        if j != len(keys) - 1:
            h = torch.where(h >= 0, h, h * slope)
    return h


def feature_volume(
    cur_feats,
    src_feats,
    src_extrinsics,
    src_poses,
    src_Ks,
    cur_invK,
    min_depth: float,
    max_depth: float,
    D: int,
    mlp_weights: Dict[str, torch.Tensor],
    return_mask: bool = False,
    planes_bdhw: Optional[torch.Tensor] = None,
):
    """FeatureVolumeManager.forward restated (cost_volume.py:437-706, 324-358); ``planes_bdhw`` as in
    ``cost_volume_dot``."""
    B, K, C, H, W = src_feats.shape
    dt = cur_feats.dtype
    planes = depth_planes(min_depth, max_depth, D, dt) if planes_bdhw is None else None
    vol = torch.empty(B, D, H * W, dtype=dt)
    overall = None
    for i in range(D):
        vec, m = feature_vector(
            cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK,
            planes[i] if planes_bdhw is None else planes_bdhw[:, i].reshape(B, 1, H * W).to(dt), return_mask
        )
        if m is not None:
            overall = m  # overwritten every plane: the LAST plane's mask survives (:603-615)
        vol[:, i] = mlp_forward(vec.transpose(1, 2), mlp_weights)[..., 0]
    vol = vol.view(B, D, H, W)
    if planes_bdhw is not None:
        return vol, torch.gather(planes_bdhw.to(dt), 1, torch.argmax(vol, 1, keepdim=True))[:, 0], planes_bdhw, overall
    lowest = planes[torch.argmax(vol, 1)]
    return vol, lowest, planes, overall
