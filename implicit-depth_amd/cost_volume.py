"""Drop-in cost-volume managers backed by the fused gfx950 kernels.

Same constructor arguments, attributes, buffers and ``forward`` signature/return tuple as the
reference classes of the same name (reference modules/cost_volume.py:17-366 and :369-715), so
they can be assigned over ``model.cost_volume`` exactly like the reference's own
``to_fast()`` swap (reference test_bd.py:80-81).  All arithmetic happens in
``csrc/cost_volume_dot.hip`` / ``csrc/feature_volume.hip`` behind the C ABI (include/idh.h);
torch is only used to own device memory and the stream.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import _lib
from .geometry import BackprojectDepth, Project3D


def to_nhwc(x_nchw: Tensor) -> Tensor:
    """(..., C, H, W) contiguous -> (..., H, W, C) via the HIP layout kernel."""
    _lib.require_cuda_f32(x_nchw)
    x = x_nchw.contiguous()
    *lead, C, H, W = x.shape
    n_img = 1
    for s in lead:
        n_img *= s
    out = torch.empty(*lead, H, W, C, device=x.device, dtype=torch.float32)
    if out.numel():
        _lib.check(_lib.lib().idh_nchw_to_nhwc_f32(_lib.ptr(x), _lib.ptr(out), n_img, C, H * W, _lib.stream_ptr()), "idh_nchw_to_nhwc_f32")
    return out


def to_nchw(x_nhwc: Tensor) -> Tensor:
    _lib.require_cuda_f32(x_nhwc)
    x = x_nhwc.contiguous()
    *lead, H, W, C = x.shape
    n_img = 1
    for s in lead:
        n_img *= s
    out = torch.empty(*lead, C, H, W, device=x.device, dtype=torch.float32)
    if out.numel():
        _lib.check(_lib.lib().idh_nhwc_to_nchw_f32(_lib.ptr(x), _lib.ptr(out), n_img, C, H * W, _lib.stream_ptr()), "idh_nhwc_to_nchw_f32")
    return out


class CostVolumeManager(nn.Module):
    """Dot-product plane-sweep cost volume (reference modules/cost_volume.py:17-366)."""

    def __init__(self, matching_height, matching_width, num_depth_bins=64, matching_dim_size=None, num_source_views=None):
        super().__init__()
        self.num_depth_bins = num_depth_bins
        self.matching_height = matching_height
        self.matching_width = matching_width
        self.register_buffer("linear_ramp_1d11", torch.linspace(0, 1, num_depth_bins).view(1, num_depth_bins, 1, 1))
        self.backprojector = BackprojectDepth(height=matching_height, width=matching_width)
        self.projector = Project3D()

    # -- helpers ------------------------------------------------------------------------
    def _check(self, cur_feats, src_feats):
        B, K, C, H, W = src_feats.shape
        if (H, W) != (self.matching_height, self.matching_width) or tuple(cur_feats.shape) != (B, C, H, W):
            raise ValueError(
                f"feature maps {tuple(cur_feats.shape)} / {tuple(src_feats.shape)} do not match the "
                f"manager's matching size {self.matching_height}x{self.matching_width}"
            )
        return B, K, C, H, W

    def generate_depth_planes(self, batch_size: int, min_depth: Tensor, max_depth: Tensor) -> Tensor:
        """Log-spaced planes as a stride-0 (B,D,H,W) view (reference :98-132)."""
        ramp = self.linear_ramp_1d11
        planes = torch.exp(torch.log(min_depth) + torch.log(max_depth / min_depth) * ramp)
        return planes.expand(batch_size, self.num_depth_bins, self.matching_height, self.matching_width)

    def build_cost_volume(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                          depth_planes_bdhw=None, return_mask=False, cur_feats_nhwc=None, src_feats_nhwc=None):
        """Returns (cost_volume B,D,H,W ; depth_planes view ; None) like reference :221-317,
        plus the arg-max depth as a 4th element (the kernel produces it in the same pass)."""
        del src_poses, return_mask  # unused by the dot-product volume, as in the reference (:270)
        if depth_planes_bdhw is not None:
            raise _lib.IdhError("caller-supplied depth_planes_bdhw is not supported by the fused kernel (planes are log-spaced from min/max depth)")
        B, K, C, H, W = self._check(cur_feats, src_feats)
        D = self.num_depth_bins
        _lib.require_cuda_f32(cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK)
        cur_n = cur_feats_nhwc if cur_feats_nhwc is not None else to_nhwc(cur_feats)
        src_n = src_feats_nhwc if src_feats_nhwc is not None else to_nhwc(src_feats)
        dev = cur_feats.device
        cost = torch.empty(B, D, H, W, device=dev, dtype=torch.float32)
        lowest = torch.empty(B, H, W, device=dev, dtype=torch.float32)
        L = _lib.lib()
        planes_d = torch.empty(D, device=dev, dtype=torch.float32)
        dmin, dmax = float(min_depth), float(max_depth)
        _lib.check(
            L.idh_cost_volume_dot_fwd(
                _lib.ptr(cur_n), _lib.ptr(src_n), _lib.ptr(src_Ks.contiguous()), _lib.ptr(src_extrinsics.contiguous()),
                _lib.ptr(cur_invK.contiguous()), dmin, dmax, B, K, C, H, W, D, _lib.ptr(cost), 0, _lib.ptr(lowest),
                _lib.ptr(planes_d), _lib.stream_ptr()),
            "idh_cost_volume_dot_fwd",
        )
        planes = planes_d.view(1, D, 1, 1).expand(B, D, H, W)
        return cost, planes, None, lowest

    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        """(cost_volume, lowest_cost, depth_planes_bdhw, overall_mask_bhw) — reference :324-358."""
        cost, planes, mask, lowest = self.build_cost_volume(
            cur_feats=cur_feats, src_feats=src_feats, src_extrinsics=src_extrinsics, src_poses=src_poses,
            src_Ks=src_Ks, cur_invK=cur_invK, min_depth=min_depth, max_depth=max_depth,
            depth_planes_bdhw=depth_planes_bdhw, return_mask=return_mask)
        return cost, lowest, planes, mask

    def to_fast(self):
        """The fused kernel already is the fast path (reference :360-366 returns the
        all-planes-at-once manager); kept so callers that do the swap keep working."""
        return self

    @classmethod
    def from_reference(cls, ref_module: nn.Module) -> "CostVolumeManager":
        """Build the drop-in from a reference manager instance (keeps buffers / device)."""
        m = cls(ref_module.matching_height, ref_module.matching_width, ref_module.num_depth_bins)
        m.load_state_dict(ref_module.state_dict(), strict=False)
        return m.to(ref_module.linear_ramp_1d11.device)


class ZeroCostVolumeManager(CostVolumeManager):
    """Ablation volume of zeros (reference modules/cost_volume.py:1307-1384)."""

    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        B = cur_feats.shape[0]
        planes = self.generate_depth_planes(B, min_depth, max_depth)
        cost = torch.zeros(B, self.num_depth_bins, self.matching_height, self.matching_width, device=cur_feats.device)
        lowest = planes[:, 0]  # argmax of an all-zero volume is index 0
        return cost, lowest, planes, None
