"""Drop-in cost-volume managers backed by the fused gfx950 kernels.

Same constructor arguments, attributes, buffers and ``forward`` signature/return tuple as the
reference classes of the same name (reference modules/cost_volume.py:17-366 and :369-715), so
they can be assigned over ``model.cost_volume`` exactly like the reference's own
``to_fast()`` swap (reference test_bd.py:80-81).  All arithmetic happens in
``csrc/cost_volume_dot.hip`` / ``csrc/feature_volume.hip`` behind the C ABI (include/idh.h);
torch is only used to own device memory and the stream.
"""
from __future__ import annotations

from typing import Optional

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


def _is_device_scalar(v) -> bool:
    return torch.is_tensor(v) and v.is_cuda


def volume_opts(B, K, C, H, W, D, planes_bdhw: Optional[Tensor] = None, cur_batch_stride: int = 0, src_batch_stride: int = 0, kernel: int = 0,
                dot_scratch_device=None):
    """``idh_volume_opts`` for one launch -> (ctypes struct or None, keep-alive list).  ``planes_bdhw`` is the
    reference's ``depth_planes_bdhw`` (modules/cost_volume.py:324-347): any (B,D,H,W) fp32 device tensor; views
    that are constant over the image (``expand()``ed (B,D,1,1) / (1,D,1,1), what generate_depth_planes returns)
    are passed by stride, anything else as a dense per-pixel map."""
    # dot-product volume (``dot_scratch_device`` given): scratch that lets the arg-max pass of a plane-split launch combine per-group
    # results instead of re-reading the volume (idh_volume_opts.scratch); taken from torch's stream-ordered caching allocator
    n_scratch = int(_lib.lib().idh_cost_volume_dot_scratch_floats(B, K, C, H, W, D)) if dot_scratch_device is not None else 0
    if planes_bdhw is None and not cur_batch_stride and not src_batch_stride and not kernel and not n_scratch:
        return None, []
    o = _lib.VolumeOpts()
    o.cur_batch_stride, o.src_batch_stride, o.kernel = int(cur_batch_stride), int(src_batch_stride), int(kernel)
    keep = []
    if n_scratch:
        sc = torch.empty(n_scratch, device=dot_scratch_device, dtype=torch.float32)
        o.scratch, o.scratch_floats = sc.data_ptr(), n_scratch
        keep.append(sc)
    if planes_bdhw is not None:
        pl = planes_bdhw
        _lib.require_cuda_f32(pl)
        if pl.dim() != 4 or pl.shape[1] != D or pl.shape[0] not in (1, B) or tuple(pl.shape[2:]) not in ((H, W), (1, 1)):
            raise ValueError(f"depth_planes_bdhw has shape {tuple(pl.shape)}, expected ({B},{D},{H},{W})")
        const_over_image = tuple(pl.shape[2:]) == (1, 1) or (pl.stride(2) == 0 and pl.stride(3) == 0)
        if const_over_image:
            sb = pl.stride(0) if pl.shape[0] == B and B > 1 else 0
            o.planes, o.planes_batch_stride, o.planes_plane_stride, o.planes_pixel_stride = pl.data_ptr(), sb, pl.stride(1), 0
        else:
            pl = pl.expand(B, D, H, W).contiguous()
            o.planes, o.planes_batch_stride, o.planes_plane_stride, o.planes_pixel_stride = pl.data_ptr(), D * H * W, H * W, 1
        keep.append(pl)
    return o, keep


class CostVolumeManager(nn.Module):
    """Dot-product plane-sweep cost volume (reference modules/cost_volume.py:17-366)."""

    def __init__(self, matching_height, matching_width, num_depth_bins=64, matching_dim_size=None, num_source_views=None):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        self.num_depth_bins = num_depth_bins
        self.matching_height = matching_height
        self.matching_width = matching_width
        self.register_buffer("linear_ramp_1d11", torch.linspace(0, 1, num_depth_bins).view(1, num_depth_bins, 1, 1))
        self.backprojector = BackprojectDepth(height=matching_height, width=matching_width)
        self.projector = Project3D()
        self.kernel = 0  # 0 = the launcher's choice; _lib.CV_KERNEL_* forces one of the three dot-volume kernels (tests / profiling)

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

    def _planes_arg(self, B, min_depth, max_depth, depth_planes_bdhw):
        """How the planes reach the kernel: caller-supplied planes and device-resident min/max depth
        tensors (what BDModel.forward passes, bd_model.py:226-229) go in as a device array — computed with the
        reference's own formula, no device->host read; plain numbers are expanded in the kernel.
        Returns (planes tensor or None, dmin, dmax)."""
        if depth_planes_bdhw is not None:
            return depth_planes_bdhw, 1.0, 1.0
        if _is_device_scalar(min_depth) or _is_device_scalar(max_depth):
            dev = self.linear_ramp_1d11.device
            as_t = lambda v: (v if torch.is_tensor(v) else torch.tensor(float(v))).to(device=dev, dtype=torch.float32)
            return self.generate_depth_planes(B, as_t(min_depth), as_t(max_depth)), 1.0, 1.0
        return None, float(min_depth), float(max_depth)

    def build_cost_volume(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                          depth_planes_bdhw=None, return_mask=False, cur_feats_nhwc=None, src_feats_nhwc=None):
        """Returns (cost_volume B,D,H,W ; depth_planes view ; None) like reference :221-317,
        plus the arg-max depth as a 4th element (the kernel produces it in the same pass)."""
        del src_poses, return_mask  # unused by the dot-product volume, as in the reference (:270)
        B, K, C, H, W = self._check(cur_feats, src_feats)
        D = self.num_depth_bins
        _lib.require_cuda_f32(cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK)
        cur_n = cur_feats_nhwc if cur_feats_nhwc is not None else to_nhwc(cur_feats)
        src_n = src_feats_nhwc if src_feats_nhwc is not None else to_nhwc(src_feats)
        dev = cur_feats.device
        cost = torch.empty(B, D, H, W, device=dev, dtype=torch.float32)
        lowest = torch.empty(B, H, W, device=dev, dtype=torch.float32)
        L = _lib.lib()
        planes_t, dmin, dmax = self._planes_arg(B, min_depth, max_depth, depth_planes_bdhw)
        opts, keep = volume_opts(B, K, C, H, W, D, planes_t, kernel=self.__dict__.get("kernel", 0), dot_scratch_device=dev)
        planes_d = torch.empty(D, device=dev, dtype=torch.float32) if planes_t is None else None
        # keep the contiguous copies alive until the launch is enqueued (a temporary's block may be
        # recycled by the next .contiguous() before the kernel runs)
        Ks_c, E_c, iK_c = src_Ks.contiguous(), src_extrinsics.contiguous(), cur_invK.contiguous()
        _lib.check(
            L.idh_cost_volume_dot_ex_fwd(
                _lib.ptr(cur_n), _lib.ptr(src_n), _lib.ptr(Ks_c), _lib.ptr(E_c),
                _lib.ptr(iK_c), dmin, dmax, B, K, C, H, W, D, _lib.ptr(cost), 0, _lib.ptr(lowest),
                _lib.ptr(planes_d), opts, _lib.stream_ptr()),
            "idh_cost_volume_dot_ex_fwd",
        )
        del keep
        planes = planes_t.expand(B, D, H, W) if planes_t is not None else planes_d.view(1, D, 1, 1).expand(B, D, H, W)
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


def feature_mlp_column_maps(K: int, C: int = 16, fold_mask: bool = False, layout: str = "packed7"):
    """Column index maps that re-order the reference MLP's first Linear (input layout of
    modules/cost_volume.py:681-695) into the K order csrc/feature_volume.hip builds its MFMA
    operands in.  -1 = structurally zero column.  Returns (voxel_cols, pixel_cols, pose_cols).

    ``layout="gen8"`` (fv_mlp_gen_k: K > 8 or C = 32): lane quarter q carries EIGHT slots per view group j (views q + 4j) - [valid, z, dot, ray
    angle | ray xyz, plane depth (group 0, quarter 0 only)] = two 16-column blocks per group, 2 ceil(K/4) blocks (2 x 2 for K <= 8) - so that a
    group's metadata is consumed inside its own loop iteration of the kernel.
    ``fold_mask`` (fv_mlp_k: fp32, K <= 8, C = 16): the per-view "valid" inputs are identically 1 (the reference clamps z to 1e-5 before it
    tests z > 0, geometry_utils.py:86 / cost_volume.py:216), so their weight columns belong to the bias (``feature_mlp_mask_columns``) and a
    lane quarter carries SIX values per view - [z, dot, ray angle, ray xyz] at 6j..6j+5 - in three 16-column blocks; the plane depth sits in
    quarter 3's first slot of the absent view 7 when K < 8, else alone in a fourth block (quarter 0, k-step 0)."""
    base = C * (K + 1)
    col_mask = lambda k: base + k
    col_z = lambda k: base + K + k
    col_plane = base + 2 * K
    col_dot = lambda k: base + 2 * K + 1 + k
    col_ang = lambda k: base + 3 * K + 1 + k
    base_r = base + 4 * K + 1
    col_ray = lambda k, e: base_r + 3 + 3 * k + e
    base_p = base_r + 3 * (K + 1)
    voxel = list(range(C * K))  # K * C/16 blocks: warped features, identity order
    # metadata blocks: lane quarter q carries, for each of its J views v = q + 4j, [mask, z, dot, ray angle, ray xyz] at
    # 7j..7j+6, and (quarter 0) the plane depth at 7J.  J = 2 for K <= 8 (fv_mlp_k's fixed layout), else ceil(K/4)
    # (fv_mlp_gen_k); ceil((7J+1)/4) blocks of 16 columns
    if fold_mask:
        if K > 8:
            raise ValueError("fold_mask is the layout of fv_mlp_k (K <= 8)")
        for cblk in range(4):  # (the packed blob keeps K + 4 blocks; block 3 is empty for K < 8)
            for q in range(4):
                for kk in range(4):
                    idx = 4 * cblk + kk
                    v = q + 4 * (idx // 6)
                    col = -1
                    if idx < 12 and v < K:
                        col = [col_z(v), col_dot(v), col_ang(v), col_ray(v, 0), col_ray(v, 1), col_ray(v, 2)][idx % 6]
                    elif (K < 8 and idx == 6 and q == 3) or (K == 8 and idx == 12 and q == 0):
                        col = col_plane
                    voxel.append(col)
        pixel = [C * K + i for i in range(C)]
        for q in range(4):
            for kk in range(4):
                pixel.append(base_r + kk if (q == 0 and kk < 3) else -1)
        return voxel, pixel, list(range(base_p, base_p + 3 * K))
    J = 2 if K <= 8 else -(-K // 4)
    if layout == "gen8":
        for cblk in range(2 * J):
            for q in range(4):
                for kk in range(4):
                    v, slot = q + 4 * (cblk >> 1), 4 * (cblk & 1) + kk
                    col = -1
                    if slot < 7 and v < K:
                        col = [col_mask(v), col_z(v), col_dot(v), col_ang(v), col_ray(v, 0), col_ray(v, 1), col_ray(v, 2)][slot]
                    elif slot == 7 and cblk == 1 and q == 0:
                        col = col_plane
                    voxel.append(col)
        pixel = [C * K + i for i in range(C)]
        for q in range(4):
            for kk in range(4):
                pixel.append(base_r + kk if (q == 0 and kk < 3) else -1)
        return voxel, pixel, list(range(base_p, base_p + 3 * K))
    if layout != "packed7":
        raise ValueError(layout)
    for cblk in range(-(-(7 * J + 1) // 4)):
        for q in range(4):
            for kk in range(4):
                idx = 4 * cblk + kk
                v = q + 4 * (idx // 7)
                col = -1
                if idx < 7 * J and v < K:
                    col = [col_mask(v), col_z(v), col_dot(v), col_ang(v), col_ray(v, 0), col_ray(v, 1), col_ray(v, 2)][idx % 7]
                elif idx == 7 * J and q == 0:
                    col = col_plane
                voxel.append(col)
    pixel = [C * K + i for i in range(C)]  # block 0: current-view features
    for q in range(4):
        for kk in range(4):
            pixel.append(base_r + kk if (q == 0 and kk < 3) else -1)
    pose = list(range(base_p, base_p + 3 * K))
    return voxel, pixel, pose


def feature_mlp_mask_columns(K: int, C: int = 16):
    """Columns of the first Linear that multiply the per-view "valid" inputs (identically 1: summed into the bias under ``fold_mask``)."""
    return [C * (K + 1) + k for k in range(K)]


# Arithmetic of the MLP kernels (feature volume, BinaryMLP): "fp32" = v_mfma_f32_16x16x4_f32 (default),
# "f16x3" = split precision on the f16 matrix cores (csrc/feature_volume.hip, fp32-equivalent results).
MLP_MATH_MODES = ("fp32", "f16x3")
DEFAULT_MLP_MATH = "fp32"  # process-wide default; per-module: the ``mlp_math`` attribute (dropin.convert(model, math=...))


class FeatureVolumeManager(CostVolumeManager):
    """MLP feature volume (reference modules/cost_volume.py:369-715): same constructor, ``mlp``
    attribute (state_dict keys ``mlp.net.{0,2,4}.*``) and forward contract."""

    def __init__(self, matching_height, matching_width, num_depth_bins=64, mlp_channels=None, matching_dim_size=16, num_source_views=7):
        super().__init__(matching_height, matching_width, num_depth_bins)
        from .networks import MLP

        mlp_channels = list(mlp_channels) if mlp_channels is not None else [202, 128, 128, 1]
        mlp_channels[0] = matching_dim_size * (1 + num_source_views) + 10 * num_source_views + 4  # reference :405-423
        self.matching_dim_size = matching_dim_size
        self.num_source_views = num_source_views
        self.mlp = MLP(channel_list=mlp_channels, disable_final_activation=True)
        self.mlp_math: Optional[str] = None  # None = DEFAULT_MLP_MATH

    def _math(self) -> str:
        m = DEFAULT_MLP_MATH if self.__dict__.get("mlp_math") is None else self.mlp_math
        if m not in MLP_MATH_MODES:
            raise _lib.IdhError(f"unknown MLP math mode {m!r} (expected one of {MLP_MATH_MODES})")
        return m

    # -- weights in kernel order, cached until a parameter changes ------------------------------
    def _packed(self):
        lins = [self.mlp.net[0], self.mlp.net[2], self.mlp.net[4]]
        math = self._math()
        key = (math,) + tuple((p.data_ptr(), _lib.param_version(p)) for p in self.mlp.parameters())
        c = self.__dict__.get("_idh_fv")
        if c is not None and c[0] == key:
            return c[1]
        K, C = self.num_source_views, self.matching_dim_size
        w1, w2, w3 = (l.weight.detach() for l in lins)
        _lib.require_cuda_f32(w1, w2, w3)
        if w1.shape[0] != 128 or tuple(w2.shape) != (128, 128) or tuple(w3.shape) != (1, 128) or C not in (16, 32) or K > 16:
            raise _lib.IdhError("feature-volume kernels cover matching_dim_size 16 / 32, up to 16 source views and MLP widths [*,128,128,1]")
        if math == "f16x3" and (C != 16 or K > 8):
            raise _lib.IdhError("the split-precision feature volume covers matching_dim_size 16 and up to 8 source views")
        fold = math == "fp32" and C == 16 and K <= 8  # fv_mlp_k's layout (csrc/feature_volume.hip); the f16x3 / generic kernels keep the mask columns
        generic = math == "fp32" and (K > 8 or C != 16)  # fv_mlp_gen_k
        vox, pix, pose = feature_mlp_column_maps(K, C, fold_mask=fold, layout="gen8" if generic else "packed7")
        dev = w1.device
        w1e = torch.cat([w1, torch.zeros(128, 1, device=dev)], 1)
        pick = lambda cols: w1e[:, torch.tensor([c if c >= 0 else w1.shape[1] for c in cols], device=dev)].contiguous()
        L, st = _lib.lib(), _lib.stream_ptr()

        def frag(mat):
            dst = torch.empty(L.idh_packed_mlp_weight_floats(mat.shape[1]), device=dev)
            _lib.check(L.idh_pack_mlp_weight(mat.data_ptr(), dst.data_ptr(), mat.shape[1], 0, mat.shape[1], st), "idh_pack_mlp_weight")
            return dst

        def frag16(mat):
            dst = torch.empty(L.idh_packed_mlp_weight_f16_bytes(mat.shape[1]) // 4, device=dev, dtype=torch.int32)
            _lib.check(L.idh_pack_mlp_weight_f16(mat.data_ptr(), dst.data_ptr(), mat.shape[1], 0, mat.shape[1], st), "idh_pack_mlp_weight_f16")
            return dst

        if math == "f16x3":  # kernel order of the split-precision variant: [4 metadata blocks][K view blocks]
            vox = vox[C * K:] + vox[:C * K]
            fragv = frag16
        else:
            fragv = frag

        vecs = torch.zeros(3, 128, device=dev)
        vecs[0], vecs[1], vecs[2, 0] = lins[1].bias.detach(), w3[0], lins[2].bias.detach()[0]
        b1 = lins[0].bias.detach()
        if fold:  # sum in fp64, as one rounding of the folded constant
            b1 = (b1.double() + w1[:, torch.tensor(feature_mlp_mask_columns(K, C), device=dev)].double().sum(1)).float()
        out = {"w1v": fragv(pick(vox)), "w1p": frag(pick(pix)), "pose": pick(pose), "b1": b1.contiguous(),
               "w2": fragv(w2.contiguous()), "vecs": vecs.contiguous(), "math": math}
        self.__dict__["_idh_fv"] = (key, out)
        return out

    def _run(self, cur_ptr, src_ptr, dims, src_extrinsics, src_poses, src_Ks, cur_invK, dmin, dmax, vol_ptr, vol_cs, want_mask, dev,
             planes_t=None, cur_batch_stride=0, src_batch_stride=0, scratch=None):
        """One launch of the fused kernel.  ``scratch``: a dict the caller keeps between calls (HotPath's plan
        state) so the outputs / workspace are allocated once per shape instead of once per step."""
        B, K, C, H, W = dims
        if K != self.num_source_views:
            raise _lib.IdhError(f"FeatureVolumeManager was built for {self.num_source_views} source views, got {K} (the MLP input width is fixed)")
        pk = self._packed()
        L = _lib.lib()
        D = self.num_depth_bins
        sc = scratch if scratch is not None else {}
        # per stream: two forwards of one HotPath on different HIP streams must not share the prologue's workspace
        key = ("fv", B, H, W, D, str(dev), torch.cuda.current_stream().cuda_stream)
        if sc.get("fv_key") != key:
            wsb = L.idh_feature_volume_workspace_bytes(B)
            sc.update(fv_key=key, fv_wsb=wsb, fv_ws=torch.empty(max(wsb // 4, 1), device=dev), fv_planes=torch.empty(D, device=dev))
        lowest = torch.empty(B, H, W, device=dev)
        mask = torch.empty(B, H, W, device=dev, dtype=torch.bool) if want_mask else None  # the kernel writes 0/1 bytes
        planes = sc["fv_planes"] if planes_t is None else None
        opts, keep = volume_opts(B, K, C, H, W, D, planes_t, cur_batch_stride, src_batch_stride)
        mats = [t if t.is_contiguous() else t.contiguous() for t in (src_Ks, src_extrinsics, src_poses, cur_invK)]  # alive until enqueued
        _lib.check(
            L.idh_feature_volume_ex_fwd(cur_ptr, src_ptr, mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr(), mats[3].data_ptr(),
                                        dmin, dmax, B, K, C, H, W, D,
                                        pk["w1v"].data_ptr(), pk["w1p"].data_ptr(), pk["pose"].data_ptr(), pk["b1"].data_ptr(), pk["w2"].data_ptr(),
                                        pk["vecs"].data_ptr(), vol_ptr, vol_cs, lowest.data_ptr(),
                                        _lib.ptr(mask), _lib.ptr(planes), sc["fv_ws"].data_ptr(), sc["fv_wsb"], int(pk["math"] == "f16x3"), opts,
                                        _lib.stream_ptr()),
            "idh_feature_volume_ex_fwd")
        del keep
        return lowest, planes, mask

    def build_cost_volume(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                          depth_planes_bdhw=None, return_mask=False, cur_feats_nhwc=None, src_feats_nhwc=None):
        B, K, C, H, W = self._check(cur_feats, src_feats)
        _lib.require_cuda_f32(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK)
        cur_n = cur_feats_nhwc if cur_feats_nhwc is not None else to_nhwc(cur_feats)
        src_n = src_feats_nhwc if src_feats_nhwc is not None else to_nhwc(src_feats)
        D = self.num_depth_bins
        vol = torch.empty(B, D, H, W, device=cur_feats.device)
        planes_t, dmin, dmax = self._planes_arg(B, min_depth, max_depth, depth_planes_bdhw)
        lowest, planes, mask = self._run(cur_n.data_ptr(), src_n.data_ptr(), (B, K, C, H, W), src_extrinsics, src_poses, src_Ks, cur_invK,
                                         dmin, dmax, vol.data_ptr(), 0, return_mask, cur_feats.device, planes_t=planes_t)
        planes_v = planes_t.expand(B, D, H, W) if planes_t is not None else planes.clone().view(1, D, 1, 1).expand(B, D, H, W)
        return vol, planes_v, mask, lowest

    def fused_into(self, cv_in, state, feats, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK, dmin, dmax, return_mask):
        """Pipeline entry: writes the volume NHWC into the CVEncoder's input buffer.  ``feats`` =
        (cur pointer, src pointer, (B,K,C,H,W), cur batch stride, src batch stride) of NHWC matching features."""
        cur_ptr, src_ptr, dims, cbs, sbs = feats
        lowest, _, mask = self._run(cur_ptr, src_ptr, dims, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK, dmin, dmax,
                                    cv_in.ptr, cv_in.cs, return_mask, cv_in.buf.device, cur_batch_stride=cbs, src_batch_stride=sbs, scratch=state)
        return lowest, mask


class ZeroCostVolumeManager(CostVolumeManager):
    """Ablation volume of zeros (reference modules/cost_volume.py:1307-1384)."""

    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        B = cur_feats.shape[0]
        planes = self.generate_depth_planes(B, min_depth, max_depth)
        cost = torch.zeros(B, self.num_depth_bins, self.matching_height, self.matching_width, device=cur_feats.device)
        lowest = planes[:, 0]  # argmax of an all-zero volume is index 0
        return cost, lowest, planes, None
