"""Deterministic synthetic inputs and weights for the cost-volume hot path.

Nothing here comes from the reference; it reproduces the *shapes and statistics* the
reference's data pipeline would hand to ``BDModel.forward``:

* matching features ~ N(0,1) (the real encoder ends in InstanceNorm,
  reference ``modules/networks.py:283``),
* ScanNet-shaped pin-hole intrinsics (fx=fy=577.87, cx=319.5, cy=239.5 at 640x480,
  reference ``datasets/scannet_dataset.py:466-486``) rescaled to the requested size,
* DVMVS-like source poses: translation (0.1(k+1), 0.02k, 0.01k) m and a y-rotation of
  0.03(k+1) rad for source view k,
* weights drawn per parameter *name* (so a reference module and its drop-in twin get
  bit-identical tensors without shipping a state_dict).

numpy's PCG64 is used instead of torch's generator so the streams are stable across
torch versions and identical on the CPU container and the GPU box.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, tag: str = "") -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(tag.encode())]))


def randn(shape, seed: int, tag: str = "", dtype=torch.float32) -> torch.Tensor:
    a = _rng(seed, tag).standard_normal(size=tuple(shape), dtype=np.float32)
    return torch.from_numpy(a).to(dtype)


def intrinsics(width: int, height: int) -> torch.Tensor:
    """4x4 pin-hole K for an image of ``width x height`` pixels (ScanNet-shaped)."""
    K = torch.eye(4, dtype=torch.float64)
    K[0, 0] = 577.87 * width / 640.0
    K[1, 1] = 577.87 * height / 480.0
    K[0, 2] = 319.5 * width / 640.0
    K[1, 2] = 239.5 * height / 480.0
    return K


def _rot_y(theta: float) -> torch.Tensor:
    c, s = math.cos(theta), math.sin(theta)
    R = torch.eye(4, dtype=torch.float64)
    R[0, 0], R[0, 2], R[2, 0], R[2, 2] = c, s, -s, c
    return R


def source_pose(k: int, behind: bool = False, big_rotation: bool = False) -> torch.Tensor:
    """cur_T_src (source camera pose expressed in the current camera frame), 4x4 fp64."""
    T = _rot_y(0.03 * (k + 1))
    if big_rotation:
        T = _rot_y(0.6)
    if behind:
        # camera looking back at the current camera from far in front of it: every
        # back-projected point lies behind this view (exercises the z<=eps clamp).
        T = _rot_y(math.pi)
        T[2, 3] = 12.0
    T[0, 3] += 0.1 * (k + 1)
    T[1, 3] += 0.02 * k
    T[2, 3] += 0.01 * k
    return T


def cost_volume_inputs(
    B: int,
    K: int,
    C: int,
    H: int,
    W: int,
    seed: int = 0,
    behind_view: int = -1,
    big_rotation_view: int = -1,
    dtype=torch.float32,
) -> Dict[str, torch.Tensor]:
    """Inputs of ``CostVolumeManager.forward`` (reference ``modules/cost_volume.py:324``)."""
    cur = randn((B, C, H, W), seed, "cur_feats")
    src = randn((B, K, C, H, W), seed, "src_feats")
    Kmat = intrinsics(W, H)
    invK = torch.linalg.inv(Kmat)
    poses = torch.stack(
        [source_pose(k, behind=(k == behind_view), big_rotation=(k == big_rotation_view)) for k in range(K)]
    )
    # small per-batch perturbation so batch elements are not copies of each other
    poses_b = []
    for b in range(B):
        P = poses.clone()
        P[:, 0, 3] += 0.013 * b
        P[:, 1, 3] -= 0.007 * b
        poses_b.append(P)
    src_poses = torch.stack(poses_b)  # B,K,4,4  (src -> cur)
    src_extr = torch.linalg.inv(src_poses)  # cur -> src
    return {
        "cur_feats": cur.to(dtype),
        "src_feats": src.to(dtype),
        "src_extrinsics": src_extr.to(dtype),
        "src_poses": src_poses.to(dtype),
        "src_Ks": Kmat.expand(B, K, 4, 4).contiguous().to(dtype),
        "cur_invK": invK.expand(B, 4, 4).contiguous().to(dtype),
        "min_depth": torch.tensor(0.25, dtype=dtype).view(1, 1, 1, 1),
        "max_depth": torch.tensor(5.0, dtype=dtype).view(1, 1, 1, 1),
    }


def fill_state_dict(module: torch.nn.Module, seed: int = 0, gain: float = 1.0) -> None:
    """Deterministically (re)initialise every parameter of ``module`` from its *name*.

    Weights ~ N(0, gain^2/fan_in), biases ~ N(0, 0.05^2): activations stay O(1) through the
    deep residual stacks, so parity errors are not hidden by vanishing/exploding scales.
    """
    with torch.no_grad():
        for name, p in module.state_dict().items():
            if not torch.is_floating_point(p) or p.ndim == 0:
                continue
            if name.endswith("weight") and p.ndim >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                w = randn(p.shape, seed, name) * (gain / math.sqrt(fan_in))
                p.copy_(w.to(p.dtype))
            elif name.endswith("bias"):
                p.copy_((randn(p.shape, seed, name) * 0.05).to(p.dtype))


def encoder_pyramid(
    B: int, img_h: int, img_w: int, seed: int = 0, channels: Iterable[int] = (24, 48, 64, 160, 256)
) -> Tuple[torch.Tensor, ...]:
    """Stand-in for the third-party image encoder's 5 feature maps (strides 2..32).

    timm's tf_efficientnetv2_s is not in the reference tree nor in this image
    (SURVEY.md §8c); only its output *shapes* (reference ``bd_model.py:47-51``) matter here.
    """
    outs = []
    for lvl, ch in enumerate(channels):
        s = 2 ** (lvl + 1)
        outs.append(randn((B, ch, img_h // s, img_w // s), seed, f"enc{lvl}") * 0.5)
    return tuple(outs)


def layer1_maps(B: int, K: int, H: int, W: int, seed: int = 0, channels: int = 64) -> torch.Tensor:
    """Stand-in for the matching backbone's output — conv1/bn1/relu/maxpool/layer1 of the third-party antialiased
    ResNet18 (reference ``modules/networks.py:262-268``): (B, K+1, 64, H, W), frame b's current image followed by its
    K source images (``bd_model.py:149-152``); non-negative like the ReLU-terminated residual block it replaces."""
    return torch.relu(randn((B, K + 1, channels, H, W), seed, "layer1"))


def rendered_depth_planes(B: int, H: int, W: int, P: int = 8) -> torch.Tensor:
    """P fronto-parallel query planes 1.5..5.0 m (reference ``generic_mvs_dataset.py:242``)."""
    d = torch.linspace(1.5, 5.0, P).view(1, P, 1, 1)
    return d.expand(B, P, H, W).contiguous()


class _FeatureInfo:
    def __init__(self, ch):
        self._ch = list(ch)

    def channels(self):
        return list(self._ch)


class StubImageEncoder(torch.nn.Module):
    """Random-init 5-level strided-conv pyramid with the output channels/strides of the
    reference's timm ``tf_efficientnetv2_s`` feature extractor (bd_model.py:47-51).  NOT the
    third-party network — a stand-in so that whole-model plumbing can run on both sides of a
    comparison with identical inputs to the hot path."""

    def __init__(self, channels=(24, 48, 64, 160, 256)):
        super().__init__()
        self._channels = list(channels)
        self.stages = torch.nn.ModuleList()
        cin = 3
        for c in channels:
            self.stages.append(torch.nn.Conv2d(cin, c, 3, 2, 1))
            cin = c
        self.feature_info = _FeatureInfo(self._channels)

    def forward(self, x):
        outs = []
        for st in self.stages:
            x = torch.tanh(st(x))
            outs.append(x)
        return outs


class StubResnetStem(torch.nn.Module):
    """Stand-in for antialiased_cnns.resnet18's conv1/bn1/relu/maxpool/layer1 (64 ch @ 1/4 res)."""

    def __init__(self):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(3, 64, 7, 2, 3)
        self.bn1 = torch.nn.Identity()
        self.relu = torch.nn.ReLU()
        self.maxpool = torch.nn.MaxPool2d(3, 2, 1)
        self.layer1 = torch.nn.Conv2d(64, 64, 3, 1, 1)


def frame_tuple(B: int, K: int, img_h: int, img_w: int, seed: int = 0, P: int = 8):
    """cur_data / src_data dictionaries with the keys BDModel.forward reads
    (bd_model.py:186-194, :266-299; produced by generic_mvs_dataset.py:742-809)."""
    Hm, Wm = img_h // 4, img_w // 4
    cvi = cost_volume_inputs(B, K, 16, Hm, Wm, seed)
    cur_pose = torch.eye(4).expand(B, 4, 4).contiguous()  # world = current camera
    src_world_T_cam = cvi["src_poses"]  # src -> world(=cur)
    K1 = intrinsics(Wm, Hm).float()
    K0 = intrinsics(img_w // 2, img_h // 2).float()
    cur = {
        "image_b3hw": randn((B, 3, img_h, img_w), seed, "cur_img"),
        "K_s1_b44": K1.expand(B, 4, 4).contiguous(), "invK_s1_b44": torch.linalg.inv(K1).expand(B, 4, 4).contiguous(),
        "K_s0_b44": K0.expand(B, 4, 4).contiguous(), "invK_s0_b44": torch.linalg.inv(K0).expand(B, 4, 4).contiguous(),
        "cam_T_world_b44": cur_pose.clone(), "world_T_cam_b44": cur_pose.clone(),
        "rendered_depth": rendered_depth_planes(B, img_h // 2, img_w // 2, P),
    }
    src = {
        "image_b3hw": randn((B, K, 3, img_h, img_w), seed, "src_img"),
        "K_s1_b44": K1.expand(B, K, 4, 4).contiguous(), "invK_s1_b44": torch.linalg.inv(K1).expand(B, K, 4, 4).contiguous(),
        "cam_T_world_b44": torch.linalg.inv(src_world_T_cam), "world_T_cam_b44": src_world_T_cam.clone(),
    }
    return cur, src


def temporal_frame(t: int, K: int, img_h: int, img_w: int, seed: int = 0):
    """Frame ``t`` of a synthetic temporal sequence (BASELINE.json config 5; the loop of reference
    ``inference/inference.py:106-157`` with the ``plane_2.0`` asset): the camera drifts along x and pans about y, the K
    source views keep the DVMVS-like relative poses of ``frame_tuple``, the query is ONE fronto-parallel plane at 2 m
    (``:128-130``), feature maps are re-drawn per frame.  Returns (cur, src, layer1 maps (1,K+1,64,h/4,w/4), encoder
    pyramid).  The prior (previous prediction + previous cam_T_world) is carried by the caller."""
    cur, src = frame_tuple(1, K, img_h, img_w, seed=seed, P=1)
    world_T_cam = _rot_y(0.01 * t)
    world_T_cam[0, 3] = 0.05 * t
    world_T_cam[2, 3] = 0.01 * t
    world_T_cam = world_T_cam.float()[None]
    src_world_T_cam = world_T_cam.unsqueeze(1) @ src["world_T_cam_b44"]  # world <- cur <- src
    cur["world_T_cam_b44"], cur["cam_T_world_b44"] = world_T_cam, torch.linalg.inv(world_T_cam)
    src["world_T_cam_b44"], src["cam_T_world_b44"] = src_world_T_cam, torch.linalg.inv(src_world_T_cam)
    cur["rendered_depth"] = torch.full((1, 1, img_h // 2, img_w // 2), 2.0)
    l1 = layer1_maps(1, K, img_h // 4, img_w // 4, seed=1000 + 7 * t + seed)
    pyr = encoder_pyramid(1, img_h, img_w, seed=2000 + 7 * t + seed)
    return cur, src, l1, pyr


def custom_depth_planes(B: int, D: int, H: int, W: int, seed: int = 0) -> torch.Tensor:
    """A caller-supplied ``depth_planes_bdhw`` (reference ``modules/cost_volume.py:324-347``): per-pixel planes —
    linearly spaced 0.4..4.5 m, tilted across the image and jittered per batch element — unlike the log-spaced,
    image-constant planes ``generate_depth_planes`` makes."""
    base = torch.linspace(0.4, 4.5, D).view(1, D, 1, 1)
    ys = torch.linspace(-1, 1, H).view(1, 1, H, 1)
    xs = torch.linspace(-1, 1, W).view(1, 1, 1, W)
    b = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1)
    return (base * (1.0 + 0.08 * xs - 0.05 * ys + 0.03 * b) + 0.01 * torch.sigmoid(randn((B, D, H, W), seed, "planes"))).contiguous()
