"""Randomised geometry stress of the LDS-window kernel against the quad kernel (same arithmetic up to summation
order): random rotations up to ~1 rad about random axes, translations up to 2 m in any direction (views beside,
behind and inside the swept volume), random focal lengths / principal points, depth ranges and map sizes.  Guards the
window kernel's table logic — run skipping, window placement, per-lane global fall-back — where a wrong decision
would silently drop or corrupt samples."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def _case(rng, B, K, H, W):
    f = rng.uniform(0.6, 2.0) * W
    Kmat = np.eye(4)
    Kmat[0, 0] = f
    Kmat[1, 1] = f * rng.uniform(0.9, 1.1)
    Kmat[0, 2] = W * rng.uniform(0.4, 0.6)
    Kmat[1, 2] = H * rng.uniform(0.4, 0.6)
    poses = np.tile(np.eye(4), (B, K, 1, 1))
    for b in range(B):
        for k in range(K):
            mode = rng.integers(0, 4)
            ang = rng.uniform(0, 0.15) if mode == 0 else rng.uniform(0, 1.0)
            poses[b, k, :3, :3] = _rot(rng.standard_normal(3), ang)
            scale = (0.2, 0.8, 2.0, 0.05)[mode]
            poses[b, k, :3, 3] = rng.standard_normal(3) * scale
    E = np.linalg.inv(poses)
    t = lambda a: torch.tensor(a, dtype=torch.float32).cuda().contiguous()
    return {"src_extrinsics": t(E), "src_poses": t(poses), "src_Ks": t(np.tile(Kmat, (B, K, 1, 1))), "cur_invK": t(np.tile(np.linalg.inv(Kmat), (B, 1, 1)))}


@pytest.mark.parametrize("seed", range(6))
def test_window_kernel_equals_quad_kernel_on_random_geometry(seed):
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import CostVolumeManager

    rng = np.random.default_rng(1000 + seed)
    worst = 0.0
    for _ in range(12):
        B = int(rng.integers(1, 4))
        K = int(rng.integers(1, 9))
        H = int(rng.integers(12, 70))
        W = int(rng.integers(48, 150))
        D = int(rng.integers(1, 70))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        inp = _case(rng, B, K, H, W)
        inp["cur_feats"] = torch.randn(B, 16, H, W, generator=g).cuda()
        inp["src_feats"] = torch.randn(B, K, 16, H, W, generator=g).cuda()
        lo = float(rng.uniform(0.1, 1.0))
        inp["min_depth"], inp["max_depth"] = lo, lo * float(rng.uniform(2.0, 40.0))
        m = CostVolumeManager(H, W, D).cuda()
        m.kernel = _lib.CV_KERNEL_QUAD
        ref, rlow, _, _ = m(**inp)
        m.kernel = _lib.CV_KERNEL_WINDOW
        got, glow, _, _ = m(**inp)
        assert bool(torch.isfinite(got).all())
        scale = float(ref.abs().max().clamp_min(1e-6))
        err = float((got - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err < 5e-6, (seed, B, K, H, W, D, err)
        assert float(((glow - rlow).abs() > 1e-5).float().mean()) < 5e-3
    assert worst < 5e-6
