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


def _err_stats(got, ref64):
    """scale-relative error statistics of a HIP volume against the fp64 oracle.  In this geometry single samples sit exactly on an
    image border or on the z = 1e-5 clamp, where fp32 and fp64 projections legitimately pick different taps (|u| reaches 1e5 px behind
    the camera), so the bar is statistical: a logic error (dropped run, misplaced window, wrong fall-back) corrupts whole planes or tiles."""
    scale = float(ref64.abs().max().clamp_min(1e-6))
    e = (got.double().cpu() - ref64).abs().flatten() / scale
    return {"max": float(e.max()), "p999": float(torch.quantile(e[:: max(1, e.numel() // 2_000_000)], 0.999)), "mean": float(e.mean()),
            "frac_gt_1e-3": float((e > 1e-3).double().mean())}


def _check_all_kernels_vs_fp64(inp, H, W, D, tag):
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import CostVolumeManager
    from oracle import cost_volume as ocv

    cpu64 = {k: (v.double().cpu() if torch.is_tensor(v) else v) for k, v in inp.items()}
    ref, rlow, _ = ocv.cost_volume_dot(cpu64["cur_feats"], cpu64["src_feats"], cpu64["src_extrinsics"], cpu64["src_Ks"], cpu64["cur_invK"],
                                       inp["min_depth"], inp["max_depth"], D)
    out = {}
    for name, code in (("quad", _lib.CV_KERNEL_QUAD), ("window", _lib.CV_KERNEL_WINDOW), ("lane", _lib.CV_KERNEL_LANE)):
        m = CostVolumeManager(H, W, D).cuda()
        m.kernel = code
        got, glow, _, _ = m(**inp)
        assert bool(torch.isfinite(got).all()), (tag, name)
        st = _err_stats(got, ref)
        out[name] = st
        assert st["p999"] < 2e-4 and st["mean"] < 2e-5 and st["frac_gt_1e-3"] < 5e-4, (tag, name, st)
        assert float(((glow.double().cpu() - rlow).abs() > 1e-5).double().mean()) < 2e-2, (tag, name, "arg-max plane")
    return out


@pytest.mark.parametrize("seed", range(6))
def test_every_kernel_vs_fp64_oracle_on_random_geometry(seed):
    """the same random geometries, each of the three kernels against the INDEPENDENT fp64 restatement (oracle/cost_volume.py: explicit
    homography, hand-rolled gather) — the window-vs-quad comparison above shares the projection code, so a common-mode error would pass it"""
    rng = np.random.default_rng(1000 + seed)  # the seeds (and so the first cases) of the HIP-vs-HIP test
    for case in range(8):  # (8 of its 12 geometries per seed: the fp64 oracle on the host is what this test costs - 48 geometries in ~100 s)
        B, K = int(rng.integers(1, 4)), int(rng.integers(1, 9))
        H, W, D = int(rng.integers(12, 70)), int(rng.integers(48, 150)), int(rng.integers(1, 70))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        inp = _case(rng, B, K, H, W)
        inp["cur_feats"] = torch.randn(B, 16, H, W, generator=g).cuda()
        inp["src_feats"] = torch.randn(B, K, 16, H, W, generator=g).cuda()
        lo = float(rng.uniform(0.1, 1.0))
        inp["min_depth"], inp["max_depth"] = lo, lo * float(rng.uniform(2.0, 40.0))
        _check_all_kernels_vs_fp64(inp, H, W, D, (seed, case, B, K, H, W, D))


@pytest.mark.parametrize("seed", range(12))
def test_source_camera_plane_cuts_through_tiles(seed):
    """z crosses 0 INSIDE a tile: the source camera looks across the swept volume (rotation of 70..110 degrees about the x or y axis) and is
    placed so that its z = 0 plane passes through the centre of the current frustum at the geometric-mean depth — every 32 x 8 tile near
    the crossing holds samples in front of the camera, behind it and on the 1e-5 clamp (geometry_utils.py:86, cost_volume.py:216-217)."""
    rng = np.random.default_rng(7000 + seed)
    B, K = 1 + seed % 2, int(rng.integers(2, 8))
    H, W, D = int(rng.integers(24, 72)), int(rng.integers(64, 150)), int(rng.integers(8, 64))
    inp = _case(rng, B, K, H, W)
    lo = float(rng.uniform(0.2, 1.0))
    hi = lo * float(rng.uniform(3.0, 20.0))
    E = inp["src_extrinsics"].cpu().double().numpy()
    invK = inp["cur_invK"].cpu().double().numpy()
    for b in range(B):
        for k in range(0, K, 2):  # every other view is a crossing view, the rest stay random
            axis = np.array([0.0, 1.0, 0.0]) if rng.integers(0, 2) else np.array([1.0, 0.0, 0.0])
            R = _rot(axis + 0.05 * rng.standard_normal(3), rng.uniform(1.22, 1.92) * (1 if rng.integers(0, 2) else -1))
            px = np.array([W * rng.uniform(0.3, 0.7), H * rng.uniform(0.3, 0.7), 1.0])
            X = math.sqrt(lo * hi) * (invK[b, :3, :3] @ px)
            t = -R @ X  # the point lands on the source camera's centre: z (and x, y) change sign around it
            t[:2] += rng.standard_normal(2) * 0.3
            E[b, k, :3, :3], E[b, k, :3, 3] = R, t
    g = torch.Generator().manual_seed(seed)
    inp["src_extrinsics"] = torch.tensor(E, dtype=torch.float32).cuda().contiguous()
    inp["src_poses"] = torch.tensor(np.linalg.inv(E), dtype=torch.float32).cuda().contiguous()
    inp["cur_feats"] = torch.randn(B, 16, H, W, generator=g).cuda()
    inp["src_feats"] = torch.randn(B, K, 16, H, W, generator=g).cuda()
    inp["min_depth"], inp["max_depth"] = lo, hi
    _check_all_kernels_vs_fp64(inp, H, W, D, ("z-cross", seed, B, K, H, W, D))
