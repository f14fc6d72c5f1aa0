"""HIP fused warp+match kernel vs the oracle, the golden vectors, and size-independent
properties at BASELINE.json's full size.  Calls go through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err
from oracle import cost_volume as ocv

pytestmark = pytest.mark.gpu


def _run(inp, D, kernel=0):
    from implicit_depth_amd.cost_volume import CostVolumeManager

    B, K, C, H, W = inp["src_feats"].shape
    m = CostVolumeManager(H, W, D).cuda()
    m.kernel = kernel
    dev = {k: v.cuda() for k, v in inp.items()}
    cv, low, planes, mask = m(**dev)
    torch.cuda.synchronize()
    assert mask is None and planes.shape == (B, D, H, W)
    return cv.cpu(), low.cpu(), planes[0, :, 0, 0].cpu()


def _lowest_mismatch(low, ref_low):
    return ((low.double() - torch.as_tensor(ref_low).double()).abs() > 1e-5).float().mean().item()


@pytest.mark.parametrize("name", ["g1_small", "g1_b2k7", "g1_ragged"])
def test_matches_reference_golden(name):
    g = load_golden(name)
    B, K, C, H, W, D, seed, bv, rv = [int(v) for v in g["dims"]]
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv)
    cv, low, planes = _run(inp, D)
    assert rel_err(cv, g["cost_volume"]) < TOL
    assert rel_err(planes, g["planes"]) < 1e-6
    assert _lowest_mismatch(low, g["lowest_cost"]) < 5e-3


# C = 32 / 64: matching_feature_dims beyond the shipped 16 (options.py:138) run on the one-lane-per-sample kernel
@pytest.mark.parametrize("shape", [(1, 1, 16, 8, 8, 1), (2, 3, 16, 17, 23, 7), (1, 8, 16, 33, 31, 9), (3, 7, 16, 24, 32, 64), (1, 16, 16, 12, 20, 5),
                                   (2, 3, 32, 17, 23, 7), (1, 8, 32, 48, 64, 16), (1, 2, 64, 9, 11, 3)])
def test_matches_oracle_fp64(shape):
    B, K, C, H, W, D = shape
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed=B + K, behind_view=K - 1 if K > 2 else -1, big_rotation_view=0 if K > 3 else -1)
    cv, low, planes = _run(inp, D)
    d = {k: v.double() for k, v in inp.items()}
    ref, rlow, rplanes = ocv.cost_volume_dot(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_Ks"], d["cur_invK"], 0.25, 5.0, D)
    assert rel_err(cv, ref) < TOL
    assert rel_err(planes, rplanes) < 1e-6
    assert _lowest_mismatch(low, rlow) < 5e-3


def test_full_size_golden_and_properties():
    """BASELINE config 2 (96x128, K=8, D=64): reference checksums/slices + linearity."""
    g = load_golden("g1_full_k8d64")
    B, K, C, H, W, D = [int(v) for v in g["dims"][:6]]
    inp = syn.cost_volume_inputs(B, K, C, H, W, 0)
    cv, low, _ = _run(inp, D)
    assert rel_err(cv[:, ::4, ::6, ::8], g["cost_slice"]) < TOL
    s = cv.double()
    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["cost_chk"][1:], rtol=1e-4)
    assert _lowest_mismatch(low[:, ::3, ::4], g["lowest_slice"]) < 5e-3
    # linearity in the current features: cv(a*cur) == a*cv(cur); additivity over source views
    inp2 = dict(inp)
    inp2["cur_feats"] = inp["cur_feats"] * 2.0
    cv2, _, _ = _run(inp2, D)
    assert rel_err(cv2, 2 * cv) < 1e-6
    parts = torch.zeros_like(cv)
    for k0, k1 in ((0, 3), (3, 8)):
        sub = dict(inp)
        for key in ("src_feats", "src_extrinsics", "src_poses", "src_Ks"):
            sub[key] = inp[key][:, k0:k1].contiguous()
        parts += _run(sub, D)[0]
    assert rel_err(parts, cv) < 1e-5
    # zero source features -> zero cost; lowest = first plane (argmax of zeros is index 0)
    z = dict(inp)
    z["src_feats"] = torch.zeros_like(inp["src_feats"])
    cvz, lowz, planes = _run(z, D)
    assert cvz.abs().max().item() == 0.0
    assert torch.all(lowz == planes[0])


def test_identity_pose_is_self_correlation():
    """Source == current view with identity relative pose: every plane samples the pixel
    centre exactly, so cost[b,d,y,x] = K * |cur[b,:,y,x]|^2 for all d."""
    B, K, C, H, W, D = 1, 2, 16, 16, 24, 6
    inp = syn.cost_volume_inputs(B, K, C, H, W, 3)
    eye = torch.eye(4).expand(B, K, 4, 4).contiguous()
    inp["src_extrinsics"], inp["src_poses"] = eye, eye
    inp["src_feats"] = inp["cur_feats"].unsqueeze(1).expand(B, K, C, H, W).contiguous()
    cv, _, _ = _run(inp, D)
    expect = K * (inp["cur_feats"] ** 2).sum(1, keepdim=True).expand(B, D, H, W)
    assert rel_err(cv, expect) < 1e-4


def test_layout_round_trip():
    from implicit_depth_amd.cost_volume import to_nchw, to_nhwc

    for shape in [(2, 3, 16, 9, 11), (1, 24, 7, 5), (3, 70, 13, 17)]:
        x = syn.randn(shape, 5, "lay").cuda()
        n = to_nhwc(x)
        assert torch.equal(n.cpu(), x.cpu().movedim(-3, -1).contiguous())
        assert torch.equal(to_nchw(n).cpu(), x.cpu())


def test_zero_cost_volume_manager():
    from implicit_depth_amd.cost_volume import ZeroCostVolumeManager

    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(1, 2, 16, 8, 8, 0).items()}
    cv, low, planes, mask = ZeroCostVolumeManager(8, 8, 4).cuda()(**inp)
    assert cv.abs().max().item() == 0 and mask is None and low.shape == (1, 8, 8)


def test_non_contiguous_matrix_arguments():
    """Regression: several non-contiguous (e.g. torch.linalg.inv column-major) arguments in one
    call must each keep their own contiguous copy alive until the launch is enqueued."""
    from implicit_depth_amd.cost_volume import CostVolumeManager

    B, K, C, H, W, D = 2, 3, 16, 12, 16, 4
    inp = syn.cost_volume_inputs(B, K, C, H, W, 1)
    ref = ocv.cost_volume_dot(inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_Ks"], inp["cur_invK"], 0.25, 5.0, D)[0]
    dev = {k: v.cuda() for k, v in inp.items()}
    for key in ("src_extrinsics", "src_Ks", "cur_invK", "src_poses"):
        dev[key] = dev[key].transpose(-1, -2).contiguous().transpose(-1, -2)  # same values, column-major strides
        assert not dev[key].is_contiguous()
    cv = CostVolumeManager(H, W, D).cuda()(**dev)[0]
    assert rel_err(cv.cpu(), ref) < TOL


def test_empty_batch_and_limits():
    """B = 0 is a no-op with correctly shaped outputs; documented limits fail loudly."""
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import CostVolumeManager, FeatureVolumeManager

    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(1, 2, 16, 8, 8, 0).items()}
    empty = {k: (v[:0] if v.shape[0] == 1 and k not in ("min_depth", "max_depth") else v) for k, v in inp.items()}
    cv, low, planes, mask = CostVolumeManager(8, 8, 4).cuda()(**empty)
    assert cv.shape == (0, 4, 8, 8) and low.shape == (0, 8, 8)
    # D at the kernel's table limit, K at IDH_MAX_SOURCE_VIEWS
    big = {k: v.cuda() for k, v in syn.cost_volume_inputs(1, 16, 16, 8, 8, 1).items()}
    cv = CostVolumeManager(8, 8, 512).cuda()(**big)[0]
    ref = ocv.cost_volume_dot(*[big[k].cpu() for k in ("cur_feats", "src_feats", "src_extrinsics", "src_Ks", "cur_invK")], 0.25, 5.0, 512)[0]
    assert rel_err(cv.cpu(), ref) < TOL
    with pytest.raises(_lib.IdhError):
        CostVolumeManager(8, 8, 513).cuda()(**big)
    # the MLP feature volume: up to IDH_MAX_SOURCE_VIEWS = 16 views (K > 8 on the generic kernel), 17 fail loudly
    k17 = {k: v.cuda() for k, v in syn.cost_volume_inputs(1, 17, 16, 8, 8, 2).items()}
    m = FeatureVolumeManager(8, 8, 4, num_source_views=17).cuda()
    with pytest.raises(_lib.IdhError):
        m(**k17)
    # matching features of 48 channels: neither volume kernel family covers them
    c48 = {k: v.cuda() for k, v in syn.cost_volume_inputs(1, 2, 48, 8, 8, 3).items()}
    with pytest.raises(_lib.IdhError):
        CostVolumeManager(8, 8, 4).cuda()(**c48)


def test_depth_range_as_numbers_device_tensors_and_per_sample_tensors():
    """min/max depth reach the kernel three ways: plain numbers (planes expanded in the kernel), the (1,1,1,1) device
    tensors BDModel.forward passes (planes computed on the device with the reference's formula — no device->host
    read), and (B,1,1,1) per-sample ranges (golden G13)."""
    from implicit_depth_amd.cost_volume import CostVolumeManager, FeatureVolumeManager

    g = load_golden("g13_custom_planes")
    B, K, C, H, W, D, seed, bv, rv = [int(v) for v in g["dims"]]
    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv).items()}
    m = CostVolumeManager(H, W, D).cuda()
    a = m(**inp)
    b = m(**dict(inp, min_depth=0.25, max_depth=5.0))
    assert rel_err(b[0].cpu(), a[0].cpu()) < 1e-6 and rel_err(b[2].cpu(), a[2].cpu()) < 1e-6
    rng = dict(inp, min_depth=torch.tensor([0.25, 0.4]).view(B, 1, 1, 1).cuda(), max_depth=torch.tensor([5.0, 3.0]).view(B, 1, 1, 1).cuda())
    cv, low, planes, _ = m(**rng)
    assert rel_err(cv.cpu(), g["range_cost_volume"]) < TOL
    assert rel_err(planes[:, :, 0, 0].cpu(), g["range_planes"]) < 1e-6
    assert _lowest_mismatch(low.cpu(), g["range_lowest"]) < 5e-3
    fm = FeatureVolumeManager(H, W, D, num_source_views=K)
    syn.fill_state_dict(fm.mlp, seed=107, gain=1.4)
    fm.cuda()
    fv, flow, _, _ = fm(**rng)
    assert rel_err(fv.cpu(), g["range_feature_volume"]) < TOL
    assert _lowest_mismatch(flow.cpu(), g["range_fv_lowest"]) < 5e-3


def test_caller_supplied_depth_planes_bdhw():
    """The reference's ``depth_planes_bdhw`` argument (modules/cost_volume.py:324-347) with per-pixel planes, both
    managers (golden G13), and an expand()ed image-constant view of the same kind generate_depth_planes returns."""
    from implicit_depth_amd.cost_volume import CostVolumeManager, FeatureVolumeManager

    g = load_golden("g13_custom_planes")
    B, K, C, H, W, D, seed, bv, rv = [int(v) for v in g["dims"]]
    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv).items()}
    planes = syn.custom_depth_planes(B, D, H, W, seed=8).cuda()
    m = CostVolumeManager(H, W, D).cuda()
    cv, low, pl, mask = m(**inp, depth_planes_bdhw=planes)
    assert mask is None and torch.equal(pl, planes)
    assert rel_err(cv.cpu(), g["cost_volume"]) < TOL
    assert _lowest_mismatch(low.cpu(), g["lowest_cost"]) < 5e-3
    fm = FeatureVolumeManager(H, W, D, num_source_views=K)
    syn.fill_state_dict(fm.mlp, seed=107, gain=1.4)
    fm.cuda()
    fv, flow, _, fmask = fm(**inp, depth_planes_bdhw=planes, return_mask=True)
    assert rel_err(fv.cpu(), g["feature_volume"]) < TOL
    assert _lowest_mismatch(flow.cpu(), g["fv_lowest"]) < 5e-3
    assert (fmask.cpu() != torch.as_tensor(g["fv_mask"])).float().mean().item() < 2e-3
    # image-constant planes passed as a stride-0 view == the same planes generated from min/max depth
    gen = m.generate_depth_planes(B, inp["min_depth"], inp["max_depth"])
    assert gen.stride(2) == 0 and gen.stride(3) == 0
    assert rel_err(m(**inp, depth_planes_bdhw=gen)[0].cpu(), m(**inp)[0].cpu()) < 1e-6
    with pytest.raises(ValueError):
        m(**inp, depth_planes_bdhw=planes[:, :-1])


# ---- the three kernels behind idh_cost_volume_dot_fwd (one lane per sample / quad-coalesced / LDS windows) ----
KERNELS = {"lane": 1, "quad": 2, "window": 3}


@pytest.mark.parametrize("kernel", ["lane", "quad", "window"])
@pytest.mark.parametrize("name", ["g1_small", "g1_b2k7", "g1_ragged", "g1_win_b2k7"])
def test_every_kernel_matches_reference_golden(name, kernel):
    """g1_b2k7 / g1_win_b2k7 have a view behind the camera and a strongly rotated one: the window kernel's per-lane
    global fallback and its skip / descend / global table modes are all exercised.  The window kernel needs a map of at
    least 48 x 12 texels; forcing it on a smaller one is an error, not a silent substitution."""
    from implicit_depth_amd import _lib

    g = load_golden(name)
    B, K, C, H, W, D, seed, bv, rv = [int(v) for v in g["dims"]]
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv)
    if kernel == "window" and (W < 48 or H < 12):
        with pytest.raises(_lib.IdhError):
            _run(inp, D, KERNELS[kernel])
        return
    cv, low, planes = _run(inp, D, KERNELS[kernel])
    assert rel_err(cv, g["cost_volume"]) < TOL
    assert _lowest_mismatch(low, g["lowest_cost"]) < 5e-3


@pytest.mark.parametrize("shape", [(2, 3, 16, 40, 70, 9), (1, 8, 16, 33, 67, 13), (3, 7, 16, 24, 64, 64), (1, 2, 16, 12, 64, 1),
                                   (2, 16, 16, 37, 90, 6), (1, 4, 16, 21, 130, 35), (1, 16, 16, 12, 48, 130), (40, 2, 16, 16, 48, 20),
                                   (12, 3, 16, 40, 160, 40), (9, 5, 16, 56, 96, 80)])
def test_window_kernel_matches_oracle_fp64(shape):
    """Ragged tiles (maps that are not multiples of 32 x 8), D not a multiple of 16 or 4, K up to 16, views behind the
    camera and strongly rotated views, batch > 1; the smallest map the kernel takes (48 x 12) with many planes (the
    launch splits the planes over workgroups and runs the separate arg-max kernel) and a batch large enough that it
    does not; 5 x 5 and 3 x 7 tile grids over many frames with 3 and 5 plane groups (the task order rotates plane groups and
    tile columns: every (tile, group) must still be computed exactly once)."""
    B, K, C, H, W, D = shape
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed=B + K, behind_view=K - 1 if K > 2 else -1, big_rotation_view=0 if K > 3 else -1)
    cv, low, planes = _run(inp, D, KERNELS["window"])
    d = {k: v.double() for k, v in inp.items()}
    ref, rlow, rplanes = ocv.cost_volume_dot(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_Ks"], d["cur_invK"], 0.25, 5.0, D)
    assert rel_err(cv, ref) < TOL
    assert _lowest_mismatch(low, rlow) < 5e-3
    # and the quad kernel agrees with it to rounding
    cvq, _, _ = _run(inp, D, KERNELS["quad"])
    assert rel_err(cv, cvq) < 2e-6


@pytest.mark.parametrize("kernel", ["lane", "quad", "window"])
def test_every_kernel_full_size_golden(kernel):
    g = load_golden("g1_full_k8d64")
    B, K, C, H, W, D = [int(v) for v in g["dims"][:6]]
    inp = syn.cost_volume_inputs(B, K, C, H, W, 0)
    cv, low, _ = _run(inp, D, KERNELS[kernel])
    assert rel_err(cv[:, ::4, ::6, ::8], g["cost_slice"]) < TOL
    from conftest import block_err

    be = block_err(cv, load_golden("g_full_blocks")["g1_full_k8d64_cost_8x8x8"], 8, 8, 8)  # every 8x8x8 block, not only the slice points
    assert be < 2e-6, be
    s = cv.double()
    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["cost_chk"][1:], rtol=1e-4)  # the plain sum cancels to 1e-3 of |.|
    assert _lowest_mismatch(low[:, ::3, ::4], g["lowest_slice"]) < 5e-3


def test_window_kernel_caller_planes_nhwc_output_and_strides():
    """The window kernel with per-pixel caller planes (never skips, per-lane fallback), the (B,H,W,D) output the
    pipeline uses, and batch-strided inputs."""
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import CostVolumeManager, to_nhwc, volume_opts

    B, K, C, H, W, D = 2, 3, 16, 40, 80, 8
    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(B, K, C, H, W, 3, 2, -1).items()}
    planes = syn.custom_depth_planes(B, D, H, W, seed=4).cuda()
    m = CostVolumeManager(H, W, D).cuda()
    ref = {}
    for name, kern in KERNELS.items():
        m.kernel = kern
        ref[name] = m(**inp, depth_planes_bdhw=planes)
    for name in ("lane", "window"):
        assert rel_err(ref[name][0].cpu(), ref["quad"][0].cpu()) < 2e-6
        assert _lowest_mismatch(ref[name][1].cpu(), ref["quad"][1].cpu()) < 5e-3
    d = {k: v.double().cpu() for k, v in inp.items()}
    oref = ocv.cost_volume_dot(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_Ks"], d["cur_invK"], 0, 0, D, planes_bdhw=planes.double().cpu())[0]
    assert rel_err(ref["window"][0].cpu(), oref) < TOL
    # one (B, K+1, H, W, C) feature buffer addressed with batch strides, NHWC output with a channel stride > D
    feats = torch.cat([to_nhwc(inp["cur_feats"])[:, None], to_nhwc(inp["src_feats"])], 1).contiguous()
    hw = H * W * C
    out = torch.zeros(B, H, W, D + 8, device="cuda")
    low = torch.empty(B, H, W, device="cuda")
    pl = torch.empty(D, device="cuda")
    for kern in (2, 3):
        opts, keep = volume_opts(B, K, C, H, W, D, None, (K + 1) * hw, (K + 1) * hw, kernel=kern)
        Ks, E, iK = (inp[k].contiguous() for k in ("src_Ks", "src_extrinsics", "cur_invK"))  # linalg.inv hands back column-major batches
        _lib.check(_lib.lib().idh_cost_volume_dot_ex_fwd(feats.data_ptr(), feats.data_ptr() + 4 * hw, Ks.data_ptr(), E.data_ptr(),
                                                         iK.data_ptr(), 0.25, 5.0, B, K, C, H, W, D, out.data_ptr(), D + 8, low.data_ptr(),
                                                         pl.data_ptr(), opts, _lib.stream_ptr()), "dot")
        m.kernel = kern
        want = m(**inp)
        assert rel_err(out[..., :D].permute(0, 3, 1, 2).cpu(), want[0].cpu()) < 1e-6
        assert float(out[..., D:].abs().max()) == 0.0
        assert torch.equal(low, want[1])


def test_argmax_scratch_gives_the_same_lowest():
    """idh_volume_opts.scratch: when the window kernel splits the planes over workgroups, the arg-max pass combines the groups'
    (best cost, plane) pairs instead of re-reading the volume — same `lowest` bit for bit (first maximum wins); a scratch that is
    too small is ignored."""
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import to_nhwc, volume_opts

    B, K, C, H, W, D = 8, 4, 16, 48, 96, 64
    L = _lib.lib()
    n = int(L.idh_cost_volume_dot_scratch_floats(B, K, C, H, W, D))
    assert n > 0  # this shape splits the planes
    assert int(L.idh_cost_volume_dot_scratch_floats(1, K, C, 24, 32, D)) == 0  # a single small frame runs on the quad kernel
    inp = {k: v.cuda() for k, v in syn.cost_volume_inputs(B, K, C, H, W, seed=3, behind_view=K - 1).items()}
    cur, src = to_nhwc(inp["cur_feats"]), to_nhwc(inp["src_feats"])
    Ks, E, iK = (inp[k].contiguous() for k in ("src_Ks", "src_extrinsics", "cur_invK"))
    res = {}
    for name, dev, shrink in (("none", None, 0), ("scratch", cur.device, 0), ("too_small", cur.device, 8)):
        opts, keep = volume_opts(B, K, C, H, W, D, kernel=_lib.CV_KERNEL_WINDOW, dot_scratch_device=dev)
        if shrink:
            opts.scratch_floats -= shrink
        cost = torch.empty(B, D, H, W, device="cuda")
        low = torch.full((B, H, W), -1.0, device="cuda")
        _lib.check(L.idh_cost_volume_dot_ex_fwd(cur.data_ptr(), src.data_ptr(), Ks.data_ptr(), E.data_ptr(), iK.data_ptr(), 0.25, 5.0, B, K, C, H, W, D,
                                                cost.data_ptr(), 0, low.data_ptr(), None, opts, _lib.stream_ptr()), "dot")
        torch.cuda.synchronize()
        res[name] = (cost, low)
    for name in ("scratch", "too_small"):
        assert torch.equal(res[name][0], res["none"][0])
        assert torch.equal(res[name][1], res["none"][1])
    assert float(res["none"][1].min()) > 0
