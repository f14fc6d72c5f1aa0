"""Parity AT THE OPERATING POINTS THE HEADLINE IS QUOTED ON (round-5 review, "what's weak" 1): the 32-frame plan bench.py times
(F(4x4) on every layer above the tile threshold, F(2x2) on the 24x32 level, direct kernels below, liveness buffer reuse active) and
the 4-frame plan a rank of the 8-GPU run executes (BASELINE.json configs[3]: batch 32 sharded over 8 GPUs) — plans whose kernel mix and
buffer aliasing exist at no other batch size.  Two legs per batch size:

* the bench's own workload object, unmodified: every frame of the batch against the same frame run alone (the one-frame plan is the one
  tests/test_bdmodel_gpu.py / test_hot_path_head_gpu.py pin to the reference's full-size goldens);
* the reference's full-size golden frame (g5_full_bdmodel_mlp, from the layer1 map) placed in a batch of that size, compared with the
  reference's outputs directly.

Reference: experiment_modules/bd_model.py:175-311 (the forward these plans replace), test_bd.py:196-212 (the timed call).
"""
import os
import sys

import numpy as np
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import ROOT, TOL, block_err, load_golden, rel_err
from hot_helpers import holder, hot_keys, rel_poses, to_cuda

pytestmark = pytest.mark.gpu


def _census(ent):
    from implicit_depth_amd import nhwc

    plan = ent["plan"]
    convs = [op for op in plan.ops if op.kind == nhwc.OP_CONV]
    return {"convs": len(convs), "wino4": sum(op.tile_m == nhwc.TILE_WINO4 for op in convs), "wino2": sum(op.tile_m == nhwc.TILE_WINO for op in convs),
            "recycled": plan.recycled, "released": plan.recycled_candidates}


def _plan_census(model):
    return _census(next(iter(model._plans.values())))


@pytest.mark.parametrize("B", [32, 4])
def test_bench_plan_every_frame_equals_its_one_frame_run(B):
    """bench.py's HotPathWorkload, default arguments (512x384, K = 7 MLP volume, D = 64, head on, 8 query planes, return_mask), nothing forced."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench

    args = bench.parse(["--batch", str(B)])
    wl = bench.HotPathWorkload(args, torch.device("cuda", 0), 0)
    with torch.inference_mode():
        wl.step()
        wl.step()  # the replay: recycled buffers now hold the previous pass's garbage
        big = {k: v.clone() for k, v in wl.out.items() if torch.is_tensor(v)}
        census = _plan_census(wl.model)
        print(f"B={B} timed plan:", census)
        if B == 32:
            assert census["wino4"] >= 60 and census["recycled"] >= 1, census
        worst = {"pred": 0.0, "lowest": 0.0, "mask": 0.0}
        for b in range(B):
            one = wl._forward(frames=slice(b, b + 1))
            worst["pred"] = max(worst["pred"], rel_err(big["pred_0"][b:b + 1].cpu(), one["pred_0"].cpu()))
            worst["lowest"] = max(worst["lowest"], ((big["lowest_cost_bhw"][b:b + 1] - one["lowest_cost_bhw"]).abs() > 1e-5).float().mean().item())
            worst["mask"] = max(worst["mask"], (big["overall_mask_bhw"][b:b + 1] != one["overall_mask_bhw"]).float().mean().item())
        one_census = [c for c in (_census(ent) for ent in wl.model._plans.values()) if c != census]
    print(f"B={B}: worst frame vs its one-frame run: logits {worst['pred']:.2e} of scale, arg-max depth mismatch {worst['lowest']:.2e}, "
          f"mask mismatch {worst['mask']:.2e}; one-frame plan: {one_census}")
    # two different kernel selections, each within a few 1e-6 of fp64 per layer: their difference after ~130 layers
    assert worst["pred"] < 5e-5, worst
    assert worst["lowest"] < 2e-3 and worst["mask"] < 1e-4, worst  # the mask does not depend on the conv plan at all
    # and the bench's own `parity` object says the same thing about the same outputs
    with torch.inference_mode():
        par = wl.parity()
    assert par["ok"] and par["worst_frame_vs_b1_rel"] <= worst["pred"] * 1.0001 + 1e-12 and par["timed_plan"]["wino4"] == census["wino4"], par


@pytest.mark.parametrize("B", [32, 4])
def test_reference_golden_frame_inside_a_bench_sized_batch(B):
    """Frame 0 = the inputs of the reference's full-size BDModel.forward golden (seeds of tests/golden/gen_golden.py), frames 1..B-1 other
    tuples; default thresholds, default buffer reuse: the reference's own outputs must come out of the BATCHED plan directly."""
    from implicit_depth_amd.dropin import hot_path_of

    g = load_golden("g5_full_bdmodel_mlp")
    K, Hi, Wi, D, Pq = [int(v) for v in g["dims"]]
    h = holder(K, "mlp", Hi // 4, Wi // 4, D)
    assert hot_keys(h) == list(g["keys"])
    h.cuda()
    # numpy's generator fills in order: the first frame of a B-frame draw is the one-frame draw of the same seed (checked below)
    cur, src = (to_cuda(d) for d in syn.frame_tuple(B, K, Hi, Wi, seed=31, P=Pq))
    E, P = rel_poses(cur, src)
    l1_host = syn.layer1_maps(B, K, Hi // 4, Wi // 4, seed=78)
    assert torch.equal(l1_host[:1], syn.layer1_maps(1, K, Hi // 4, Wi // 4, seed=78))
    pyr_host = syn.encoder_pyramid(B, Hi, Wi, seed=73)
    assert all(torch.equal(a[:1], b_) for a, b_ in zip(pyr_host, syn.encoder_pyramid(1, Hi, Wi, seed=73)))
    l1, pyr = l1_host.cuda(), [t.cuda() for t in pyr_host]
    hot = hot_path_of(h)
    with torch.inference_mode():
        for _ in range(2):  # second pass = replay over recycled buffers
            out = hot(None, None, pyr, E, P, src["K_s1_b44"], cur["invK_s1_b44"], rendered_depth=cur["rendered_depth"], return_mask=True,
                      matching_layer1=l1, return_matching_feats=True)
    census = _plan_census(hot)
    print(f"B={B} plan:", census)
    if B == 32:
        assert census["wino4"] >= 60 and census["recycled"] >= 1, census
    feats = torch.cat([out["matching_cur_feats"][:1, None], out["matching_src_feats"][:1]], 1).cpu()
    assert rel_err(feats[:, :, :, ::6, ::8], g["head_feats_slice"]) < TOL
    pred, low = out["pred_0"][:1].cpu(), out["lowest_cost_bhw"][:1].cpu()
    e = rel_err(pred[:, :, ::6, ::8], g["head_pred_slice"])
    be = block_err(pred, load_golden("g_full_blocks")["g5_full_bdmodel_mlp_head_pred_1x8x8"], 1, 8, 8)  # every 8x8 block of every plane
    print(f"B={B}: frame 0 vs the reference's golden: slice {e:.2e}, 8x8 block sums {be:.2e}")
    assert e < TOL and be < 5e-5, (e, be)
    s = pred.double()
    np.testing.assert_allclose([s.abs().sum().item(), (s * s).sum().item()], g["head_pred_chk"][1:], rtol=2e-4)
    assert ((low[:, ::3, ::4] - torch.as_tensor(g["head_lowest_slice"])).abs() > 1e-5).float().mean().item() < 5e-3
    assert (out["overall_mask_bhw"][:1].cpu()[:, ::3, ::4] != torch.as_tensor(g["head_mask_slice"])).float().mean().item() < 2e-3
    # the other frames are different tuples (not copies of frame 0) and finite
    assert torch.isfinite(out["pred_0"]).all() and not torch.equal(out["pred_0"][0], out["pred_0"][B - 1])


def test_warp_match_bench_batch_properties():
    """BASELINE.json configs[1] at the size bench.py times it (B = 32, K = 8, D = 64, 96x128 map: the window kernel with its run-list pre-pass,
    planes split over workgroups, arg-max combined from the scratch): size-independent properties instead of an oracle that would take minutes -
    (i) linearity: doubling the current features doubles the volume BIT FOR BIT (a power-of-two scale commutes with every fp32 rounding) and leaves
    the arg-max depth untouched; (ii) frames 0, 13 and 31 equal their one-frame runs (another kernel: cv_dot_quad_k) to 1e-5 of scale;
    (iii) a frame's volume does not depend on its neighbours in the batch (frame 5 alone in a 2-frame batch, bit for bit: same window kernel)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import volume_opts

    args = bench.parse(["--workload", "warp_match_dot"])
    w = bench.WarpMatchDot(args, torch.device("cuda", 0), 0)
    assert (w.B, w.K, w.D, w.H, w.W) == (32, 8, 64, 96, 128) and w.dominant_kernel.startswith("cv_dot_win_k")
    L, p = _lib.lib(), _lib.ptr

    def run(cur, src, Ks, E, iK, B):
        cost = torch.empty(B, w.D, w.H, w.W, device="cuda")
        low = torch.empty(B, w.H, w.W, device="cuda")
        opts, keep = volume_opts(B, w.K, w.C, w.H, w.W, w.D, dot_scratch_device=cost.device)
        _lib.check(L.idh_cost_volume_dot_ex_fwd(p(cur), p(src), p(Ks), p(E), p(iK), 0.25, 5.0, B, w.K, w.C, w.H, w.W, w.D, p(cost), 0, p(low), None, opts,
                                                _lib.stream_ptr()), "dot")
        torch.cuda.synchronize()
        return cost, low

    cost, low = run(w.cur, w.src, w.Ks, w.E, w.invK, w.B)
    cost2, low2 = run((2.0 * w.cur).contiguous(), w.src, w.Ks, w.E, w.invK, w.B)
    assert torch.equal(cost2, 2.0 * cost) and torch.equal(low2, low)
    scale = cost.abs().max().item()
    for b in (0, 13, 31):
        c1, l1 = run(w.cur[b:b + 1].contiguous(), w.src[b:b + 1].contiguous(), w.Ks[b:b + 1].contiguous(), w.E[b:b + 1].contiguous(), w.invK[b:b + 1].contiguous(), 1)
        assert (c1[0] - cost[b]).abs().max().item() < 1e-5 * scale
        assert ((l1[0] - low[b]).abs() > 1e-5).float().mean().item() < 5e-3
    sl = slice(5, 7)
    c2, l2 = run(w.cur[sl].contiguous(), w.src[sl].contiguous(), w.Ks[sl].contiguous(), w.E[sl].contiguous(), w.invK[sl].contiguous(), 2)
    assert torch.equal(c2[0], cost[5]) and torch.equal(l2[0], low[5])
