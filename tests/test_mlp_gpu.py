"""Fused occlusion MLP (csrc/mlp.hip) vs reference goldens and the oracle."""
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err
from oracle import networks as onet

pytestmark = pytest.mark.gpu


def _setup(use_prior):
    from implicit_depth_amd import networks as net

    feat = torch.as_tensor(load_golden("g3_bddecoder")["s0"])
    Bq, Hq, Wq, P = 1, 48, 64, 3
    rd = syn.rendered_depth_planes(Bq, Hq, Wq, P)
    rd[:, 1, :5, :7] = 0.0
    prior = torch.sigmoid(syn.randn((Bq, 1, Hq, Wq), 14, "prior"))
    m = net.BinaryMLPNetwork([64, 64, 128, 256], mlp_size=128, use_prior=use_prior)
    syn.fill_state_dict(m, seed=15, gain=1.2)
    pri = torch.cat([prior * 2 - 1] + [-torch.ones_like(prior)] * (P - 1), 1) if use_prior else None
    return m, feat, rd, pri


@pytest.mark.parametrize("use_prior", [False, True])
def test_fused_logits_golden(use_prior):
    from implicit_depth_amd.mlp import occlusion_logits

    m, feat, rd, pri = _setup(use_prior)
    g = load_golden(f"g4_binarymlp_prior{int(use_prior)}")
    m.cuda()
    f_nhwc = feat.permute(0, 2, 3, 1).contiguous().cuda()
    out = occlusion_logits(m, f_nhwc, 0, 64, rd.cuda(), pri.cuda() if pri is not None else None)
    assert rel_err(out.cpu(), g["logits"]) < TOL


@pytest.mark.parametrize("use_prior", [False, True])
def test_module_interface_matches_oracle(use_prior):
    """BinaryMLPNetwork.forward([BHWC], max_scale_only) — the reference's call (bd_model.py:439)."""
    m, feat, rd, pri = _setup(use_prior)
    w = {k: v.clone() for k, v in m.state_dict().items()}
    parts = [rd[:, 0:1], feat] + ([pri[:, 0:1]] if use_prior else [])
    x = torch.cat(parts, 1).permute(0, 2, 3, 1).contiguous()
    ref = onet.binary_mlp(x.double(), {k: v.double() for k, v in w.items()})
    out = m.cuda()([x.cuda()], max_scale_only=True)
    assert list(out) == ["pred_0"] and out["pred_0"].shape == ref.shape
    assert rel_err(out["pred_0"].cpu(), ref) < TOL


@pytest.mark.parametrize("use_prior", [False, True])
def test_module_interface_reads_the_references_permuted_view_in_place(use_prior):
    """BDModel.run_mlp_val (bd_model.py:415-439) concatenates [depth | feature_s0 | prior] along dim 1 (NCHW) and passes
    ``model_inputs.permute(0, 2, 3, 1)`` - a strided VIEW.  The drop-in reads its channel planes in place (idh_binary_mlp_strided_fwd): same
    bits as the contiguous (B,H,W,Cin) rows, no copy of the tensor."""
    m, feat, rd, pri = _setup(use_prior)
    m.cuda()
    parts = [rd[:, 0:1], feat] + ([pri[:, 0:1]] if use_prior else [])
    x_nchw = torch.cat(parts, 1).cuda()
    view = x_nchw.permute(0, 2, 3, 1)
    assert not view.is_contiguous()
    rows = view.contiguous()
    a = m([view], max_scale_only=True)["pred_0"]
    b = m([rows], max_scale_only=True)["pred_0"]
    assert a.shape == b.shape == (*view.shape[:3], 1) and torch.equal(a, b)
    ref = onet.binary_mlp(rows.cpu().double(), {k: v.detach().cpu().double() for k, v in m.state_dict().items()})
    assert rel_err(a.cpu(), ref) < TOL
    # two frames, odd size, a batch-strided view (every second frame of a larger tensor)
    big = torch.randn(4, x_nchw.shape[1], 7, 9, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    v2 = big[::2].permute(0, 2, 3, 1)
    assert torch.equal(m([v2], max_scale_only=True)["pred_0"], m([v2.contiguous()], max_scale_only=True)["pred_0"])


def test_prior_absent_is_minus_one_and_odd_sizes():
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.mlp import occlusion_logits

    B, H, W, P = 2, 13, 19, 5  # M = 494 rows: not a multiple of the 32-pixel wave tile
    feat = syn.randn((B, 64, H, W), 31, "f")
    rd = syn.rendered_depth_planes(B, H, W, P)
    m = net.BinaryMLPNetwork([64, 64, 128, 256], use_prior=True)
    syn.fill_state_dict(m, seed=32, gain=1.2)
    w = {k: v.double() for k, v in m.state_dict().items()}
    ref = onet.occlusion_logits(feat.double(), rd.double(), w, -torch.ones(B, P, H, W, dtype=torch.float64))
    # features live in a channel slice of a wider NHWC buffer (as in the fused pipeline)
    buf = torch.zeros(B, H, W, 80)
    buf[..., 8:72] = feat.permute(0, 2, 3, 1)
    out = occlusion_logits(m.cuda(), buf.cuda(), 8, 64, rd.cuda(), None)
    assert rel_err(out.cpu(), ref) < TOL


def test_all_scales_interface():
    from implicit_depth_amd import networks as net

    m = net.BinaryMLPNetwork([64, 64, 128, 256], use_prior=False)
    syn.fill_state_dict(m, seed=40, gain=1.2)
    xs = [syn.randn((1, 6, 5, c + 1), 41 + i, "x") for i, c in enumerate([64, 64, 128, 256])]
    out = m.cuda()([x.cuda() for x in xs], max_scale_only=False)
    for s, x in enumerate(xs):
        seq = m.mlps[f"s{s}"]
        h = onet.elu(x.double() @ seq[0].weight.cpu().double().t() + seq[0].bias.cpu().double())
        h = onet.elu(h @ seq[2].weight.cpu().double().t() + seq[2].bias.cpu().double())
        ref = h @ seq[4].weight.cpu().double().t() + seq[4].bias.cpu().double()
        assert rel_err(out[f"pred_{s}"].cpu(), ref) < TOL


def test_sample_prior_golden_and_oracle():
    from implicit_depth_amd.mlp import sample_prior

    g = load_golden("g4_sample_prior")
    Hq, Wq = [int(v) for v in g["dims"]]
    rd = syn.rendered_depth_planes(1, Hq, Wq, 3)
    rd[:, 1, :5, :7] = 0.0
    prior = torch.sigmoid(syn.randn((1, 1, Hq, Wq), 14, "prior"))
    Ks0 = syn.intrinsics(Wq, Hq).float()[None]
    cur_pose, prev_pose = syn.source_pose(0).float()[None], syn.source_pose(1).float()[None]
    args = (cur_pose, torch.linalg.inv(prev_pose), Ks0, torch.linalg.inv(Ks0))
    sp = sample_prior(rd[:, 1:2].cuda(), prior.cuda(), *[a.cuda() for a in args]).cpu()
    assert ((sp - torch.as_tensor(g["sampled"])).abs() > 1e-6).float().mean().item() < 2e-3
    # all planes at once == plane by plane through the oracle
    allp = sample_prior(rd.cuda(), prior.cuda(), *[a.cuda() for a in args]).cpu()
    for p in range(3):
        ref = onet.sample_prior(rd[:, p : p + 1], prior, *args)
        assert ((allp[:, p : p + 1] - ref).abs() > 1e-6).float().mean().item() < 2e-3
    assert (allp[:, 1, :5, :7] == -1).all()


def test_fused_binary_depth_search_matches_reference_loop():
    """infer_depth (bd_model.py:273-292): 12 dependent MLP passes, restated with the oracle MLP."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.mlp import infer_depth

    B, H, W = 1, 20, 28
    feat = syn.randn((B, 64, H, W), 51, "f")
    m = net.BinaryMLPNetwork([64, 64, 128, 256], use_prior=False)
    syn.fill_state_dict(m, seed=52, gain=1.2)
    with torch.no_grad():
        m.mlps["s0"][0].weight[:, 0] *= 3.0  # make the logit depend visibly on the query depth
    w = {k: v.double() for k, v in m.state_dict().items()}
    f64 = feat.double()
    lo = torch.full((B, 1, H, W), 0.5, dtype=torch.float64)
    hi = torch.full((B, 1, H, W), 8.0, dtype=torch.float64)
    sd = torch.full((B, 1, H, W), 7.5 / 2.0, dtype=torch.float64)
    margins = []
    for _ in range(12):
        logit = onet.occlusion_logits(f64, sd, w)
        margins.append(logit.abs())
        vis = torch.sigmoid(logit) < 0.5
        hi = torch.where(vis, sd, hi)
        lo = torch.where(~vis, sd, lo)
        sd = (hi + lo) / 2
    got_sd, got_logit = infer_depth(m.cuda(), feat.permute(0, 2, 3, 1).contiguous().cuda(), 0, 64)
    # a pixel whose logit came within fp32 noise of 0 at some step may legitimately branch the
    # other way; everywhere else the 12 decisions — hence the final depth — must be identical
    agree = (got_sd.cpu().double() - sd).abs() < 1e-6
    assert agree.float().mean().item() > 0.99
    knife_edge = torch.stack(margins).min(0).values < 1e-4
    assert bool((agree | knife_edge).all())
    assert rel_err(got_logit.cpu()[agree], logit[agree]) < TOL
    assert got_sd.min().item() >= 0.5 and got_sd.max().item() <= 8.0


def test_fused_search_with_per_depth_thresholder():
    """bd_model.py:282-283 with a Thresholder (binary_metrics_utils.py:42-52): the threshold of each step
    is looked up from the step's query depth."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.metrics import Thresholder
    from implicit_depth_amd.mlp import infer_depth

    B, H, W = 2, 12, 20
    feat = syn.randn((B, 64, H, W), 61, "f")
    m = net.BinaryMLPNetwork([64, 64, 128, 256], use_prior=False)
    syn.fill_state_dict(m, seed=62, gain=1.2)
    with torch.no_grad():
        m.mlps["s0"][0].weight[:, 0] *= 3.0
    planes = torch.tensor([1.5 + 0.5 * i for i in range(8)])
    th = Thresholder(planes, torch.tensor([0.3, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7]))
    w = {k: v.double() for k, v in m.state_dict().items()}
    f64 = feat.double()
    lo = torch.full((B, 1, H, W), 0.5, dtype=torch.float64)
    hi = torch.full((B, 1, H, W), 8.0, dtype=torch.float64)
    sd = torch.full((B, 1, H, W), 7.5 / 2.0, dtype=torch.float64)
    margins = []
    for _ in range(12):
        logit = onet.occlusion_logits(f64, sd, w)
        thr = th.thresholds.double()[torch.bucketize(sd, th.bins.double())]   # the reference's get_thresholds
        margins.append((torch.sigmoid(logit) - thr).abs())
        vis = torch.sigmoid(logit) < thr
        hi = torch.where(vis, sd, hi)
        lo = torch.where(~vis, sd, lo)
        sd = (hi + lo) / 2
    got_sd, got_logit = infer_depth(m.cuda(), feat.permute(0, 2, 3, 1).contiguous().cuda(), 0, 64, thresholder=th)
    agree = (got_sd.cpu().double() - sd).abs() < 1e-6
    assert agree.float().mean().item() > 0.99
    knife_edge = torch.stack(margins).min(0).values < 1e-4
    assert bool((agree | knife_edge).all())
    # and it differs from the constant-0.5 search (the table is doing something)
    const_sd, _ = infer_depth(m, feat.permute(0, 2, 3, 1).contiguous().cuda(), 0, 64)
    assert (const_sd.cpu().double() - sd).abs().max().item() > 1e-3
