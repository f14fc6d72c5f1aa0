"""Winograd F(2x2,3x3) conv kernel (csrc/conv_wino.hip) vs an fp64 torch reference, vs the direct kernel, and through the
network goldens with the kernel forced onto every eligible layer.  Replaces the same nn.Conv2d calls as the direct kernel
(reference modules/layers.py:59-95); fp32 operands and accumulation, so the bar stays the 1e-4 scale-relative tolerance of
BASELINE.json — observed ~4e-7."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err
from oracle import networks as onet

pytestmark = pytest.mark.gpu


@pytest.fixture
def wino_everywhere():
    """Force the Winograd kernel onto every eligible layer regardless of grid size / tile fill."""
    from implicit_depth_amd import nhwc

    old = (nhwc.WINOGRAD, nhwc.WINO_MIN_TILES, nhwc.WINO_MIN_FILL)
    nhwc.WINOGRAD, nhwc.WINO_MIN_TILES, nhwc.WINO_MIN_FILL = True, 1, 0.0
    yield nhwc
    nhwc.WINOGRAD, nhwc.WINO_MIN_TILES, nhwc.WINO_MIN_FILL = old


def _run(nhwc, conv, x_nhwc, res, act, slope, wino, out_view=None):
    old = nhwc.WINOGRAD
    nhwc.WINOGRAD = wino
    try:
        p = nhwc.Plan(x_nhwc.device)
        B, H, W, cs = x_nhwc.shape
        out = p.buffer(B, H, W, conv.out_channels) if out_view is None else out_view
        p.conv(nhwc.View(x_nhwc, 0, conv.in_channels), conv, out, act=act, slope=slope, res=res)
    finally:
        nhwc.WINOGRAD = old
    assert (p.ops[0].tile_m == nhwc.TILE_WINO) == wino
    p.run()
    torch.cuda.synchronize()
    return out.dense().clone()


# (B, cin, cout, H, W, residual, act): ragged maps (partial tiles in both directions, odd sizes), channel counts that need
# zero-padded input buffers (24, 112), wide outputs (NT = 3, 8), every activation, one map narrower than a tile
@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 64, True, 1), (1, 24, 64, 37, 45, False, 1), (3, 112, 96, 9, 33, True, 2), (1, 192, 64, 64, 96, False, 0),
                                   (2, 128, 256, 24, 32, True, 1), (1, 16, 32, 8, 32, False, 1), (1, 64, 32, 5, 17, True, 1), (5, 32, 64, 16, 70, False, 1)])
def test_wino_conv_vs_fp64_and_direct(shape, wino_everywhere):
    nhwc = wino_everywhere
    B, cin, cout, H, W, use_res, act = shape
    conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    syn.fill_state_dict(conv, seed=cin + cout + H)
    g = torch.Generator(device="cuda").manual_seed(H * W)
    xb = torch.zeros(B, H, W, nhwc.ceil16(cin), device="cuda")
    xb[..., :cin] = torch.randn(B, H, W, cin, device="cuda", generator=g)
    rb = torch.randn(B, H, W, cout, device="cuda", generator=g) if use_res else None
    res = nhwc.View(rb, 0, cout) if use_res else None
    ref = F.conv2d(xb[..., :cin].permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double(), padding=1)
    if use_res:
        ref = ref + rb.permute(0, 3, 1, 2).double()
    ref = (F.leaky_relu(ref, 0.2) if act == 1 else F.elu(ref) if act == 2 else ref).permute(0, 2, 3, 1)
    yw = _run(nhwc, conv, xb, res, act, 0.2, True)
    yd = _run(nhwc, conv, xb, res, act, 0.2, False)
    assert rel_err(yw.cpu(), ref.cpu()) < 1e-5, "Winograd kernel vs fp64"
    assert rel_err(yw.cpu(), yd.cpu()) < 1e-5, "Winograd kernel vs direct kernel"


# (B, cin, cout, H, W, cin2, act): fused 1x1 projection of a second tensor (BasicBlock's conv2(h) + downsample(x)): channel
# counts with an odd number of 16-channel chunks (24, 112, 208: the last 32-channel P step is half empty), wide ones, ragged maps
@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 64, 192, 1), (1, 64, 64, 37, 45, 24, 1), (3, 32, 96, 9, 33, 112, 2), (1, 128, 128, 16, 32, 384, 1),
                                   (2, 64, 32, 8, 70, 208, 0), (1, 16, 64, 24, 32, 16, 1)])
def test_wino_conv_with_fused_projection(shape, wino_everywhere):
    nhwc = wino_everywhere
    B, cin, cout, H, W, cin2, act = shape
    conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    proj = nn.Conv2d(cin2, cout, 1).cuda()
    syn.fill_state_dict(conv, seed=cin + cout + H)
    syn.fill_state_dict(proj, seed=cin2 + H)
    g = torch.Generator(device="cuda").manual_seed(H * W + cin2)
    xb = torch.randn(B, H, W, cin, device="cuda", generator=g)
    x2 = torch.zeros(B, H, W, nhwc.ceil16(cin2), device="cuda")
    x2[..., :cin2] = torch.randn(B, H, W, cin2, device="cuda", generator=g)
    ref = (F.conv2d(xb.permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double(), padding=1) +
           F.conv2d(x2[..., :cin2].permute(0, 3, 1, 2).double(), proj.weight.double(), proj.bias.double()))
    ref = (F.leaky_relu(ref, 0.2) if act == 1 else F.elu(ref) if act == 2 else ref).permute(0, 2, 3, 1)
    outs = []
    for wino in (True, False):
        old = nhwc.WINOGRAD
        nhwc.WINOGRAD = wino
        try:
            p = nhwc.Plan(xb.device)
            out = p.buffer(B, H, W, cout)
            p.conv(nhwc.View(xb, 0, cin), conv, out, act=act, slope=0.2, x2=nhwc.View(x2, 0, cin2), conv2=proj)
        finally:
            nhwc.WINOGRAD = old
        assert (p.ops[0].tile_m == nhwc.TILE_WINO) == wino
        p.run()
        p.run()  # persistent kernel state must not leak between launches
        torch.cuda.synchronize()
        outs.append(out.dense().clone())
    assert rel_err(outs[0].cpu(), ref.cpu()) < 1e-5, "Winograd kernel + fused projection vs fp64"
    assert rel_err(outs[0].cpu(), outs[1].cpu()) < 1e-5, "vs the direct kernel"


def test_wino_conv_channel_strided_views(wino_everywhere):
    """input = channel slice of a wider concat buffer, output = slice of another, residual strided too (torch.cat elimination)"""
    nhwc = wino_everywhere
    B, H, W, cin, cout = 2, 24, 64, 64, 64
    conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    syn.fill_state_dict(conv, seed=5)
    g = torch.Generator(device="cuda").manual_seed(9)
    big = torch.randn(B, H, W, 192, device="cuda", generator=g)
    rbig = torch.randn(B, H, W, 128, device="cuda", generator=g)
    obig = torch.full((B, H, W, 192), 7.0, device="cuda")
    p = nhwc.Plan(big.device)
    out = nhwc.View(obig, 64, cout)
    p.conv(nhwc.View(big, 128, cin), conv, out, act=1, slope=0.2, res=nhwc.View(rbig, 64, cout))
    assert p.ops[0].tile_m == nhwc.TILE_WINO
    p.run()
    ref = F.conv2d(big[..., 128:192].permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double(), padding=1) + rbig[..., 64:128].permute(0, 3, 1, 2).double()
    ref = F.leaky_relu(ref, 0.2).permute(0, 2, 3, 1)
    assert rel_err(obig[..., 64:128].cpu(), ref.cpu()) < 1e-5
    assert torch.all(obig[..., :64] == 7.0) and torch.all(obig[..., 128:] == 7.0), "neighbouring channel slices must stay untouched"


def test_wino_weight_update_invalidates_packed_cache(wino_everywhere):
    nhwc = wino_everywhere
    conv = nn.Conv2d(32, 32, 3, 1, 1).cuda()
    syn.fill_state_dict(conv, seed=1)
    x = torch.randn(1, 16, 32, 32, device="cuda")
    y0 = _run(nhwc, conv, x, None, 0, 0.2, True)
    with torch.no_grad():
        conv.weight.mul_(0.5)
    y1 = _run(nhwc, conv, x, None, 0, 0.2, True)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double(), padding=1).permute(0, 2, 3, 1)
    assert rel_err(y1.cpu(), ref.cpu()) < 1e-5 and not torch.equal(y0, y1)


def test_networks_with_wino_forced_match_goldens(wino_everywhere):
    """CVEncoder + BDDecoderPP goldens (reference outputs) with every eligible layer on the Winograd kernel"""
    from implicit_depth_amd import networks as net

    Hm, Wm, Dcv = 24, 32, 16
    pyr = syn.encoder_pyramid(1, Hm * 4, Wm * 4, seed=11)
    cvol = syn.randn((1, Dcv, Hm, Wm), 11, "cv_in")
    cve = net.CVEncoder(num_ch_cv=Dcv, num_ch_enc=[48, 64, 160, 256], num_ch_outs=[64, 128, 256, 384])
    syn.fill_state_dict(cve, seed=12)
    g = load_golden("g3_cvencoder")
    outs = cve.cuda()(cvol.cuda(), [p.cuda() for p in pyr[1:]])
    plan = next(iter(cve.__dict__["_idh_plans"].values()))[0]
    assert any(op.kind == 1 and op.tile_m == wino_everywhere.TILE_WINO for op in plan.ops), "no layer took the Winograd kernel"
    for i, o in enumerate(outs):
        assert rel_err(o.cpu(), g[f"o{i}"]) < TOL
    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    syn.fill_state_dict(dec, seed=13)
    gd = load_golden("g3_bddecoder")
    dec_in = [pyr[0]] + [torch.as_tensor(g[f"o{i}"]) for i in range(4)]
    out = dec.cuda()([t.cuda() for t in dec_in])
    for i in range(4):
        assert rel_err(out[f"feature_s{i}_b1hw"].cpu(), gd[f"s{i}"]) < TOL


def test_default_plan_uses_wino_at_bench_batch():
    """the default heuristics put the big plain 3x3 stride-1 layers of a batch on a Winograd kernel - F(4x4) where its 32 x 8 x 64-channel
    tiles fill the chip (>= WINO4_MIN_TILES; with or without a fused 1x1 projection), F(2x2) below that - and none at one
    small frame (fewer tiles than resident workgroups)"""
    from implicit_depth_amd import nhwc

    conv, proj = nn.Conv2d(64, 64, 3, 1, 1).cuda(), nn.Conv2d(32, 64, 1).cuda()
    for B, H, W, want, want_proj in ((16, 96, 128, nhwc.TILE_WINO4, nhwc.TILE_WINO4), (2, 96, 128, nhwc.TILE_WINO, nhwc.TILE_WINO), (1, 24, 32, None, None)):
        x, x2 = torch.randn(B, H, W, 64, device="cuda"), torch.randn(B, H, W, 32, device="cuda")
        p = nhwc.Plan(x.device)
        p.conv(nhwc.View(x, 0, 64), conv, p.buffer(B, H, W, 64))
        p.conv(nhwc.View(x, 0, 64), conv, p.buffer(B, H, W, 64), x2=nhwc.View(x2, 0, 32), conv2=proj)
        for op, w in zip(p.ops, (want, want_proj)):
            assert (op.tile_m == w) if w is not None else (op.tile_m not in (nhwc.TILE_WINO, nhwc.TILE_WINO4)), (B, H, W, op.tile_m, w)


def test_grouped_launch_of_a_level_is_bit_identical(wino_everywhere):
    """independent Winograd convs of one dependency level run as ONE persistent grid (conv3x3_wino_group_k): every op keeps
    its own tiles and arithmetic, so the results are bit-identical to one launch per op — for plain convs, for convs with a
    fused 1x1 source, and for ops with fewer tiles than workgroups (some workgroups skip an op entirely)"""
    nhwc = wino_everywhere
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(2, 64, 64, 48, 64, 0), (1, 32, 64, 24, 40, 0), (2, 96, 32, 16, 32, 0), (1, 64, 64, 8, 32, 0), (2, 64, 64, 24, 64, 48), (1, 32, 96, 16, 32, 80)]
    convs, projs, xs, x2s = [], [], [], []
    for i, (B, cin, cout, H, W, c2) in enumerate(shapes):
        conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
        syn.fill_state_dict(conv, seed=11 + i)
        convs.append(conv)
        xs.append(torch.randn(B, H, W, cin, device="cuda", generator=g))
        proj = nn.Conv2d(c2, cout, 1).cuda() if c2 else None
        if proj is not None:
            syn.fill_state_dict(proj, seed=31 + i)
        projs.append(proj)
        x2s.append(torch.randn(B, H, W, c2, device="cuda", generator=g) if c2 else None)
    outs = {}
    for group in (False, True):
        old = nhwc.WINO_GROUP
        nhwc.WINO_GROUP = group
        try:
            p = nhwc.Plan(xs[0].device)
            bufs = []
            for conv, proj, x, x2, (B, cin, cout, H, W, c2) in zip(convs, projs, xs, x2s, shapes):
                o = p.buffer(B, H, W, cout)
                p.conv(nhwc.View(x, 0, cin), conv, o, act=1, slope=0.2, x2=None if proj is None else nhwc.View(x2, 0, c2), conv2=proj)
                bufs.append(o)
            p.schedule()
            assert all(op.tile_m == nhwc.TILE_WINO for op in p.ops)
            assert p.count_launches() == (2 if group else len(shapes))  # one grid for the plain convs, one for those with a 1x1 source
            p.run()
            torch.cuda.synchronize()
            outs[group] = [b.dense().clone() for b in bufs]
        finally:
            nhwc.WINO_GROUP = old
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
