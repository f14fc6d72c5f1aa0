"""HIP implicit-GEMM conv stack (BasicBlock / CVEncoder / UNet++ decoders) vs the reference
goldens and the oracle.  fp32 MFMA is exact fp32, so the bar is the 1e-4 scale-relative
tolerance of BASELINE.json with a lot of room."""
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err
from oracle import networks as onet

pytestmark = pytest.mark.gpu


def _cpu_sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("tag", ["id", "proj", "down"])
def test_basic_block_golden(tag):
    from implicit_depth_amd.layers import BasicBlock

    g = load_golden(f"g3_basicblock_{tag}")
    cin, cout, stride = [int(v) for v in g["dims"]]
    bb = BasicBlock(cin, cout, stride)
    syn.fill_state_dict(bb, seed=10)
    x = syn.randn((2, 24, 12, 20), 7, "bb_x")
    y = bb.cuda()(x.cuda()).cpu()
    assert rel_err(y, g["y"]) < TOL


@pytest.mark.parametrize("shape", [(1, 16, 16, 9, 7, 1), (3, 64, 64, 33, 47, 1), (2, 112, 64, 24, 32, 1), (1, 64, 128, 31, 45, 2),
                                   (2, 416, 256, 6, 8, 1), (1, 256, 384, 12, 16, 2), (4, 640, 384, 3, 4, 1), (1, 24, 64, 40, 52, 1),
                                   (2, 64, 128, 32, 64, 2), (1, 128, 256, 24, 40, 2)])
def test_basic_block_vs_oracle(shape):
    """odd sizes, channel counts of every CVEncoder/decoder layer family, split-K (tiny maps), stride-2 blocks whose
    conv2 + strided 1x1 projection runs in the LDS-staged kernel (output width >= 16) or the direct kernel."""
    from implicit_depth_amd.layers import BasicBlock

    N, cin, cout, H, W, stride = shape
    bb = BasicBlock(cin, cout, stride)
    syn.fill_state_dict(bb, seed=cin + cout)
    x = syn.randn((N, cin, H, W), 3, "x")
    ref = onet.basic_block(x.double(), {k: v.double() for k, v in _cpu_sd(bb).items()}, stride)
    y = bb.cuda()(x.cuda()).cpu()
    assert y.shape == ref.shape
    assert rel_err(y, ref) < TOL
    # second call replays the cached plan on fresh inputs
    x2 = syn.randn((N, cin, H, W), 4, "x2")
    ref2 = onet.basic_block(x2.double(), {k: v.double() for k, v in _cpu_sd(bb).items()}, stride)
    assert rel_err(bb(x2.cuda()).cpu(), ref2) < TOL


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 64), (1, 128, 256, 31, 45), (3, 48, 96, 24, 40), (1, 256, 384, 24, 32), (2, 16, 64, 9, 33)])
def test_stride2_first_conv_on_the_lds_loader_vs_oracle(shape):
    """nhwc.S2_FIRST: conv1 of a stride-2 BasicBlock (layers.py:62-66) through the LDS-staged kernel's stride-2 loader (an empty first
    source + the strided 3x3 as the second one) instead of conv_mfma_k - forced here at every size (threshold 0), incl. odd maps, 32-channel
    tiles (Cout = 96), a 16-channel input and split K; against the fp64 oracle, and the plan must really hold an 8- / 4-row stride-2 op."""
    from implicit_depth_amd import nhwc
    from implicit_depth_amd.layers import BasicBlock

    N, cin, cout, H, W = shape
    bb = BasicBlock(cin, cout, 2)
    syn.fill_state_dict(bb, seed=cin + cout + 1)
    x = syn.randn((N, cin, H, W), 5, "x")
    ref = onet.basic_block(x.double(), {k: v.double() for k, v in _cpu_sd(bb).items()}, 2)
    old, nhwc.S2_FIRST_MIN_BLOCKS = nhwc.S2_FIRST_MIN_BLOCKS, 0
    try:
        y = bb.cuda()(x.cuda()).cpu()
        plan = next(iter(bb.__dict__["_idh_plans"].values()))[0]
        hit = [op for op in plan.ops if op.kind == nhwc.OP_CONV and op.src[0].stride == 2 and not op.src[1].in_ and op.tile_m in (8, 9)]
        assert len(hit) == 1
    finally:
        nhwc.S2_FIRST_MIN_BLOCKS = old
        bb.__dict__.pop("_idh_plans", None)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < TOL


def test_weight_update_invalidates_packed_cache():
    from implicit_depth_amd.layers import BasicBlock

    bb = BasicBlock(16, 16).cuda()
    syn.fill_state_dict(bb, seed=1)
    x = syn.randn((1, 16, 8, 8), 1, "x").cuda()
    y0 = bb(x).clone()
    with torch.no_grad():
        bb.conv2.weight.mul_(0.5)
        bb.conv2.bias.zero_()
    y1 = bb(x)
    ref = onet.basic_block(x.cpu().double(), {k: v.double() for k, v in _cpu_sd(bb).items()})
    assert rel_err(y1.cpu(), ref) < TOL and not torch.equal(y0, y1)


def _nets():
    from implicit_depth_amd import networks as net

    Hm, Wm, Dcv = 24, 32, 16
    pyr = syn.encoder_pyramid(1, Hm * 4, Wm * 4, seed=11)
    cvol = syn.randn((1, Dcv, Hm, Wm), 11, "cv_in")
    cve = net.CVEncoder(num_ch_cv=Dcv, num_ch_enc=[48, 64, 160, 256], num_ch_outs=[64, 128, 256, 384])
    syn.fill_state_dict(cve, seed=12)
    return net, pyr, cvol, cve


def test_cvencoder_golden():
    net, pyr, cvol, cve = _nets()
    g = load_golden("g3_cvencoder")
    outs = cve.cuda()(cvol.cuda(), [p.cuda() for p in pyr[1:]])
    for i, o in enumerate(outs):
        assert rel_err(o.cpu(), g[f"o{i}"]) < TOL


@pytest.mark.parametrize("which", ["bd", "depth"])
def test_decoders_golden(which):
    net, pyr, cvol, cve = _nets()
    gin = load_golden("g3_cvencoder")
    dec_in = [pyr[0]] + [torch.as_tensor(gin[f"o{i}"]) for i in range(4)]
    cls, nm, key = ((net.BDDecoderPP, "g3_bddecoder", "feature_s{}_b1hw") if which == "bd"
                    else (net.DepthDecoderPP, "g3_depthdecoder", "log_depth_pred_s{}_b1hw"))
    g = load_golden(nm)
    dec = cls([24, 64, 128, 256, 384])
    syn.fill_state_dict(dec, seed=13)
    out = dec.cuda()([t.cuda() for t in dec_in])
    assert sorted(out) == sorted(key.format(i) for i in range(4))
    for i in range(4):
        assert rel_err(out[key.format(i)].cpu(), g[f"s{i}"]) < TOL


def test_cvencoder_decoder_batched_vs_oracle():
    """B=2, D=64 (ds_conv_0 identity residual) at a non-golden size."""
    from implicit_depth_amd import networks as net

    B, Hm, Wm, D = 2, 16, 24, 64
    pyr = syn.encoder_pyramid(B, Hm * 4, Wm * 4, seed=21)
    cvol = syn.randn((B, D, Hm, Wm), 21, "cv")
    cve = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    syn.fill_state_dict(cve, seed=22)
    syn.fill_state_dict(dec, seed=23)
    ref_e = onet.cv_encoder(cvol, list(pyr[1:]), _cpu_sd(cve))
    ref_d = onet.unetpp_decoder([pyr[0]] + ref_e, _cpu_sd(dec), depth_head=False)
    cve.cuda(), dec.cuda()
    outs = cve(cvol.cuda(), [p.cuda() for p in pyr[1:]])
    for o, r in zip(outs, ref_e):
        assert rel_err(o.cpu(), r) < TOL
    out = dec([pyr[0].cuda()] + outs)
    for i in range(4):
        assert rel_err(out[f"feature_s{i}_b1hw"].cpu(), ref_d[f"feature_s{i}_b1hw"]) < TOL


def test_matching_head_golden_and_layouts():
    """1x1 conv + InstanceNorm + LeakyReLU + replicate-padded 3x3 conv + InstanceNorm
    (reference networks.py:279-283); NCHW and channels-last hand-over."""
    from implicit_depth_amd import networks as net

    g = load_golden("g7_matching_head")
    stem = syn.StubResnetStem()
    enc = net.ResnetMatchingEncoder([stem.conv1, stem.bn1, stem.relu, stem.maxpool, stem.layer1], 16)
    syn.fill_state_dict(enc, seed=40)
    enc.cuda()
    x = syn.randn((3, 64, 24, 32), 41, "mh_x").cuda()
    from implicit_depth_amd.nhwc import matching_head_forward

    y = matching_head_forward(enc, x)
    assert rel_err(y.cpu(), g["y"]) < TOL
    y_cl = matching_head_forward(enc, x, channels_last=True)
    assert torch.equal(y_cl.permute(0, 3, 1, 2).contiguous(), y)
    # whole module incl. the (stand-in) backbone run by torch
    img = syn.randn((2, 3, 48, 64), 42, "img").cuda()
    ref = onet.matching_head(enc.backbone(img).cpu().double(), {k: v.cpu().double() for k, v in enc.state_dict().items()})
    assert rel_err(enc(img).cpu(), ref) < TOL


def test_instance_norm_of_a_channel_with_a_large_mean():
    """nn.InstanceNorm2d on a channel whose |mean| is ~1e3 x its spread (a biased 1x1 conv on ReLU features): E[x^2] - mean^2 on
    the raw fp32 values loses ~(mean/std)^2 * 1e-7 = 10 % of the variance; the kernels accumulate around a pivot instead."""
    from implicit_depth_amd import nhwc

    N, H, W, C = 2, 40, 48, 32
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(N, H, W, C, device="cuda", generator=g)
    x[..., : C // 2] += 1000.0  # mean / std = 1e3
    x[..., C // 2 :] *= 50.0
    p = nhwc.Plan(x.device)
    out = p.buffer(N, H, W, C)
    p.instance_norm(nhwc.View(x, 0, C), out)
    p.run()
    ref = torch.nn.functional.instance_norm(x.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
    assert rel_err(out.dense().cpu(), ref.cpu()) < 1e-4


@pytest.mark.parametrize("C", [16, 64, 128, 256, 512, 1024])
@pytest.mark.parametrize("N", [1, 260])
def test_instance_norm_every_accepted_width_and_block_size(C, N):
    """IDH_OP_INSTNORM accepts every C with (C/4) | 256: the statistics kernel's combine must cover all 8 * C/4 (quad, sum) slots whatever
    the block size (round-5 advisor: C >= 256 at 256 threads left slots unwritten).  N = 1 takes the 1024-thread launch (few (image, chunk)
    pairs), N = 260 the 256-thread one (>= 512 pairs)."""
    from implicit_depth_amd import nhwc

    H, W = 33, 33  # two chunks of 1024 pixels, the second partial
    g = torch.Generator(device="cuda").manual_seed(C + N)
    x = torch.randn(N, H, W, C, device="cuda", generator=g) * torch.linspace(0.5, 3.0, C, device="cuda") + torch.linspace(-2.0, 2.0, C, device="cuda")
    p = nhwc.Plan(x.device)
    out = p.buffer(N, H, W, C)
    out.buf.fill_(float("nan"))
    p.instance_norm(nhwc.View(x, 0, C), out)
    p.run()
    ref = torch.nn.functional.instance_norm(x.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
    err = ((out.dense().double() - ref).abs().max() / ref.abs().max()).item()  # (NaN if any slot of the statistics was left unwritten)
    assert err < 1e-5, err


@pytest.mark.parametrize("C", [64, 128])
def test_instance_norm_statistics_do_not_depend_on_the_batch_size(C):
    """The reduction tree of instnorm_stats_k is fixed (1024 / (C/4) rows per chunk at either block size): frame 0 normalised alone
    (1024-thread blocks) and inside a 48-image batch (256-thread blocks) are bit-identical."""
    from implicit_depth_amd import nhwc

    H, W = 96, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(48, H, W, C, device="cuda", generator=g) * 3 + 1

    def run(n):
        p = nhwc.Plan(x.device)
        out = p.buffer(n, H, W, C)
        p.instance_norm(nhwc.View(x[:n].contiguous(), 0, C), out)
        p.run()
        return out.dense().clone()

    assert torch.equal(run(1)[0], run(48)[0])


@pytest.mark.parametrize("reg", [False, True])
def test_skip_decoder_golden(reg):
    """SkipDecoder / SkipDecoderRegression (networks_fast.py): ELU convs, nearest x2, concat, 1x1 heads."""
    from implicit_depth_amd import networks as net

    g = load_golden("g8_skipdecoder_reg" if reg else "g8_skipdecoder")
    dec = (net.SkipDecoderRegression if reg else net.SkipDecoder)([24, 64, 128, 256, 384])
    syn.fill_state_dict(dec, seed=45)
    pyr = syn.encoder_pyramid(1, 96, 128, seed=11)
    enc = [torch.as_tensor(load_golden("g3_cvencoder")[f"o{i}"]) for i in range(4)]
    out = dec.cuda()([t.cuda() for t in [pyr[0]] + enc])
    assert sorted(out) == sorted(k for k in g if k != "keys")
    for k in out:
        assert rel_err(out[k].cpu(), g[k]) < TOL, k


def test_conv_block_relu_and_use_bn_flag():
    """ConvBlock(use_elu=False) -> ReLU; use_bn=True is accepted and, as in the reference (networks_fast.py:10-28 never builds a
    normalisation layer), changes nothing."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd import nhwc

    x = syn.randn((2, 24, 20, 36), 3, "cb_x")
    for use_elu, use_bn in ((False, False), (True, True)):
        blk = net.ConvBlock(24, 32, use_elu=use_elu, use_bn=use_bn)
        assert sorted(blk.state_dict()) == ["conv1.bias", "conv1.weight", "conv2.bias", "conv2.weight"]
        syn.fill_state_dict(blk, seed=9)
        f = torch.nn.functional
        act = f.elu if use_elu else f.relu
        ref = act(f.conv2d(act(f.conv2d(x.double(), blk.conv1.weight.double(), blk.conv1.bias.double(), padding=1)), blk.conv2.weight.double(),
                           blk.conv2.bias.double(), padding=1))
        blk.cuda()
        p = nhwc.Plan(torch.device("cuda:0"))
        xin = p.buffer(2, 20, 36, 24)
        i_in = p.import_nchw(x.shape, xin)
        y = nhwc._conv_block(p, xin, blk)
        out = torch.empty(2, 32, 20, 36, device="cuda")
        p.export_nchw(y, out)
        p.set_in(i_in, x.cuda())
        p.run()
        assert rel_err(out.cpu(), ref) < TOL


@pytest.mark.parametrize("shape", [(1, 16, 1, 1), (2, 64, 5, 7), (3, 32, 12, 16), (1, 256, 3, 9), (33, 64, 48, 64)])
def test_upsample2_matches_interpolate_on_strided_slices(shape):
    """upsample2_k (one thread per 2x2 output quad, clamped 3x3 neighbourhood): F.interpolate(scale_factor=2, mode="bilinear",
    align_corners=False) (utils/generic_utils.py:94-103 of the reference's ``upsample``) on odd sizes, 1x1 maps, a source that is a channel
    slice of a wider buffer and a destination slice of a concat buffer whose other channels must stay untouched (level_k's copy of the body runs in test_level_merged_launches_*)."""
    import torch.nn.functional as F
    from implicit_depth_amd import nhwc

    N, C, H, W = shape
    dev = torch.device("cuda")
    p = nhwc.Plan(dev)
    src = p.buffer(N, H, W, C + 16)
    cat = p.buffer(N, 2 * H, 2 * W, 2 * C + 16)
    g = torch.Generator().manual_seed(N * 1000 + C + H)
    xt = torch.randn(N, H, W, C + 16, generator=g).to(dev)
    src.dense().copy_(xt)
    cat.dense().fill_(7.0)
    p.upsample2(src.slice(16, C), cat.slice(C, C))
    p.schedule()
    p.run()
    ref = F.interpolate(xt[..., 16:].permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    got = cat.dense()
    assert float((got[..., C:2 * C].double() - ref).abs().max()) < 1e-6
    assert bool((got[..., :C] == 7.0).all()) and bool((got[..., 2 * C:] == 7.0).all())


@pytest.mark.parametrize("shape", [(2, 64, 10, 14), (1, 128, 6, 18), (3, 64, 32, 48)])
def test_pointwise_up_matches_conv1x1_plus_interpolate(shape):
    """IDH_OP_POINTWISE_UP (pointwise_up_k): out = conv1x1(x) + F.interpolate(low, x2, bilinear, align_corners=False), x a channel slice of a
    wider concat buffer, sizes that leave a ragged last 64-pixel tile; against fp64 torch."""
    import torch.nn.functional as F
    from torch import nn
    from implicit_depth_amd import nhwc

    N, C, Hl, Wl = shape
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(C + Hl)
    conv = nn.Conv2d(C, C, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
        conv.bias.copy_(torch.randn(C, generator=g))
    conv.to(dev)
    p = nhwc.Plan(dev)
    cat = p.buffer(N, 2 * Hl, 2 * Wl, 3 * C)
    low = p.buffer(N, Hl, Wl, C)
    out = p.buffer(N, 2 * Hl, 2 * Wl, C)
    xt = torch.randn(N, 2 * Hl, 2 * Wl, 3 * C, generator=g).to(dev)
    lt = torch.randn(N, Hl, Wl, C, generator=g).to(dev)
    cat.dense().copy_(xt)
    low.dense().copy_(lt)
    p.pointwise_up(cat.slice(0, C), conv, low, out)
    p.schedule()
    p.run()
    ref = F.conv2d(xt[..., :C].permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double()) + \
        F.interpolate(lt.permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=False)
    assert rel_err(out.dense().permute(0, 3, 1, 2).cpu(), ref.cpu()) < 2e-6


@pytest.mark.parametrize("batch", [1, 2])
def test_decoder_low_resolution_projection_matches_the_fused_projection(batch):
    """nhwc.PROJ_LOWRES: the in_conv blocks of the UNet++ decoders with the projection of the upsampled concat slices evaluated at low resolution
    (Plan.basic_block_upcat: W cat = W_a right + up(W_b lo + W_c lo2), conv2 + residual) - forced at every size here - against the default plan
    (projection fused into conv2) and, through it, the reference goldens; the plan must really hold the pointwise-up ops."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd import nhwc

    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    syn.fill_state_dict(dec, seed=21)
    dec.cuda()
    feats = [t.cuda() for t in syn.encoder_pyramid(batch, 256, 384, seed=5, channels=(24, 64, 128, 256, 384))]
    old = nhwc.PROJ_LOWRES, nhwc.PROJ_LOWRES_MIN_TILES
    try:
        nhwc.PROJ_LOWRES = False
        dec.__dict__.pop("_idh_plans", None)
        ref = {k: v.clone() for k, v in dec(feats).items()}
        nhwc.PROJ_LOWRES, nhwc.PROJ_LOWRES_MIN_TILES = True, 0
        dec.__dict__.pop("_idh_plans", None)
        got = {k: v.clone() for k, v in dec(feats).items()}
        plan = next(iter(dec.__dict__["_idh_plans"].values()))[0]
        assert sum(1 for op in plan.ops if op.kind == nhwc.OP_POINTWISE_UP) >= 4
    finally:
        nhwc.PROJ_LOWRES, nhwc.PROJ_LOWRES_MIN_TILES = old
        dec.__dict__.pop("_idh_plans", None)
    for k in ref:
        assert rel_err(got[k].cpu(), ref[k].cpu()) < 2e-5, k


def test_fused_upsample_concat_is_bit_identical_to_materialised():
    """nhwc.FUSE_UPSAMPLE: the decoder's x2 bilinear upsampling + concat interpolated inside the consumer conv's halo
    loader (idh_conv_src.up_*) instead of upsample2_k writing a concat buffer — same blend expression, so the outputs
    must be bit-identical (B=2: 8-row tiles at the top level, 4-row + grouped launches below)."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd import nhwc

    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    syn.fill_state_dict(dec, seed=21)
    dec.cuda()
    pyr = syn.encoder_pyramid(2, 256, 384, seed=5, channels=(24, 64, 128, 256, 384))
    feats = [t.cuda() for t in pyr]
    old = nhwc.FUSE_UPSAMPLE, nhwc.FUSED_UP_ROWS
    old_wino, nhwc.WINOGRAD = nhwc.WINOGRAD, False  # bit-identity holds kernel by kernel: a materialised concat would take a Winograd kernel
    old_wino4, nhwc.WINOGRAD4 = nhwc.WINOGRAD4, False
    try:
        outs = {}
        for fuse, rows in ((False, 8), (True, 8), (True, 4)):
            nhwc.FUSE_UPSAMPLE, nhwc.FUSED_UP_ROWS = fuse, rows
            dec.__dict__.pop("_idh_plans", None)
            outs[(fuse, rows)] = {k: v.clone() for k, v in dec(feats).items()}
            plan = next(iter(dec.__dict__["_idh_plans"].values()))[0]
            n_up = sum(1 for op in plan.ops if op.kind == nhwc.OP_UPSAMPLE2)
            fused_srcs = sum(1 for op in plan.ops if op.kind == nhwc.OP_CONV and (op.src[0].up_in[0] or op.src[1].up_in[0]))
            assert (n_up == 0 and fused_srcs > 0) if fuse else (n_up > 0 and fused_srcs == 0)
    finally:
        nhwc.FUSE_UPSAMPLE, nhwc.FUSED_UP_ROWS = old
        nhwc.WINOGRAD = old_wino
        nhwc.WINOGRAD4 = old_wino4
        dec.__dict__.pop("_idh_plans", None)
    for key in ((True, 8), (True, 4)):
        for k, v in outs[(False, 8)].items():
            assert torch.equal(outs[key][k], v), (key, k)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 6])
def test_level_merged_launches_are_bit_identical_and_fewer(batch):
    """nhwc.MERGE_LEVELS: at one frame the independent ops of a dependency level (4-row LDS convs with 64- and 32-channel
    tiles, the stride-2 direct conv, bilinear upsampling) share one level_k grid (members of <= 512 workgroups: at 6 frames
    only the low-resolution levels).  Same tile bodies, same arithmetic: outputs bit-identical to the
    one-kernel-per-kind schedule, with fewer launches (idh_count_launches)."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd import nhwc

    cve = net.CVEncoder(64, [48, 64, 160, 256], [64, 128, 256, 384])
    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    syn.fill_state_dict(cve, seed=31)
    syn.fill_state_dict(dec, seed=32)
    cve.cuda(), dec.cuda()
    pyr = [t.cuda() for t in syn.encoder_pyramid(batch, 384, 512, seed=6)]
    vol = torch.randn(batch, 64, 96, 128, generator=torch.Generator().manual_seed(3)).cuda()
    old = nhwc.MERGE_LEVELS
    outs, launches = {}, {}
    try:
        for merge in (False, True):
            nhwc.MERGE_LEVELS = merge
            for m in (cve, dec):
                m.__dict__.pop("_idh_plans", None)
            enc = cve(vol, pyr[1:])
            o = dec([pyr[0]] + list(enc))
            outs[merge] = [e.clone() for e in enc] + [o[k].clone() for k in sorted(o)]
            launches[merge] = sum(next(iter(m.__dict__["_idh_plans"].values()))[0].count_launches() for m in (cve, dec))
    finally:
        nhwc.MERGE_LEVELS = old
        for m in (cve, dec):
            m.__dict__.pop("_idh_plans", None)
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    # one frame: a launch per level instead of per op; 6 frames: only the low-resolution levels still have small members
    assert launches[True] < launches[False] - 10 if batch == 1 else launches[True] <= launches[False], launches


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 24, 32), (2, 96, 128), (9, 17, 50)])
def test_head_fusions_are_bit_identical_to_the_unfused_plan(shape):
    """The matching head's two fusions — the first 1x1 conv reading the backbone's NCHW map in place
    (IDH_OP_POINTWISE_NCHW) and InstanceNorm + LeakyReLU applied by the 3x3 conv while it stages its halo
    (idh_conv_src.norm) — against the plan that imports the layout and materialises the normalised tensor: the
    normalisation is the same expression (bit-identical); the 1x1 conv sums its 64 products in the same MFMA order."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd import nhwc

    N, H, W = shape
    enc = net.ResnetMatchingEncoder([torch.nn.Identity() for _ in range(5)], 16)
    syn.fill_state_dict(enc, seed=44)
    enc.cuda()
    x = syn.randn((N, 64, H, W), 45, "mh_fuse").cuda()
    ref = onet.matching_head(x.cpu().double(), {k: v.cpu().double() for k, v in enc.state_dict().items()})
    old = nhwc.FUSE_HEAD_NORM, nhwc.FUSE_HEAD_IMPORT
    outs = {}
    try:
        for fuse in (False, True):
            nhwc.FUSE_HEAD_NORM = nhwc.FUSE_HEAD_IMPORT = fuse
            enc.__dict__.pop("_idh_plans", None)
            outs[fuse] = nhwc.matching_head_forward(enc, x).clone()
            kinds = [op.kind for op in next(iter(enc.__dict__["_idh_plans"].values()))[0].ops]
            assert (nhwc.OP_POINTWISE_NCHW in kinds) == fuse and (nhwc.OP_IMPORT in kinds) == (not fuse)
    finally:
        nhwc.FUSE_HEAD_NORM, nhwc.FUSE_HEAD_IMPORT = old
        enc.__dict__.pop("_idh_plans", None)
    assert rel_err(outs[True].cpu(), ref) < TOL
    assert torch.equal(outs[True], outs[False])


@pytest.mark.gpu
def test_normalise_on_load_with_zero_padding():
    """idh_conv_src.norm on a zero-padded conv: the padding is applied AFTER the normalisation (a padded tap is 0, not
    act((0 - mean) * rstd)) — the reference's InstanceNorm2d -> LeakyReLU -> Conv2d(padding=1)."""
    from implicit_depth_amd import nhwc

    N, C, H, W, Co = 2, 32, 20, 33, 16
    x = syn.randn((N, C, H, W), 46, "nz_x") * 3.0 + 1.5
    conv = torch.nn.Conv2d(C, Co, 3, padding=1)
    syn.fill_state_dict(conv, seed=47)
    ref = torch.nn.functional.conv2d(torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x.double()), 0.2),
                                     conv.weight.detach().double(), conv.bias.detach().double(), padding=1)
    conv.cuda()
    outs = []
    for fuse in (True, False):
        p = nhwc.Plan(torch.device("cuda"))
        xin = p.buffer(N, H, W, C)
        i_in = p.import_nchw(x.shape, xin)
        y = p.buffer(N, H, W, Co)
        if fuse:
            assert nhwc.norm_on_load_eligible(p, xin, conv, nhwc.PAD_ZEROS)
            p.conv(xin, conv, y, norm=(p.instance_norm(xin, None), nhwc.ACT_LRELU, 0.2))
        else:
            xn = p.buffer(N, H, W, C)
            p.instance_norm(xin, xn, act=nhwc.ACT_LRELU, slope=0.2)
            p.conv(xn, conv, y)
        i_out = p.export_nchw(y)
        p.schedule()
        out = torch.empty(N, Co, H, W, device="cuda")
        p.set_in(i_in, x.cuda())
        p.set_out(i_out, out)
        p.run()
        outs.append(out.cpu())
    assert rel_err(outs[0], ref) < TOL
    assert torch.equal(outs[0], outs[1])
