"""The network-level C entry points (include/idh_net.h: idh_basic_block_fwd, idh_cvencoder_fwd, idh_unetpp_fwd) called through ctypes the way a
non-Python host would — size query, pack into a caller-owned blob, forward into caller-owned NHWC / NCHW tensors with a caller-owned workspace —
against (a) the reference's goldens (modules/layers.py:78-95, modules/networks.py:186-215, :20-84, :118-183) and (b) the Python drop-ins, whose
plans the C++ builder mirrors op for op: bit-identical."""
import ctypes as C

import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _bind():
    from implicit_depth_amd import _lib
    from implicit_depth_amd import net_abi as na

    return _lib, _lib.lib(), na


def _alloc(sz):
    # (+64: room to hand the library a 256-byte-aligned base whatever torch returns)
    blob = torch.empty(sz.weight_floats + 64, device="cuda")
    ws = torch.full((max(sz.workspace_floats, 1) + 64,), float("nan"), device="cuda")  # NaN: a read of anything the pass did not write shows up
    assert blob.data_ptr() % 256 == 0 and ws.data_ptr() % 256 == 0
    return blob, ws


@pytest.mark.parametrize("tag", ["id", "proj", "down"])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_basic_block_entry_point_golden(tag, layout):
    from implicit_depth_amd.layers import BasicBlock

    _lib, L, na = _bind()
    g = load_golden(f"g3_basicblock_{tag}")
    cin, cout, stride = [int(v) for v in g["dims"]]
    bb = BasicBlock(cin, cout, stride).cuda()
    syn.fill_state_dict(bb, seed=10)
    x = syn.randn((2, 24, 12, 20), 7, "bb_x").cuda()
    N, _, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    keep = []
    blk = na.block_params(bb, keep)
    if layout == "nchw":
        out = torch.empty(N, cout, Ho, Wo, device="cuda")
        tx, to = na.nchw(x), na.nchw(out)
    else:  # 24 channels: a whole zero-padded 32-channel buffer; the output as a channel slice of a wider buffer
        xb = torch.zeros(N, H, W, 32, device="cuda")
        xb[..., :24] = x.permute(0, 2, 3, 1)
        wide = torch.full((N, Ho, Wo, cout + 16), 7.0, device="cuda")
        tx = na.nhwc(xb, 24)
        to = na.Tensor(wide.data_ptr() + 4 * 16, na.LAYOUT_NHWC, cout, Ho, Wo, cout + 16)
    sz = na.NetSizes()
    _lib.check(L.idh_basic_block_sizes(C.byref(blk), N, C.byref(tx), C.byref(to), C.byref(sz)), "sizes")
    blob, ws = _alloc(sz)
    _lib.check(L.idh_basic_block_pack(C.byref(blk), N, C.byref(tx), C.byref(to), blob.data_ptr(), _lib.stream_ptr()), "pack")
    for _ in range(2):
        _lib.check(L.idh_basic_block_fwd(C.byref(blk), blob.data_ptr(), N, C.byref(tx), C.byref(to), ws.data_ptr(), sz.workspace_floats, _lib.stream_ptr()), "fwd")
    y = out if layout == "nchw" else wide[..., 16:].permute(0, 3, 1, 2)
    assert rel_err(y.cpu(), g["y"]) < TOL
    assert torch.equal(y, bb(x)), "same op list as the Python drop-in"
    if layout == "nhwc":
        assert bool((wide[..., :16] == 7.0).all()), "the channels beside the output slice are untouched"
    # a workspace one float short is refused before anything is launched
    if sz.workspace_floats:
        assert L.idh_basic_block_fwd(C.byref(blk), blob.data_ptr(), N, C.byref(tx), C.byref(to), ws.data_ptr(), sz.workspace_floats - 1, _lib.stream_ptr()) == -4


def _small_nets():
    from implicit_depth_amd import networks as net

    Hm, Wm, Dcv = 24, 32, 16
    pyr = syn.encoder_pyramid(1, Hm * 4, Wm * 4, seed=11)
    cvol = syn.randn((1, Dcv, Hm, Wm), 11, "cv_in")
    cve = net.CVEncoder(num_ch_cv=Dcv, num_ch_enc=[48, 64, 160, 256], num_ch_outs=[64, 128, 256, 384])
    syn.fill_state_dict(cve, seed=12)
    return net, pyr, cvol, cve


def _run_cvencoder(cve, cvol, img, out_layout="nhwc"):
    _lib, L, na = _bind()
    N, D, H, W = cvol.shape
    keep = []
    blocks = na.cvencoder_blocks(cve, keep)
    cost_nhwc = cvol.permute(0, 2, 3, 1).contiguous()
    cost_nchw = cvol.contiguous()
    cost = na.nhwc(cost_nhwc) if out_layout == "nhwc" else na.nchw(cost_nchw)  # (the all-NCHW call imports the volume through the workspace)
    imgs = na.tensors([na.nchw(t) for t in img])
    chans = cve.num_ch_enc
    if out_layout == "nhwc":
        outs_t = [torch.empty(N, H >> i, W >> i, c, device="cuda") for i, c in enumerate(chans)]
        outs = na.tensors([na.nhwc(t) for t in outs_t])
    else:
        outs_t = [torch.empty(N, c, H >> i, W >> i, device="cuda") for i, c in enumerate(chans)]
        outs = na.tensors([na.nchw(t) for t in outs_t])
    sz = na.NetSizes()
    _lib.check(L.idh_cvencoder_sizes(blocks, 4, N, C.byref(cost), imgs, outs, C.byref(sz)), "sizes")
    blob, ws = _alloc(sz)
    _lib.check(L.idh_cvencoder_pack(blocks, 4, N, C.byref(cost), imgs, outs, blob.data_ptr(), _lib.stream_ptr()), "pack")
    for _ in range(2):
        _lib.check(L.idh_cvencoder_fwd(blocks, 4, blob.data_ptr(), N, C.byref(cost), imgs, outs, ws.data_ptr(), sz.workspace_floats, _lib.stream_ptr()), "fwd")
    res = [t.permute(0, 3, 1, 2) if out_layout == "nhwc" else t for t in outs_t]
    return res, sz.as_dict()


@pytest.mark.parametrize("out_layout", ["nhwc", "nchw"])
def test_cvencoder_entry_point_golden(out_layout):
    net, pyr, cvol, cve = _small_nets()
    g = load_golden("g3_cvencoder")
    cve.cuda()
    img = [p.cuda() for p in pyr[1:]]
    outs, sz = _run_cvencoder(cve, cvol.cuda(), img, out_layout)
    ref = cve(cvol.cuda(), img)
    for i, o in enumerate(outs):
        assert rel_err(o.cpu(), g[f"o{i}"]) < TOL
        assert torch.equal(o, ref[i]), f"level {i}: same op list as the Python drop-in"
    print("cvencoder entry point:", sz)


def _run_unetpp(dec, feats_nchw, level0_nchw=True):
    """feats: 5 NCHW tensors.  Level 0 (24 channels) goes in as NCHW (imported by the library) or as a zero-padded NHWC buffer; the others NHWC."""
    _lib, L, na = _bind()
    N = feats_nchw[0].shape[0]
    keep = []
    blocks, heads = na.unetpp_blocks(dec, keep)
    nh = [t.permute(0, 2, 3, 1).contiguous() for t in feats_nchw]
    if level0_nchw:
        f0 = na.nchw(feats_nchw[0].contiguous())
    else:
        c0 = feats_nchw[0].shape[1]
        pad = torch.zeros(N, nh[0].shape[1], nh[0].shape[2], (c0 + 15) // 16 * 16, device="cuda")
        pad[..., :c0] = nh[0]
        keep.append(pad)
        f0 = na.nhwc(pad, c0)
    feats = na.tensors([f0] + [na.nhwc(t) for t in nh[1:]])
    H0, W0 = feats_nchw[0].shape[2:]
    chans = [64, 64, 128, 256]
    fo_t = [torch.empty(N, H0 >> i, W0 >> i, c, device="cuda") for i, c in enumerate(chans)]
    fouts = na.tensors([na.nhwc(t) for t in fo_t])
    ld = dp = None
    if heads is not None:
        ld = [torch.empty(N, 1, H0 >> i, W0 >> i, device="cuda") for i in range(4)]
        dp = [torch.empty(N, 1, H0 >> i, W0 >> i, device="cuda") for i in range(4)]
    sz = na.NetSizes()
    _lib.check(L.idh_unetpp_sizes(blocks, na.UNETPP_BLOCKS, heads, N, feats, fouts, C.byref(sz)), "sizes")
    blob, ws = _alloc(sz)
    _lib.check(L.idh_unetpp_pack(blocks, na.UNETPP_BLOCKS, heads, N, feats, fouts, blob.data_ptr(), _lib.stream_ptr()), "pack")
    for _ in range(2):
        _lib.check(L.idh_unetpp_fwd(blocks, na.UNETPP_BLOCKS, heads, blob.data_ptr(), N, feats, fouts, na.ptr_array(ld), na.ptr_array(dp), ws.data_ptr(),
                                    sz.workspace_floats, _lib.stream_ptr()), "fwd")
    return [t.permute(0, 3, 1, 2) for t in fo_t], ld, dp, sz.as_dict()


@pytest.mark.parametrize("which", ["bd", "depth"])
@pytest.mark.parametrize("level0_nchw", [True, False])
def test_unetpp_entry_point_golden(which, level0_nchw):
    net, pyr, cvol, cve = _small_nets()
    gin = load_golden("g3_cvencoder")
    dec_in = [pyr[0].cuda()] + [torch.as_tensor(gin[f"o{i}"]).cuda() for i in range(4)]
    cls, nm, key = ((net.BDDecoderPP, "g3_bddecoder", "feature_s{}_b1hw") if which == "bd" else (net.DepthDecoderPP, "g3_depthdecoder", "log_depth_pred_s{}_b1hw"))
    g = load_golden(nm)
    dec = cls([24, 64, 128, 256, 384]).cuda()
    syn.fill_state_dict(dec, seed=13)
    feats, ld, dp, sz = _run_unetpp(dec, dec_in, level0_nchw)
    ref = dec(dec_in)
    for i in range(4):
        got = feats[i] if which == "bd" else ld[i]
        assert rel_err(got.cpu(), g[f"s{i}"]) < TOL
        assert torch.equal(got, ref[key.format(i)]), f"scale {i}: same op list as the Python drop-in"
        if which == "depth":
            assert torch.equal(dp[i], torch.exp(ld[i])) or rel_err(dp[i].cpu(), torch.exp(ld[i]).cpu()) < 1e-6
    print(f"unetpp entry point ({which}):", sz)


@pytest.mark.parametrize("N", [1, 6])
def test_entry_points_at_full_size_are_bit_identical_to_the_dropins(N):
    """512x384 shapes (matching map 96x128, decoder top scale 192x256), D = 64: at N = 6 the plans hold F(4x4), F(2x2), LDS-staged and direct
    launches, split-K reductions, level-merged grids and recycled activation buffers; the C++ builder must reproduce the Python plan's
    results bit for bit (and its census of kernels)."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd import nhwc

    cve = net.CVEncoder(64, [48, 64, 160, 256], [64, 128, 256, 384]).cuda()
    dec = net.BDDecoderPP([24] + cve.num_ch_enc).cuda()
    syn.fill_state_dict(cve, seed=21)
    syn.fill_state_dict(dec, seed=22)
    pyr = [t.cuda() for t in syn.encoder_pyramid(N, 384, 512, seed=23)]
    cvol = syn.randn((N, 64, 96, 128), 24, "cv").cuda()
    enc_ref = cve(cvol, pyr[1:])
    enc_c, sz_e = _run_cvencoder(cve, cvol, pyr[1:])
    for i in range(4):
        assert torch.equal(enc_c[i], enc_ref[i]), f"CVEncoder level {i}"
    dec_ref = dec([pyr[0]] + enc_ref)
    feats, _, _, sz_d = _run_unetpp(dec, [pyr[0]] + [t.contiguous() for t in enc_ref])
    for i in range(4):
        assert torch.equal(feats[i], dec_ref[f"feature_s{i}_b1hw"]), f"decoder scale {i}"

    def census(mod):
        p = next(iter(mod.__dict__["_idh_plans"].values()))[0]
        convs = [op for op in p.ops if op.kind == nhwc.OP_CONV]
        return {"wino4": sum(op.tile_m == nhwc.TILE_WINO4 for op in convs), "wino2": sum(op.tile_m == nhwc.TILE_WINO for op in convs), "recycled": p.recycled}

    ce, cd = census(cve), census(dec)
    print(f"N={N}: CVEncoder python {ce} / C {sz_e}; decoder python {cd} / C {sz_d}")
    for py, c in ((ce, sz_e), (cd, sz_d)):
        assert (py["wino4"], py["wino2"]) == (c["wino4"], c["wino2"])
        # (outputs written straight into the caller's NHWC tensors take no pooled buffer: the counts may differ by the number of outputs)
        assert abs(py["recycled"] - c["recycled"]) <= 4 and (py["recycled"] > 0) == (c["recycled"] > 0)
