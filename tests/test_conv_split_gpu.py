"""Split-precision conv kernels (csrc/conv_split.hip, IDH_OP_CONV tile_m = 10 / 11) against fp64.

The kernels expand every fp32 operand into 16-bit pieces (3 bf16, or 2 power-of-two-scaled f16) and
accumulate the significant cross products in fp32 on the 16-bit matrix cores; the claim under test
is that the results are fp32-equivalent: the same error against fp64 as the fp32-MFMA kernel
(1e-4 of scale is the parity bar; the observed error is ~5e-7), including operands with a wide
dynamic range.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import implicit_depth_amd as idh
from implicit_depth_amd import nhwc, synthetic
from implicit_depth_amd.layers import BasicBlock
from oracle import networks as onet


MODES = ("f16x3",)


@pytest.fixture(params=MODES)
def split_math(request):
    old = nhwc.DEFAULT_MATH
    nhwc.DEFAULT_MATH = request.param
    yield request.param
    nhwc.DEFAULT_MATH = old


rows_used = set()


def _rel(a, b):
    return float((a - b).detach().abs().max() / b.detach().abs().max())


def _conv_plan(x, conv, act=nhwc.ACT_NONE, res=None, math="f16x3"):
    dev = x.device
    N, C, H, W = x.shape
    p = nhwc.Plan(dev, math=math)
    xin = p.buffer(N, H, W, C)
    i0 = p.import_nchw(x.shape, xin)
    rv = None
    if res is not None:
        rv = p.buffer(N, H, W, conv.out_channels)
        i1 = p.import_nchw(res.shape, rv)
    out = p.buffer(N, H, W, conv.out_channels)
    p.conv(xin, conv, out, act=act, res=rv)
    y = torch.empty(N, conv.out_channels, H, W, device=dev)
    ie = p.export_nchw(out, y)
    p.set_in(i0, x)
    if res is not None:
        p.set_in(i1, res)
    p.run()
    torch.cuda.synchronize()
    return y, p


@pytest.mark.parametrize("math", MODES)
@pytest.mark.parametrize("cin,cout,H,W,N", [(64, 64, 32, 48, 8), (48, 128, 16, 16, 4), (192, 64, 40, 72, 2), (20, 64, 33, 50, 3), (32, 64, 8, 16, 1),
                                            (64, 64, 64, 64, 48)])
def test_split_conv_matches_fp64(cin, cout, H, W, N, math):
    torch.manual_seed(cin + cout)
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    x = torch.randn(N, cin, H, W, device=dev)
    res = torch.randn(N, cout, H, W, device=dev)
    old = nhwc.SPLIT_MIN_BLOCKS
    nhwc.SPLIT_MIN_BLOCKS = 1
    try:
        y, p = _conv_plan(x, conv, act=nhwc.ACT_LRELU, res=res, math=math)
        assert [op.tile_m for op in p.ops if op.kind == nhwc.OP_CONV] == [nhwc.SPLIT_CODE[math]]
        rows_used.add([op.tile_n for op in p.ops if op.kind == nhwc.OP_CONV][0])
        y32, _ = _conv_plan(x, conv, act=nhwc.ACT_LRELU, res=res, math="fp32")
    finally:
        nhwc.SPLIT_MIN_BLOCKS = old
    ref = torch.nn.functional.leaky_relu(
        torch.nn.functional.conv2d(x.double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu(), padding=1) + res.double().cpu(), 0.2)
    e_split = _rel(y.double().cpu(), ref)
    e_fp32 = _rel(y32.double().cpu(), ref)
    print(f"{math} err {e_split:.3e}  fp32-mfma err {e_fp32:.3e}")
    assert e_split < 2e-6, e_split            # fp32-equivalent, far inside the 1e-4 parity bar
    assert e_split < 4 * e_fp32 + 1e-7


@pytest.mark.parametrize("math", MODES)
def test_split_conv_wide_dynamic_range(math):
    """Per-(image, channel) activation scales spanning 2^-40..2^40 and per-output-channel weight
    scales spanning 2^-20..2^20: bf16 pieces keep fp32's exponent range; the f16 mode rescales per
    halo chunk / per output channel.  Checked per (image, output channel) against its own scale."""
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(dev)
    x = torch.randn(4, 32, 32, 32, device=dev) * torch.exp2(torch.randint(-40, 40, (4, 32, 1, 1), device=dev).float())
    with torch.no_grad():
        conv.weight.mul_(torch.exp2(torch.randint(-20, 20, (64, 1, 1, 1), device=dev).float()))
    old = nhwc.SPLIT_MIN_BLOCKS
    nhwc.SPLIT_MIN_BLOCKS = 1
    try:
        y, _ = _conv_plan(x, conv, math=math)
    finally:
        nhwc.SPLIT_MIN_BLOCKS = old
    ref = torch.nn.functional.conv2d(x.double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu(), padding=1)
    err = ((y.double().cpu() - ref).abs().amax((2, 3)) / ref.abs().amax((2, 3))).max()
    assert float(err) < 2e-6, float(err)


@pytest.mark.parametrize("math", MODES)
def test_split_conv_zero_and_nonfinite_inputs(math):
    """All-zero tiles (scale exponent clamp) give exact zeros + bias; inf / nan propagate like fp32."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(dev)
    old = nhwc.SPLIT_MIN_BLOCKS
    nhwc.SPLIT_MIN_BLOCKS = 1
    try:
        x = torch.zeros(2, 32, 32, 32, device=dev)
        y, _ = _conv_plan(x, conv, math=math)
        assert torch.equal(y, conv.bias.detach().view(1, -1, 1, 1).expand_as(y))
        x = torch.randn(2, 32, 32, 32, device=dev)
        x[0, 3, 5, 7] = float("inf")
        x[1, 4, 20, 20] = float("nan")
        y, _ = _conv_plan(x, conv, math=math)
        ref = conv(x)
        assert not torch.isfinite(y[0, :, 4:7, 6:9]).any() and torch.isnan(y[1, :, 19:22, 19:22]).all()
        far = torch.isfinite(ref)
        far[0, :, :16, :16] = False   # the f16 mode scales per 16x16 tile: a non-finite value poisons its tile
        far[1, :, 16:, 16:] = False
        assert torch.allclose(y[far], ref[far], rtol=0, atol=1e-5 * float(ref[far].abs().max()))
    finally:
        nhwc.SPLIT_MIN_BLOCKS = old


def test_split_basic_block_matches_oracle(split_math):
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    old = nhwc.SPLIT_MIN_BLOCKS
    nhwc.SPLIT_MIN_BLOCKS = 1
    try:
        for cin, cout in ((64, 64), (96, 64)):
            blk = BasicBlock(cin, cout).to(dev)
            synthetic.fill_state_dict(blk, seed=11 + cin)
            x = torch.randn(2, cin, 48, 64, device=dev)
            y = blk(x)
            sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
            ref = onet.basic_block(x.cpu(), sd)
            assert _rel(y.cpu(), ref) < 1e-5
    finally:
        nhwc.SPLIT_MIN_BLOCKS = old


def test_zz_both_tile_heights_were_exercised():
    assert rows_used >= {8, 16}, rows_used


def test_split_abi_rejects_unsupported_shapes():
    """Loud failures, no silent fallback: Cout % 64 != 0 at pack time, a strided conv flagged as split."""
    import ctypes as C

    from implicit_depth_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda:0")
    assert L.idh_packed_split_weight_bytes(48, 32, 0, 11) == 0
    w = torch.randn(48, 32, 3, 3, device=dev)
    dst = torch.empty(1 << 16, device=dev, dtype=torch.int32)
    rc = L.idh_pack_conv_weight_split(w.data_ptr(), None, dst.data_ptr(), 48, 32, 0, 11, _lib.stream_ptr())
    assert rc == -2 and b"not covered" in L.idh_error_string(rc)  # IDH_EUNSUPPORTED
    assert L.idh_pack_conv_weight_split(w.data_ptr(), None, dst.data_ptr(), 64, 32, 0, 7, _lib.stream_ptr()) != 0  # unknown mode
    # a stride-2 conv may not be tagged for the split kernel
    conv = torch.nn.Conv2d(32, 64, 3, stride=2, padding=1).to(dev)
    p = nhwc.Plan(dev, math="f16x3")
    x = p.buffer(2, 32, 32, 32)
    out = p.buffer(2, 16, 16, 64)
    p.conv(x, conv, out)
    assert p.ops[-1].tile_m not in (10, 11)      # the planner never does it ...
    p.ops[-1].tile_m = 11                        # ... and the library refuses it
    p._arr = None
    with pytest.raises(_lib.IdhError):
        p.run()
