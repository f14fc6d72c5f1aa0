"""A/B of the Winograd F(4x4,3x3) conv kernel (csrc/conv_wino4.hip) against the F(2x2) kernel and the direct LDS kernel, layer by
layer in ONE process (interleaved rounds), plus the scale-relative error of each against an fp64 torch reference.

  python tools/perf_wino4.py [B] [rounds]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import torch.nn.functional as F
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import nhwc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# (cin, cout, H, W, residual): residual 0 none, 1 identity residual, c >= 16: fused 1x1 projection of a c-channel tensor (BasicBlock conv2 + downsample)
LAYERS = [(64, 64, 192, 256, 0), (64, 64, 192, 256, 1), (192, 64, 192, 256, 0), (64, 64, 96, 128, 1), (192, 64, 96, 128, 0), (128, 128, 48, 64, 1),
          (384, 128, 48, 64, 0), (256, 256, 24, 32, 1), (64, 64, 192, 256, 192), (64, 64, 192, 256, 128), (64, 64, 192, 256, 24), (64, 64, 96, 128, 192),
          (128, 128, 48, 64, 384), (256, 256, 24, 32, 512)]
sel = os.environ.get("LAYERS")
if sel:
    LAYERS = [LAYERS[int(i)] for i in sel.split(",")]
VARIANTS = [("direct8", 8), ("wino2", nhwc.TILE_WINO), ("wino4", nhwc.TILE_WINO4)]
if os.environ.get("VARIANTS"):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ["VARIANTS"].split(",")]


def build(conv, x, res, tm, proj=None, x2=None):
    p = nhwc.Plan(x.device)
    Bn, H, W, cin = x.shape
    out = p.buffer(Bn, H, W, conv.out_channels)
    old = (nhwc.WINOGRAD, nhwc.WINOGRAD4, nhwc.WINO_MIN_TILES, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL)
    nhwc.WINOGRAD, nhwc.WINOGRAD4, nhwc.WINO_MIN_TILES, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL = tm == nhwc.TILE_WINO, tm == nhwc.TILE_WINO4, 1, 1, 0.0
    try:
        p.conv(nhwc.View(x, 0, cin), conv, out, act=1, slope=0.2, res=None if res is None else nhwc.View(res, 0, conv.out_channels),
               x2=None if proj is None else nhwc.View(x2, 0, proj.in_channels), conv2=proj)
    finally:
        nhwc.WINOGRAD, nhwc.WINOGRAD4, nhwc.WINO_MIN_TILES, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL = old
    op = p.ops[0]
    if tm == 8:
        op.tile_m, op.tile_n = 8, 0
    else:
        assert op.tile_m == tm, (op.tile_m, tm)
    p._arr = None
    return p, out


def time_plan(p, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        p.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (cin, cout, H, W, use_res) in LAYERS:
    conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda()
    syn.fill_state_dict(conv, 1)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, H, W, cin, device="cuda", generator=g)
    res = torch.randn(B, H, W, cout, device="cuda", generator=g) if use_res == 1 else None
    proj = x2 = None
    if use_res >= 16:
        proj = nn.Conv2d(use_res, cout, 1).cuda()
        syn.fill_state_dict(proj, 2)
        x2 = torch.zeros(B, H, W, nhwc.ceil16(use_res), device="cuda")
        x2[..., :use_res] = torch.randn(B, H, W, use_res, device="cuda", generator=g)
    plans = [(name,) + build(conv, x, res, tm, proj, x2) for name, tm in VARIANTS]
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double(), padding=1)
    if use_res == 1:
        ref = ref + res[:nb].permute(0, 3, 1, 2).double()
    if proj is not None:
        ref = ref + F.conv2d(x2[:nb, ..., :use_res].permute(0, 3, 1, 2).double(), proj.weight.double(), proj.bias.double())
    ref = F.leaky_relu(ref, 0.2).permute(0, 2, 3, 1)
    errs = {}
    for name, p, out in plans:
        p.run()
        torch.cuda.synchronize()
        errs[name] = ((out.dense()[:nb].double() - ref).abs().max() / ref.abs().max()).item()
    best = {name: 1e9 for name, _, _ in plans}
    for _ in range(ROUNDS):
        for name, p, _ in plans:
            for _ in range(3):
                p.run()
            best[name] = min(best[name], time_plan(p, 10))
    fl = 2.0 * B * H * W * cout * (cin * 9 + (use_res if use_res >= 16 else 0))
    base = best[plans[0][0]]
    print(f"{cin:3d}->{cout:3d} @{H}x{W} B={B} res={use_res}: " + "  ".join(
        f"{name} {best[name] * 1e3:7.1f} us {fl / best[name] / 1e9:6.1f} TF-equiv x{base / best[name]:.2f} err {errs[name]:.1e}" for name, _, _ in plans), flush=True)
