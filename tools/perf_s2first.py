"""conv1 of the three stride-2 BasicBlocks of the CVEncoder at B frames: direct-fragment kernel (conv_mfma_k) vs the LDS-staged kernel's
stride-2 loader (nhwc.S2_FIRST): python tools/perf_s2first.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from implicit_depth_amd import nhwc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
for (cin, cout, H, W) in ((64, 128, 96, 128), (128, 256, 48, 64), (256, 384, 24, 32)):
    conv = nn.Conv2d(cin, cout, 3, 2, 1, bias=False).to(dev)
    res = {}
    xt = torch.randn(B, H, W, cin, device=dev)
    for name, on in (("conv_mfma_k", False), ("lds stride-2 loader", True)):
        nhwc.S2_FIRST = on
        p = nhwc.Plan(dev)
        x = p.buffer(B, H, W, cin); x.dense().copy_(xt)
        y = p.buffer(B, H // 2, W // 2, cout)
        p.conv(x, conv, y, act=nhwc.ACT_LRELU, slope=0.2)
        p.schedule()
        for _ in range(3): p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        op = p.ops[0]
        res[name] = y.dense().clone()
        fl = 2 * 9 * cin * cout * B * (H // 2) * (W // 2)
        print(f"{cin}->{cout} @{H//2}x{W//2} B={B} {name:22s} t{op.tile_m}n{op.tile_n}s{op.split_k}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s")
    a, b = res.values()
    print(f"   max|diff| between the two kernels {float((a-b).abs().max()):.2e} (scale {float(a.abs().max()):.2f})")
nhwc.S2_FIRST = True
