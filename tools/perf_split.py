"""fp32-MFMA LDS conv vs split-bf16 conv, per layer shape: python tools/perf_split.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicit_depth_amd as idh
from implicit_depth_amd import nhwc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
shapes = [(64, 64, 192, 256), (128, 64, 192, 256), (192, 64, 192, 256), (64, 128, 96, 128), (128, 128, 96, 128), (256, 128, 96, 128),
          (384, 128, 96, 128), (256, 256, 48, 64), (512, 256, 48, 64), (384, 384, 24, 32)]
nhwc.SPLIT_MIN_BLOCKS = 1
if len(sys.argv) > 2:
    shapes = [(256, 128, 96, 128)]
for cin, cout, H, W in shapes:
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    x = torch.randn(N, H, W, cin, device=dev)
    row = []
    outs = []
    for math in ("fp32", "f16x3"):
        p = nhwc.Plan(dev, math=math)
        xin = nhwc.View(x, 0, cin)
        out = p.buffer(N, H, W, cout)
        res = p.buffer(N, H, W, cout); res.buf.zero_()
        p.conv(xin, conv, out, act=nhwc.ACT_LRELU, res=res)
        for _ in range(3):
            p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            p.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row.append((ms, p.flops / ms / 1e9, p.ops[0].tile_m))
        outs.append(out.dense().clone())
    err = [float((outs[0] - o).abs().max() / outs[0].abs().max()) for o in outs[1:]]
    print(f"N={N} {cin:4d}->{cout:4d} @{H}x{W}: fp32 {row[0][0]:6.3f} ms {row[0][1]:5.1f} TF | f16x3 {row[1][0]:6.3f} ms {row[1][1]:5.1f} x{row[0][0]/row[1][0]:.2f} d={err[0]:.1e}")
