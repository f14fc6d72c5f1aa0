"""upsample2_k alone at the hot path's four decoder shapes (B frames, 64..256 channels into a 3x wider concat buffer) + bit check
against F.interpolate: python tools/perf_upsample.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from implicit_depth_amd import nhwc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
tot_ms = tot_b = 0.0
for (C, H, W) in ((64, 96, 128), (64, 48, 64), (128, 24, 32), (256, 12, 16)):
    p = nhwc.Plan(dev)
    x = p.buffer(B, H, W, C)
    cat = p.buffer(B, 2 * H, 2 * W, 3 * C)
    xt = torch.randn(B, H, W, C, device=dev)
    x.dense().copy_(xt)
    p.upsample2(x, cat.slice(C, C))
    p.schedule()
    for _ in range(3): p.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): p.run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = B * H * W * C * 4 * 5
    ref = F.interpolate(xt.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    got = cat.dense()[..., C:2 * C]
    print(f"{C:4d}ch {H}x{W} -> {2*H}x{2*W}: {ms*1e3:8.1f} us  {by/ms/1e9:6.2f} TB/s  max|diff| vs F.interpolate {float((got-ref).abs().max()):.2e}")
    tot_ms += ms; tot_b += by
print(f"sum {tot_ms*1e3:.1f} us, {tot_b/tot_ms/1e9:.2f} TB/s")
