"""Time one 3x3 conv layer on the LDS kernel: time_conv.py cin cout H W tm tn split [B] [res 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import nhwc
nhwc.WINOGRAD = False  # this tool forces the direct kernels' tile codes
cin, cout, H, W, tm, tn, split = [int(v) for v in sys.argv[1:8]]
B = int(sys.argv[8]) if len(sys.argv) > 8 else 32
use_res = len(sys.argv) > 9 and sys.argv[9] == "1"
conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda(); syn.fill_state_dict(conv, 1)
x = torch.randn(B, H, W, cin, device="cuda")
p = nhwc.Plan(x.device)
out = p.buffer(B, H, W, cout)
res = p.buffer(B, H, W, cout) if use_res else None
p.conv(nhwc.View(x, 0, cin), conv, out, act=1, res=res)
op = p.ops[0]; op.tile_m, op.tile_n, op.split_k = tm, tn, split
p._arr = None
for _ in range(5): p.run()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): p.run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
fl = 2.0 * B * H * W * cout * cin * 9
print(f"{cin}>{cout}@{H}x{W} B={B} t{tm}n{tn} res={int(use_res)}: {best * 1e3:.1f} us  {fl / best / 1e9:.1f} TFLOP/s ({fl / best / 1e9 / 157.3:.3f})")
