"""Run one conv layer config N times (for rocprofv3 --pmc): one_conv.py cin cout H W tm tn split [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import nhwc
cin, cout, H, W, tm, tn, split = [int(v) for v in sys.argv[1:8]]
B = int(sys.argv[8]) if len(sys.argv) > 8 else 4
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda(); syn.fill_state_dict(conv, 1)
x = torch.randn(B, H, W, cin, device="cuda")
p = nhwc.Plan(x.device)
out = p.buffer(B, H, W, cout)
nhwc.WINOGRAD, nhwc.WINO_MIN_TILES = tm == nhwc.TILE_WINO, 1  # Winograd kernel: the plan packs Winograd-domain weights
nhwc.WINOGRAD4, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL = tm == nhwc.TILE_WINO4, 1, 0.0
p.conv(nhwc.View(x, 0, cin), conv, out, act=1)
op = p.ops[0]; op.tile_m, op.tile_n, op.split_k = tm, tn, split
if split > 1:
    ws = torch.empty(split * B * H * W * cout, device="cuda"); p.keep.append(ws); op.ws = ws.data_ptr()
p._arr = None
for _ in range(iters): p.run()
torch.cuda.synchronize()
