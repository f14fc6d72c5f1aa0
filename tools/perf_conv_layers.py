"""Per-layer-shape sweep of the conv kernel's tile / split-K choices (TFLOP/s per config)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import nhwc
nhwc.WINOGRAD = False  # this tool forces the direct kernels' tile codes

B = int(os.environ.get("B", 4))
# (cin, cout, H, W, ks, stride)  representative layers of CVEncoder / BDDecoderPP @512x384
LAYERS = [(64, 64, 192, 256, 3, 1), (192, 64, 192, 256, 3, 1), (24, 64, 192, 256, 3, 1), (64, 64, 96, 128, 3, 1), (192, 64, 96, 128, 3, 1),
          (112, 64, 96, 128, 3, 1), (128, 128, 48, 64, 3, 1), (384, 128, 48, 64, 3, 1), (256, 256, 24, 32, 3, 1), (416, 256, 24, 32, 3, 1),
          (384, 384, 12, 16, 3, 1), (640, 384, 12, 16, 3, 1)]
sel = os.environ.get("LAYERS")
if sel:
    LAYERS = [LAYERS[int(i)] for i in sel.split(",")]

def time_plan(p, n=8):
    p.run(); p.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): p.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (cin, cout, H, W, ks, st) in LAYERS:
    conv = nn.Conv2d(cin, cout, ks, st, ks // 2).cuda()
    syn.fill_state_dict(conv, 1)
    x = torch.randn(B, H, W, cin, device="cuda")
    res = []
    nsub = cout // 16
    for tm, tn in list(itertools.product((4, 2, 1), (4, 2))) + [(8, 0), (9, 0)]:
        if tn and nsub % tn: continue
        for split in (1, 2, 4, 8):
            if tm >= 8:
                waves = 4 * B * (-(-(H // st) // (8 if tm == 8 else 4))) * (-(-(W // st) // 16)) * (cout // 64) * split
                if split > 1 and waves > 8192 * 2: continue
                if split > cin // 16: continue
            else:
              pass
            if tm >= 8 and (cout % 64 or ks != 3 or st != 1): continue
            M = B * (H // st) * (W // st)
            if tm < 8:
                waves = -(-M // (16 * tm)) * (nsub // tn) * split
                if split > 1 and waves > 8192: continue
                if waves < 256: continue
            p = nhwc.Plan(x.device)
            xin = nhwc.View(x, 0, cin)
            out = p.buffer(B, H // st, W // st, cout)
            p.conv(xin, conv, out, act=1)
            op = p.ops[0]
            op.tile_m, op.tile_n, op.split_k = tm, tn, split
            if split > 1:
                ws = torch.empty(split * M * cout, device="cuda"); p.keep.append(ws); op.ws = ws.data_ptr()
            p._arr = None
            ms = time_plan(p)
            res.append((p.flops / ms / 1e9, tm, tn, split, waves, ms))
    res.sort(reverse=True)
    auto = nhwc.choose_tiles(B * (H // st) * (W // st), cout, ks * ks * ((cin + 15) // 16))
    print(f"{cin:4d}->{cout:4d} {H}x{W} k{ks}s{st} B={B}  auto={auto}  " + "  ".join(f"[{tm},{tn},s{s}: {tf:.0f}TF {ms*1e3:.0f}us w{w}]" for tf, tm, tn, s, w, ms in res[:6]), flush=True)
