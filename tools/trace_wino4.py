"""Phase timeline of conv3x3_wino4_k from the -DIDH_ABL_W4_TRACE build (tools/abl_wino4.sh trace): per-wave s_memtime stamps along
a workgroup's second tile -> cycles per stage part (vertical transform block, the 12 row iterations, barrier wait) and per epilogue part.

  IDH_LIB=.../libidh_ablw4_TRACE.so python tools/trace_wino4.py cin cout H W [B]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch import nn
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import nhwc

cin, cout, H, W = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
use_res = int(sys.argv[6]) if len(sys.argv) > 6 else 0
nhwc.WINOGRAD4, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL, nhwc.WINO4_MIN_CIN = True, 1, 0.0, 0
conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda(); syn.fill_state_dict(conv, 1)
x = torch.randn(B, H, W, cin, device="cuda")
res = torch.randn(B, H, W, cout, device="cuda") if use_res else None
p = nhwc.Plan(x.device)
out = p.buffer(B, H, W, cout)
p.conv(nhwc.View(x, 0, cin), conv, out, act=1, res=None if res is None else nhwc.View(res, 0, cout))
op = p.ops[0]
assert op.tile_m == nhwc.TILE_WINO4
tiles = B * (-(-H // 16)) * (-(-W // 64)) * (cout // 32)
blocks = min(tiles, 256)
tr = torch.zeros(blocks * 4 * 160, dtype=torch.int64, device="cuda")
op.ws = tr.data_ptr()
p._arr = None
for _ in range(3): p.run()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(blocks, 4, 160).astype(np.int64)
nS = (cin + 15) // 16 * 2
ns = min(nS, 8)
med = lambda a: float(np.median(a))
print(f"{cin}->{cout} @{H}x{W} B={B} res={use_res}: {tiles} tiles on {blocks} workgroups, {nS} stages per tile (MFMA time per stage: 4608 cycles)")
st = np.stack([t[:, :, 1 + 15 * c: 1 + 15 * c + 15] for c in range(ns)], 2)  # blocks, waves, stage, 15
d = np.diff(st, axis=-1)
print(f"  stage total (entry -> barrier passed): median {med(st[..., 14] - st[..., 0]):.0f}  per stage " + " ".join(f"{med(st[:, :, c, 14] - st[:, :, c, 0]):.0f}" for c in range(ns)))
print(f"  vertical block {med(d[..., 0]):.0f}   barrier wait {med(d[..., 13]):.0f} (p90 {np.percentile(d[..., 13], 90):.0f})")
print("  iterations (median over waves/stages): " + " ".join(f"{med(d[..., 1 + i]):.0f}" for i in range(12)))
print("  iterations p90:                        " + " ".join(f"{np.percentile(d[..., 1 + i], 90):.0f}" for i in range(12)))
for c in range(ns):
    print(f"  stage {c}: start {med(d[:, :, c, 0]):.0f} | " + " ".join(f"{med(d[:, :, c, 1 + i]):.0f}" for i in range(12)) + f" | barrier {med(d[:, :, c, 13]):.0f}")
rotp = ((np.arange(blocks) >> 4) % nS) & 1
for rp in (0, 1):
    sel = rotp == rp
    print(f"  workgroups with rot parity {rp}: even stages " + " ".join(f"{med(d[sel][:, :, 0:ns:2, 1 + i]):.0f}" for i in range(12)) + "   odd stages: " + " ".join(f"{med(d[sel][:, :, 1:ns:2, 1 + i]):.0f}" for i in range(12)))
for w in range(0):
    print(f"  wave {w}, even stages: " + " ".join(f"{med(d[:, w, 0:ns:2, 1 + i]):.0f}" for i in range(12)) + "   odd stages: " + " ".join(f"{med(d[:, w, 1:ns:2, 1 + i]):.0f}" for i in range(12)))
sl = t[:, :, 130:154]
print("  stage 2, slots of iteration 6 then 7 (cycles since the previous stamp): " + " ".join(f"{med(sl[:, :, i] - (sl[:, :, i - 1] if i else st[:, :, 2, 7])):.0f}" for i in range(24)))
for par in (0, 1):
    print(f"  parity {par} stages: iterations " + " ".join(f"{med(d[:, :, par::2, 1 + i]):.0f}" for i in range(12)) + f"  barrier {med(d[:, :, par::2, 13]):.0f}")
if nS <= 8:
    print(f"  epilogue: K loop done -> first copies issued {med(t[:, :, 126] - t[:, :, 125]):.0f}; row pass block 0 {med(t[:, :, 127] - t[:, :, 126]):.0f}; block 1 {med(t[:, :, 128] - t[:, :, 127]):.0f}"
          f"; column pass + stores {med(t[:, :, 129] - t[:, :, 128]):.0f}; tile start -> first stage {med(t[:, :, 1] - t[:, :, 0]):.0f}; whole tile {med(t[:, :, 129] - t[:, :, 0]):.0f}")
else:
    print(f"  epilogue: row pass block 0 {med(t[:, :, 127] - t[:, :, 126]):.0f}; block 1 {med(t[:, :, 128] - t[:, :, 127]):.0f}; column pass + stores {med(t[:, :, 129] - t[:, :, 128]):.0f}; whole tile {med(t[:, :, 129] - t[:, :, 0]):.0f}")
