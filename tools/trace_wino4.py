"""Phase timeline of conv3x3_wino4_k from the -DIDH_ABL_W4_TRACE build (tools/abl_wino4.sh trace): per-wave s_memtime stamps along a
workgroup's second tile.

  IDH_LIB=.../libidh_ablw4_TRACE.so python tools/trace_wino4.py cin cout H W [B] [res]
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
nhwc.WINOGRAD4, nhwc.WINO4_MIN_TILES, nhwc.WINO4_MIN_FILL = True, 1, 0.0
conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda(); syn.fill_state_dict(conv, 1)
x = torch.randn(B, H, W, cin, device="cuda")
res = torch.randn(B, H, W, cout, device="cuda") if use_res else None
p = nhwc.Plan(x.device)
out = p.buffer(B, H, W, cout)
p.conv(nhwc.View(x, 0, cin), conv, out, act=1, res=None if res is None else nhwc.View(res, 0, cout))
op = p.ops[0]
assert op.tile_m == nhwc.TILE_WINO4
tiles = B * (-(-H // 8)) * (-(-W // 32)) * (cout // 64)
blocks = min(tiles, 512)
tr = torch.zeros(blocks * 4 * 80, dtype=torch.int64, device="cuda")
op.ws = tr.data_ptr()
p._arr = None
for _ in range(3): p.run()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(blocks, 4, 80).astype(np.int64)
nS = (cin + 15) // 16 * 2
ns = min(nS, 8)
med = lambda a: float(np.median(a))
p90 = lambda a: float(np.percentile(a, 90))
print(f"{cin}->{cout} @{H}x{W} B={B} res={use_res}: {tiles} tiles on {blocks} workgroups, {nS} stages per tile (own MFMA time per stage and wave: 2304 cycles; two waves per SIMD)")
st = np.stack([t[:, :, 1 + 8 * c: 9 + 8 * c] for c in range(ns)], 2)  # blocks, waves, stage, 8
d = np.diff(st, axis=-1)
names = ["halo loads issued", "MFMA rows 0-5", "rows 6-11", "rows 12-17", "halo written", "transform", "barrier wait"]
print(f"  stage total: median {med(st[..., 7] - st[..., 0]):.0f}  p90 {p90(st[..., 7] - st[..., 0]):.0f}   per stage " + " ".join(f"{med(st[:, :, c, 7] - st[:, :, c, 0]):.0f}" for c in range(ns)))
for i, nme in enumerate(names):
    print(f"  {nme:18s} median {med(d[..., i]):6.0f}  p90 {p90(d[..., i]):6.0f}   per stage " + " ".join(f"{med(d[:, :, c, i]):.0f}" for c in range(ns)))
for w in range(4):
    print(f"  wave {w}: " + "  ".join(f"{nme} {med(d[:, w, :, i]):.0f}" for i, nme in enumerate(names)))
print(f"  epilogue {med(t[:, :, 71] - t[:, :, 70]):.0f} (p90 {p90(t[:, :, 71] - t[:, :, 70]):.0f}); tile start -> first stage {med(t[:, :, 1] - t[:, :, 0]):.0f}; whole tile {med(t[:, :, 71] - t[:, :, 0]):.0f}")
