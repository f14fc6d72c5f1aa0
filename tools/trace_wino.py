"""Phase timeline of conv3x3_wino_k from the -DIDH_ABL_WINO_TRACE build (tools/trace_wino.sh): per-wave s_memtime stamps
-> where a workgroup's life goes (prologue, first DMA wait, K steps, barrier waits, epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch import nn
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import nhwc

cin, cout, H, W, tn = [int(v) for v in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 32
cin2 = int(sys.argv[7]) if len(sys.argv) > 7 else 0  # fused 1x1 projection of a cin2-channel tensor
waves = 4
nhwc.WINOGRAD, nhwc.WINO_MIN_TILES = True, 1
conv = nn.Conv2d(cin, cout, 3, 1, 1).cuda(); syn.fill_state_dict(conv, 1)
x = torch.randn(B, H, W, cin, device="cuda")
p = nhwc.Plan(x.device)
out = p.buffer(B, H, W, cout)
proj = x2 = None
if cin2:
    proj = nn.Conv2d(cin2, cout, 1).cuda(); syn.fill_state_dict(proj, 2)
    x2 = torch.randn(B, H, W, cin2, device="cuda")
p.conv(nhwc.View(x, 0, cin), conv, out, act=1, x2=None if proj is None else nhwc.View(x2, 0, cin2), conv2=proj)
op = p.ops[0]
rows = 8
tiles = B * (-(-H // rows)) * (-(-W // 32)) * (cout // 32)
per_cu = 2
blocks = min(tiles, 256 * per_cu)
tr = torch.zeros(blocks * waves * 64, dtype=torch.int64, device="cuda")
op.ws = tr.data_ptr()
p._arr = None
for _ in range(3): p.run()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(blocks, waves, 64).astype(np.int64)
nS = (cin + 15) // 16 * 2 + ((cin2 + 15) // 16 + 1) // 2
print(f"{cin}->{cout} @{H}x{W} B={B} tile_n={tn}: {tiles} tiles on {blocks} persistent workgroups x {waves} waves, {nS} K steps per tile")
med = lambda a: float(np.median(a))
print(f"  workgroup life (entry -> exit)  {med(t[:, :, 62] - t[:, :, 0]):9.0f} cycles = {tiles / blocks:.1f} tiles -> {med(t[:, :, 62] - t[:, :, 0]) / (tiles / blocks):.0f} per tile")
print(f"  entry -> first DMA issued       {med(t[:, :, 1] - t[:, :, 0]):9.0f}")
print(f"  first DMA issued -> barrier     {med(t[:, :, 2] - t[:, :, 1]):9.0f}")
per_tile = 2 * nS + 1
ntile = (61 - 3) // per_tile
for k in range(ntile):
    b = 2 + k * per_tile  # stamp before the tile's first K step
    comp = np.stack([t[:, :, b + 1 + 2 * c] - t[:, :, b + 2 * c] for c in range(nS)], -1)
    bar = np.stack([t[:, :, b + 2 + 2 * c] - t[:, :, b + 1 + 2 * c] for c in range(nS)], -1)
    epi = t[:, :, b + 2 * nS + 1] - t[:, :, b + 2 * nS]
    print(f"  tile {k}: K step compute median {med(comp):7.0f} (first {med(comp[..., 0]):.0f} last {med(comp[..., -1]):.0f} p10 {np.percentile(comp, 10):.0f} p90 {np.percentile(comp, 90):.0f})"
          f"  barrier wait {med(bar):6.0f} (mean {bar.mean():.0f} p90 {np.percentile(bar, 90):.0f})  epilogue {med(epi):6.0f}  tile total {med(t[:, :, b + 2 * nS + 1] - t[:, :, b]):.0f}")
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"trace_wino_{cin}_{cout}_{tn}.npy"), t[:64])
