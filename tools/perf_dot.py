"""Time one of the three dot-volume kernels alone: python tools/perf_dot.py <kernel 0|1|2|3> [B] [K] [D] [iters]
(0 = the launcher's choice).  Prints ms per launch, algorithmic GB/s (SURVEY 8d bytes) and frames/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from implicit_depth_amd import _lib
from implicit_depth_amd.cost_volume import volume_opts

kern = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D = int(sys.argv[4]) if len(sys.argv) > 4 else 64
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
a = bench.parse(["--workload", "warp_match_dot", "--views", str(K), "--planes", str(D)])
a.batch = B
w = bench.WarpMatchDot(a, torch.device("cuda:0"), 0)
p = _lib.ptr
opts, _keep = volume_opts(w.B, w.K, w.C, w.H, w.W, w.D, None, 0, 0, kernel=kern, dot_scratch_device=None if os.environ.get('IDH_NO_SCRATCH') else torch.device('cuda:0'))


def step():
    _lib.check(w.L.idh_cost_volume_dot_ex_fwd(p(w.cur), p(w.src), p(w.Ks), p(w.E), p(w.invK), 0.25, 5.0, w.B, w.K, w.C, w.H, w.W, w.D, p(w.cost), 0,
                                              p(w.lowest), p(w.planes), opts, _lib.stream_ptr()), "dot")


for _ in range(5):
    step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"kernel={kern} B={B} K={K} D={D}: {ms:.4f} ms/launch  {w.algorithmic_bytes_per_launch() / ms / 1e6:.1f} GB/s algorithmic  {B / ms * 1e3:.0f} frames/s  "
      f"({ms / B * 1e3:.2f} us/frame)")

if os.environ.get("IDH_PRINT_LOWEST"):
    lo = w.lowest.float()
    print("lowest: mean planes per lane", float((lo % 1000).mean()), "mean runs per lane", float((lo // 1000).mean()))
    t = (lo % 1000).view(w.B, w.H // 8, 8, w.W // 32, 32)[:, :, 0, :, 0].flatten().cpu()
    print("planes per tile: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f  mean %.1f" % (t.min(), t.quantile(0.1), t.median(), t.quantile(0.9), t.max(), t.mean()))
    print("per frame 0 tiles (12 rows x 4 cols):", t[:48].view(12, 4).int().tolist())

if os.environ.get("IDH_PRINT_TRACE"):
    import collections
    PS = int(os.environ.get("IDH_TRACE_PSPLIT", "1"))
    nwg = 48 * PS
    t = w.lowest.view(w.B, -1)[:, : nwg * 16].view(w.B, nwg, 4, 4).float().cpu()  # frame, workgroup, wave, {total, start >> 6, cu, compute}
    print("ticks per workgroup (wave 0): total mean %.0f max %.0f ; compute mean %.0f" % (t[:, :, 0, 0].mean(), t[:, :, 0, 0].max(), t[:, :, 0, 3].mean()))
    w0 = t[:, :, 0]
    per_cu = collections.defaultdict(list)
    for f in range(w.B):
        for i in range(nwg):
            tot, st, cu, _ = w0[f, i].tolist()
            per_cu[int(cu)].append((st * 64.0, st * 64.0 + tot))
    print("distinct CUs seen:", len(per_cu), " workgroups per CU: min %d max %d" % (min(len(v) for v in per_cu.values()), max(len(v) for v in per_cu.values())))
    conc = []
    for cu, iv in per_cu.items():
        ev = sorted([(a0, 1) for a0, _ in iv] + [(b0, -1) for _, b0 in iv])
        c = mx = 0
        for _, d in ev:
            c += d; mx = max(mx, c)
        conc.append(mx)
    print("max concurrent workgroups on a CU: histogram", collections.Counter(conc))
    busy = [sum(b0 - a0 for a0, b0 in iv) for iv in per_cu.values()]
    span = [max(b0 for _, b0 in iv) - min(a0 for a0, _ in iv) for iv in per_cu.values()]
    print("per-CU sum of workgroup times: mean %.0f max %.0f ; per-CU span first start -> last end: mean %.0f min %.0f max %.0f" % (sum(busy) / len(busy), max(busy), sum(span) / len(span), min(span), max(span)))
    cu0 = sorted(per_cu.keys())[5]
    iv = sorted(per_cu[cu0]); base = iv[0][0]
    print("CU", cu0, "timeline (start, end) in kiloticks:", [(round((a0 - base) / 1e3), round((b0 - base) / 1e3)) for a0, b0 in iv])
    if PS > 1:
        bysp = t[:, :, 0, 0].view(w.B, 48, PS).mean((0, 1))
        print("mean workgroup ticks by plane split index:", [int(x) for x in bysp], " by tile column:", [int(x) for x in t[:, :, 0, 0].view(w.B, 12, 4, PS).mean((0, 1, 3))])
