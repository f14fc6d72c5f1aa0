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
opts, _ = volume_opts(w.B, w.K, w.C, w.H, w.W, w.D, None, 0, 0, kernel=kern)


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
