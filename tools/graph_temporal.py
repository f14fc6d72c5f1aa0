"""Experiment: the temporal loop (B = 1, D = 96, prior carried) replayed as ONE captured graph per frame against the eager launch sequence.
  python tools/graph_temporal.py [frames]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ap_args = bench.parse(["--workload", "temporal", "--no-cpu-baseline"])
dev = torch.device("cuda:0")
wl = bench.TemporalWorkload(ap_args, dev, 0)
for _ in range(8):
    wl.step()
torch.cuda.synchronize()


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


eager = min(timed(wl.step, frames) for _ in range(3))
ref_out = wl.out["pred_0"].clone()
# static inputs of the captured frame
prev_pred = wl.prev[0].clone()
prev_cam = wl.prev[1].clone()
wTc = wl.poses[0][0].clone()
static = {"prior_prediction": prev_pred, "prior_cam_T_world": prev_cam, "world_T_cam_b44": wTc, "K_s0_b44": wl.K0, "invK_s0_b44": wl.invK0}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        out = wl._forward(prior_inputs=static)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = wl._forward(prior_inputs=static)
    nxt = torch.sigmoid(out["pred_0"])
t = [0]


def graph_step():
    wT, cT = wl.poses[t[0] % len(wl.poses)]
    wTc.copy_(wT)
    g.replay()
    prev_pred.copy_(nxt)
    prev_cam.copy_(cT)
    t[0] += 1


for _ in range(8):
    graph_step()
graphed = min(timed(graph_step, frames) for _ in range(3))
print(f"temporal B=1 D={wl.D}: eager {eager:.3f} ms/frame ({1e3 / eager:.1f} frames/s)   graph replay {graphed:.3f} ms/frame ({1e3 / graphed:.1f} frames/s)")
# same arithmetic: replay one frame from identical inputs both ways
wl.prev = (prev_pred.clone(), prev_cam.clone()); wl.t = t[0]
wT, cT = wl.poses[t[0] % len(wl.poses)]
wTc.copy_(wT); g.replay(); torch.cuda.synchronize()
a = out["pred_0"].clone()
wl.step(); torch.cuda.synchronize()
print("bit-identical to the eager frame:", bool(torch.equal(a, wl.out["pred_0"])))
