"""Hot-path step time with the decoder's upsample + concat fused into the consumer convs (8- or 4-row tiles) or
materialised by upsample2_k: python tools/perf_up.py [batch]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from implicit_depth_amd import nhwc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
outs = {}
for name, fuse, rows in (("materialised", False, 8), ("fused, 8-row tiles", True, 8), ("fused, 4-row tiles", True, 4)):
    nhwc.FUSE_UPSAMPLE, nhwc.FUSED_UP_ROWS = fuse, rows
    a = bench.parse([])
    a.batch = B
    wl = bench.HotPathWorkload(a, torch.device("cuda:0"), 0)
    with torch.inference_mode():
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            wl.step()
        e1.record()
        torch.cuda.synchronize()
    outs[name] = wl.out["pred_0"].clone()
    n_ops = len(next(iter(wl.model._plans.values()))["plan"].ops)
    print(f"B={B} {name:22s}: {e0.elapsed_time(e1) / 10:.3f} ms/step, {n_ops} ops")
    del wl
    torch.cuda.empty_cache()
ref = outs["materialised"]
for k, v in outs.items():
    print(f"  {k}: max|diff| vs materialised = {float((v - ref).abs().max()):.3e}")
