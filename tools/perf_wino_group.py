"""Whole-step time of the hot path with nhwc.WINO_GROUP off / on: python tools/perf_wino_group.py [batches...]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implicit_depth_amd import nhwc
from bench import HotPathWorkload

for b in [int(v) for v in sys.argv[1:]] or [1, 4, 8, 32]:
    res = {}
    for merge in (False, True, False, True):
        nhwc.WINO_GROUP = merge
        a = argparse.Namespace(batch=b, views=7, planes=64, height=384, width=512, volume="mlp")
        wl = HotPathWorkload(a, torch.device("cuda"), 0)
        for _ in range(3): wl.step()
        torch.cuda.synchronize()
        n = max(10, 200 // b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): wl.step()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(merge, []).append(e0.elapsed_time(e1) / n)
        del wl
    print(f"B={b}: group off {min(res[False]):.3f} ms, on {min(res[True]):.3f} ms ({100 * (min(res[True]) / min(res[False]) - 1):+.2f} %)")
