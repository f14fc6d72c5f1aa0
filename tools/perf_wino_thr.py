"""Whole-step time of the hot path against nhwc.WINO_MIN_TILES (grouped Winograd launches on): python tools/perf_wino_thr.py [batches...]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implicit_depth_amd import nhwc
from bench import HotPathWorkload

THR = [int(v) for v in os.environ.get("THR", "24,48,96,128,256").split(",")]
for b in [int(v) for v in sys.argv[1:]] or [1, 2, 4]:
    res = {}
    for rep in range(2):
        for thr in THR:
            nhwc.WINO_MIN_TILES = thr
            a = argparse.Namespace(batch=b, views=7, planes=64, height=384, width=512, volume="mlp")
            wl = HotPathWorkload(a, torch.device("cuda"), 0)
            for _ in range(3): wl.step()
            torch.cuda.synchronize()
            n = max(10, 200 // b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): wl.step()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(thr, []).append(e0.elapsed_time(e1) / n)
            del wl
    print(f"B={b}: " + "  ".join(f"min_tiles {t}: {min(v):.3f} ms" for t, v in res.items()))
