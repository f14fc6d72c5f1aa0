"""Does splitting the 32-frame batch over two HIP streams hide the per-launch fill / drain of the conv kernels?
python tools/perf_two_streams.py [B_total] [n_streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")


def mk(B):
    a = bench.parse([])
    a.batch = B
    return bench.HotPathWorkload(a, dev, 0)


with torch.inference_mode():
    one = mk(Bt)
    for _ in range(3):
        one.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        one.step()
    e1.record()
    torch.cuda.synchronize()
    print(f"1 stream  x B={Bt}: {e0.elapsed_time(e1) / 10:.3f} ms per {Bt} frames")
    del one
    torch.cuda.empty_cache()
    parts = [mk(Bt // ns) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for w, s in zip(parts, streams):
        with torch.cuda.stream(s):
            for _ in range(3):
                w.step()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        for w, s in zip(parts, streams):
            s.wait_stream(torch.cuda.current_stream()) if False else None
            with torch.cuda.stream(s):
                w.step()
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    print(f"{ns} streams x B={Bt // ns}: {e0.elapsed_time(e1) / 10:.3f} ms per {Bt} frames")
