"""hipGraph probe: capture one hot-path step with torch.cuda.CUDAGraph and compare replay time with eager launches.
python tools/graph_probe.py <batch> <math>"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import HotPathWorkload

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
math = sys.argv[2] if len(sys.argv) > 2 else "fp32"
args = argparse.Namespace(batch=B, views=7, planes=64, height=384, width=512, volume="mlp", conv_math=math, mlp_math="f16x3" if math == "f16x3" else "fp32")
dev = torch.device("cuda:0")
wl = HotPathWorkload(args, dev, 0)
with torch.inference_mode():
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    def timeit(fn, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = timeit(wl.step)
    ref = wl.out["pred_0"].clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        wl.step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        wl.step()
    out = wl.out["pred_0"]
    g.replay(); torch.cuda.synchronize()
    graph = timeit(g.replay)
    print(f"B={B} {math}: eager {eager:.3f} ms/step ({B/eager*1e3:.1f} fps)  graph {graph:.3f} ms/step ({B/graph*1e3:.1f} fps)  max diff {float((out-ref).abs().max()):.2e}")
