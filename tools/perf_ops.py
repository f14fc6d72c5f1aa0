"""Per-op timing of the hot-path plan (each op replayed alone)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import networks as net, nhwc, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W, D = 384, 512, 64
pyr = [p.cuda() for p in syn.encoder_pyramid(B, H, W, seed=0)]
cvol = syn.randn((B, D, H // 4, W // 4), 0, "cv").cuda()
cve = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384]).cuda()
dec = net.BDDecoderPP([24, 64, 128, 256, 384]).cuda()
syn.fill_state_dict(cve, 1); syn.fill_state_dict(dec, 2)
outs = cve(cvol, pyr[1:]); dec([pyr[0]] + outs)
torch.cuda.synchronize()
L = _lib.lib()
rows = []
KIND = {1: "conv", 2: "up2", 3: "imp", 4: "exp", 6: "head"}
for name, m in (("cve", cve), ("dec", dec)):
    p = list(m._idh_plans.values())[0][0]
    for idx, op in enumerate(p.ops):
        arr = (nhwc.Op * 1)(op)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2): L.idh_run_ops(C.cast(arr, C.c_void_p), 1, _lib.stream_ptr())
        n = 5
        e0.record()
        for _ in range(n): L.idh_run_ops(C.cast(arr, C.c_void_p), 1, _lib.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 0
        desc = KIND.get(op.kind, str(op.kind))
        if op.kind == 1:
            for s in op.src:
                if s.in_: fl += 2 * op.N * op.Ho * op.Wo * op.Cout * s.Cin * s.ks * s.ks
            desc += f" {op.src[0].Cin}" + (f"+{op.src[1].Cin}(k{op.src[1].ks}s{op.src[1].stride})" if op.src[1].in_ else "") + f"->{op.Cout} {op.Ho}x{op.Wo} k{op.src[0].ks}s{op.src[0].stride} t{op.tile_m},{op.tile_n} s{op.split_k}"
        else:
            desc += f" C{op.src[0].Cin} {op.src[0].H}x{op.src[0].W}"
        rows.append((ms, fl, name, idx, desc))
tot = sum(r[0] for r in rows)
print(f"B={B} total isolated {tot:.3f} ms")
agg = {}
for ms, fl, name, idx, desc in rows:
    k = desc
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{ms:8.3f} ms {100*ms/tot:5.1f}%  x{n:<3d} {fl/ms/1e9 if ms else 0:6.1f} TF  {k}")
