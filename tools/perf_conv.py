"""Timing of the conv stage (CVEncoder + BDDecoderPP) at BASELINE size; prints TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import networks as net, nhwc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W, D = 384, 512, 64
pyr = [p.cuda() for p in syn.encoder_pyramid(B, H, W, seed=0)]
cvol = syn.randn((B, D, H // 4, W // 4), 0, "cv").cuda()
cve = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384]).cuda()
dec = net.BDDecoderPP([24, 64, 128, 256, 384]).cuda()
syn.fill_state_dict(cve, 1); syn.fill_state_dict(dec, 2)

def run():
    outs = cve(cvol, pyr[1:])
    return dec([pyr[0]] + outs)

for _ in range(3): run()
torch.cuda.synchronize()
flops = sum(p[0].flops for m in (cve, dec) for p in m._idh_plans.values())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"B={B} conv stage {ms:.3f} ms/step  {ms/B:.3f} ms/frame  {flops/ms/1e9:.1f} TFLOP/s  ({flops/1e9/B:.1f} GFLOP/frame)")
for m, name in ((cve, "cve"), (dec, "dec")):
    p = list(m._idh_plans.values())[0][0]
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): p.run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n
    print(f"  {name}: {t:.3f} ms  {p.flops/t/1e9:.1f} TFLOP/s  ops={len(p.ops)}")
