"""Time the fused occlusion MLP (binary_mlp_k) alone: python tools/perf_mlp.py [B] [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import networks as net
from implicit_depth_amd.mlp import occlusion_logits

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, W = 192, 256
m = net.BinaryMLPNetwork([64, 64, 128, 256])
syn.fill_state_dict(m, 3)
m.cuda()
feat = torch.randn(B, H, W, 64, device="cuda")
rd = syn.rendered_depth_planes(B, H, W, P).cuda()
for _ in range(3):
    occlusion_logits(m, feat, 0, 64, rd)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    occlusion_logits(m, feat, 0, 64, rd)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * H * W * (P * (128 * 128 + 128) + 64 * 128)  # executed: feature part of layer 1 once per pixel
alg = 2.0 * B * H * W * P * (65 * 128 + 128 * 128 + 128)   # what the reference computes per plane
print(f"B={B} P={P}: {ms:.3f} ms  executed {fl / ms / 1e9:.1f} TFLOP/s  algorithmic {alg / ms / 1e9:.1f} TFLOP/s")
