"""Time the fused MLP feature volume alone: python tools/perf_fv.py [B] [K] [D] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import cost_volume as cv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
H, W = 96, 128
dev = torch.device("cuda:0")
if os.environ.get("MLP_MATH"):
    cv.DEFAULT_MLP_MATH = os.environ["MLP_MATH"]  # (fp32 | f16x3)
d = {k: v.to(dev) for k, v in syn.cost_volume_inputs(B, K, 16, H, W, seed=0).items()}
d["min_depth"], d["max_depth"] = 0.25, 5.0
m = cv.FeatureVolumeManager(H, W, D, num_source_views=K).to(dev)
syn.fill_state_dict(m.mlp, 99, gain=1.4)
cur_n, src_n = cv.to_nhwc(d["cur_feats"]), cv.to_nhwc(d["src_feats"])
vol = torch.empty(B, H, W, D, device=dev)
st = {}


def step():
    m._run(cur_n.data_ptr(), src_n.data_ptr(), (B, K, 16, H, W), d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"], 0.25, 5.0,
           vol.data_ptr(), D, False, dev, scratch=st)


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * D * H * W * (16 * (K + 1) + 10 * K + 4) * 128 + 2.0 * B * D * H * W * (128 * 128 + 128)
print(f"B={B} K={K} D={D}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s algorithmic ({fl / ms / 1e9 / 157.3:.3f} of peak)")
