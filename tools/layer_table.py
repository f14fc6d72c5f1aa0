"""Conv-layer table of the hot path's CVEncoder + UNet++ plan (shapes, fused sources, share of the 2*MAC flops), built on
the meta device — no GPU needed.  python tools/layer_table.py [B]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implicit_depth_amd import nhwc, networks as net

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nhwc.packed_weight = lambda conv: torch.empty(1, device="meta")
nhwc.packed_wino_weight = lambda conv: torch.empty(1, device="meta")
nhwc.packed_wino4_weight = lambda conv: torch.empty(1, device="meta")
enc_ch = [24, 48, 64, 160, 256]
cve = net.CVEncoder(64, enc_ch[1:], [64, 128, 256, 384])
dec = net.BDDecoderPP(enc_ch[:1] + cve.num_ch_enc)
p = nhwc.Plan("meta")
_bias = torch.nn.Parameter.detach
cost = p.buffer(B, 96, 128, 64)
shapes = [(B, c, 192 >> i, 256 >> i) for i, c in enumerate(enc_ch)]
outs, _ = nhwc.build_cv_encoder(p, cve, cost, shapes[1:])
f0 = p.buffer(B, 192, 256, 24)
nhwc.build_decoder(p, dec, [f0] + outs)
rows = collections.OrderedDict()
tot = 0
for op in p.ops:
    if op.kind != nhwc.OP_CONV:
        continue
    s0, s1 = op.src[0], op.src[1]
    fl = 2 * op.N * op.Ho * op.Wo * op.Cout * (s0.Cin * s0.ks * s0.ks + (s1.Cin * s1.ks * s1.ks if s1.ks else 0))
    key = (s0.Cin, s0.ks, s0.stride, (s1.Cin, s1.ks, s1.stride) if s1.ks else None, op.Cout, op.Ho, op.Wo, bool(op.res), op.tile_m, op.tile_n, op.split_k)
    r = rows.setdefault(key, [0, 0])
    r[0] += 1; r[1] += fl; tot += fl
print(f"B={B}: {sum(r[0] for r in rows.values())} convs, {tot / 1e12:.3f} TFLOP")
for k, (n, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    cin, ks, st, s1, cout, Ho, Wo, res, tm, tn, sk = k
    print(f"{n:3d} x  {cin:4d}->{cout:4d} k{ks}s{st} @{Ho}x{Wo}" + (f" + {s1[0]}ch k{s1[1]}s{s1[2]}" if s1 else "") + (" +res" if res else "") +
          f"   tile {tm}/{tn} split {sk}   {100 * fl / tot:5.1f} %")
