#!/usr/bin/env python
"""Time the REFERENCE's own modules on this host's CPU cores (SURVEY.md 8(d) "Timing method").

Build container only: imports /root/reference with the stub recipe of tests/golden/gen_golden.py (absent third-party packages
replaced by empty modules; the timm / antialiased_cnns backbones by the seeded stand-ins of implicit_depth_amd.synthetic — they are
outside the hot path and only `BDModel.forward` touches them).  Never ships to the GPU box (it reads /root/reference); the numbers it
writes to profiles/cpu_reference.json are the baseline quoted beside bench.py's `cpu_baseline` (the oracle port timed on the GPU box).

    python tools/time_reference_cpu.py [--repeats 5]

Per component: torch.set_num_threads(nproc), fp32, inference_mode, one untimed warm-up + `repeats` timed runs -> best and median.
Components (512x384 input, 96x128 matching map, synthetic SURVEY 8(d) inputs):
  CostVolumeManager (modules/cost_volume.py:221-358) K=8 D=64, K=7 D=64, K=7 D=96; FeatureVolumeManager (:437-706) K=7 D=64;
  CVEncoder, BDDecoderPP (modules/networks.py:20-215); BDModel.forward (experiment_modules/bd_model.py:175-311) with
  mlp_feature_volume K=7 and simple_cost_volume K=8 (8 query planes), starting at the stand-in backbones' outputs.
"""
import argparse
import contextlib
import io
import json
import os
import platform
import statistics
import sys
import time

os.environ["PYTORCH_JIT"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def bench(fn, repeats):
    fn()  # warm-up (allocator, thread pool, oneDNN primitive caches)
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return {"best_s": min(ts), "median_s": statistics.median(ts), "runs": repeats}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0, help="0 = os.cpu_count()")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_reference.json"))
    args = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (build container only)")
    import gen_golden as gg

    gg.import_reference()
    import implicit_depth_amd.synthetic as syn
    import antialiased_cnns
    import timm

    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        gg._stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    sys.modules["kornia"].filters.sobel = None
    timm.create_model = lambda *a, **k: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **k: syn.StubResnetStem())
    torch.nn.Module.save_hyperparameters = lambda self, *a, **k: None
    from experiment_modules.bd_model import BDModel
    from modules.cost_volume import CostVolumeManager, FeatureVolumeManager
    from modules.networks import BDDecoderPP, CVEncoder
    from options import Options

    nthreads = args.threads or os.cpu_count()
    torch.set_num_threads(nthreads)
    torch.set_grad_enabled(False)
    Hi, Wi, H, W, C = 384, 512, 96, 128, 16
    res = {}
    with torch.inference_mode():
        for K, D in ((8, 64), (7, 64), (7, 96)):
            inp = syn.cost_volume_inputs(1, K, C, H, W, 0)
            m = CostVolumeManager(H, W, D)
            res[f"CostVolumeManager_k{K}_d{D}"] = bench(lambda: m(**inp), args.repeats)
            print(f"CostVolumeManager K={K} D={D}: {res[f'CostVolumeManager_k{K}_d{D}']}", flush=True)
        K, D = 7, 64
        inp = syn.cost_volume_inputs(1, K, C, H, W, 0)
        fv = FeatureVolumeManager(H, W, D, num_source_views=K)
        syn.fill_state_dict(fv.mlp, seed=99, gain=1.4)
        res["FeatureVolumeManager_k7_d64"] = bench(lambda: fv(**inp), max(3, args.repeats // 2))
        print(f"FeatureVolumeManager K=7 D=64: {res['FeatureVolumeManager_k7_d64']}", flush=True)
        enc_ch = [24, 48, 64, 160, 256]
        pyr = list(syn.encoder_pyramid(1, Hi, Wi, seed=0))
        cve = CVEncoder(D, enc_ch[1:], [64, 128, 256, 384])
        syn.fill_state_dict(cve, seed=100)
        cvol = syn.randn((1, D, H, W), 5, "cv")
        res["CVEncoder_d64"] = bench(lambda: cve(cvol, pyr[1:]), args.repeats)
        print(f"CVEncoder: {res['CVEncoder_d64']}", flush=True)
        dec = BDDecoderPP(enc_ch[:1] + cve.num_ch_enc)
        syn.fill_state_dict(dec, seed=101)
        dec_in = [pyr[0]] + cve(cvol, pyr[1:])
        res["BDDecoderPP"] = bench(lambda: dec(dec_in), args.repeats)
        print(f"BDDecoderPP: {res['BDDecoderPP']}", flush=True)
        for name, fvt, K in (("BDModel_forward_mlp_k7_d64", "mlp_feature_volume", 7), ("BDModel_forward_dot_k8_d64", "simple_cost_volume", 8)):
            o = Options()
            o.image_width, o.image_height = Wi, Hi
            o.matching_num_depth_bins = 64
            o.feature_volume_type = fvt
            o.model_num_views = K + 1
            o.binary_loss_positive_weight = 1.0
            o.bd_edge_regularision = False
            o.use_prior = False
            with contextlib.redirect_stdout(io.StringIO()):
                model = BDModel(o)
            model.eval()
            syn.fill_state_dict(model, seed=30, gain=1.0)
            cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=31, P=8)
            # the in-scope path: stand-in backbone OUTPUTS are resident (as in bench.py): layer1 maps -> head, encoder pyramid
            layer1 = syn.layer1_maps(1, K, H, W, seed=78)
            pyr_m = list(syn.encoder_pyramid(1, Hi, Wi, seed=73))

            def head_feats(*a, _m=model, _l1=layer1, **k):
                f = torch.cat([_m.matching_model.net[5:](x) for x in _l1[0].split(1, dim=0)], 0)[None]
                return f[:, 0], f[:, 1:].contiguous()

            model.compute_matching_feats = head_feats
            model.encoder.forward = lambda x, _p=pyr_m: _p
            res[name] = bench(lambda: model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=False), max(3, args.repeats // 2))
            res[name]["frames_per_s_best"] = 1.0 / res[name]["best_s"]
            res[name]["frames_per_s_median"] = 1.0 / res[name]["median_s"]
            print(f"{name}: {res[name]}", flush=True)
    out = {"host": {"cpu_model": cpu_model(), "logical_cpus": os.cpu_count(), "torch_threads": nthreads, "torch": torch.__version__,
                    "platform": platform.platform()},
           "what": "the reference's own modules (/root/reference), imported with the stub recipe of tests/golden/gen_golden.py; fp32, inference_mode, "
                   "1 warm-up + N timed runs; 512x384 input, 96x128 matching map; BDModel.forward from the stand-in backbones' outputs (the in-scope path)",
           "timings": res}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
