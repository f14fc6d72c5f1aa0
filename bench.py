#!/usr/bin/env python
"""bench.py — frames/s of the MI355X-native cost-volume hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B]

One "step" = one pass of the hot path over one batch of 32 synthetic frames PER GPU (BASELINE.json configs[3]: batch_size = 32; the
reference's multi-GPU mode is Lightning DDP, train.py:124,135, where batch_size is per process) - "scaling": "weak", the path shards over
independent frames with no data-path collective.  `--scaling strong` shards ONE global batch of --batch frames over the GPUs instead
(32/N frames per GPU: the configuration rounds 1-5 quoted; a 1-GPU run is the same workload either way).
Inputs are generated once and are resident in HBM before the timed region.  `--gpus N` (N > 1) without a launcher starts its own
ranks (python -m torch.distributed.run, one process per GPU, RCCL, 127.0.0.1 rendezvous) and forwards rank 0's line; under a launcher
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) it is one of the ranks.  There is no data-path collective; the only message
is an all-gather of per-frame metric vectors after the timed region - at N = 1 through a 1-rank RCCL group.

Workloads
  hot_path        (default) matching backbone's layer1 map -> encoder head -> fused volume -> CVEncoder ->
                  UNet++ decoder -> occlusion MLP x8 query planes; 512x384, D=64, 8-frame tuples
  warp_match_dot  BASELINE.json configs[1]: the fused warp+match kernel alone (K=8, D=64)
  temporal        BASELINE.json configs[4]: one sequence per GPU, B=1, D=96, the previous frame's prediction
                  carried as the prior (inference/inference.py:139-157); a step = one frame
  fused_forward   the reference's call shape (test_bd.py:196-212) from raw 512x384 images through dropin.fused_forward,
                  stand-in backbones inside the timed region
  module_swap     the same call through dropin.convert + the reference's own forward sequence (module by module)

Prints ONE JSON line on rank 0 (see the driver contract) with extra objects:
  roofline     — dominant kernel, algorithmic flops (bytes) per launch ÷ HIP-event-measured average launch
                 time vs the gfx950 peak
  warp_match   — the north-star kernel on its own (K=8, D=64, same per-GPU batch): achieved algorithmic GB/s
                 and its fraction of the 8 TB/s HBM roofline (rank 0)
  temporal     — configs[4] frames/s on one GPU (rank 0, N=1 only)
  cpu_baseline — the oracle (CPU restatement) timed on this host's cores on a bounded sample of the same
                 workload (rank 0, N=1 only)
  parity       — frames of the TIMED output against the same frames run alone (the timed plan's kernel mix and buffer
                 aliasing exist at no other batch size); bar 1e-4 of scale
  extra        — fused_forward / module_swap rates at the same batch (N=1 only)
  dist_backend, ranks, devices, allgather_us, launcher — the process group behind the barriers and the metric all-gather
stdout carries exactly this one line (native libraries' banners go to stderr).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak (no xf32/TF32 on gfx950)
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # bf16 / f16 MFMA dense peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="hot_path", choices=["hot_path", "warp_match_dot", "temporal", "fused_forward", "module_swap"])
    ap.add_argument("--batch", type=int, default=32, help="frames per step and GPU (BASELINE.json configs[3]: batch_size=32); with --scaling strong: the GLOBAL batch, sharded over the GPUs")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU runs --batch frames, the job's batch grows with N (DDP semantics of the reference, batch_size per process); "
                         "strong: --batch frames in total, sharded by frame (ragged shards allowed)")
    ap.add_argument("--views", type=int, default=0, help="source views K; 0 = 7 for --volume mlp (reference-native 8-frame tuple = 1 cur + 7 src), 8 for --volume dot (BASELINE.json literal)")
    ap.add_argument("--volume", default="mlp", choices=["mlp", "dot"], help="mlp = FeatureVolumeManager (every shipped BDModel config), dot = CostVolumeManager")
    ap.add_argument("--planes", type=int, default=0, help="depth planes D; 0 = 64 (96 for --workload temporal)")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--sequences", type=int, default=1, help="--workload temporal: independent sequences per GPU in one batch, each carrying its own prior")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="--workload temporal, one sequence: F consecutive frames share the volume / conv launches of a step; the occlusion MLP and its "
                         "carried prior stay frame by frame (HotPath.forward(frame_chain=...)); same outputs as F single-frame steps")
    ap.add_argument("--no-head", action="store_true", help="start at finished matching features (round-1 workload) instead of the layer1 map")
    ap.add_argument("--conv-math", default="fp32", choices=["fp32", "f16x3"],
                    help="arithmetic of the 3x3 stride-1 convs: fp32 MFMA (default) or the fp32-equivalent split-precision kernels")
    ap.add_argument("--mlp-math", default="fp32", choices=["fp32", "f16x3"], help="arithmetic of the MLP kernels (feature volume)")
    ap.add_argument("--math", default=None, choices=["fp32", "f16x3"],
                    help="shorthand: sets --conv-math, and --mlp-math f16x3 when f16x3")
    ap.add_argument("--split-line", action="store_true",
                    help="also time the opt-in split-precision (f16x3) kernels on the same inputs -> the `split_precision` object (frozen code path, "
                         "never `value`; off by default since round 6)")
    ap.add_argument("--no-split-line", action="store_true", help=argparse.SUPPRESS)  # round <= 5 spelling of the default
    ap.add_argument("--no-process-group", action="store_true",
                    help="--gpus 1 without a launcher: do NOT create the 1-rank RCCL group (the default creates it so that the N = 1 record "
                         "exercises ncclAllGather too)")
    ap.add_argument("--no-parity", action="store_true", help="skip the `parity` object (frames of the timed batch re-run one at a time)")
    ap.add_argument("--no-extras", action="store_true", help="skip the warp_match / temporal objects")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl (= RCCL over xGMI; one rank per GPU) or gloo (host-side collectives: lets several ranks share "
                         "ONE device with --ranks-on-device, which RCCL refuses — the way the N > 1 code path is exercised on a 1-GPU box)")
    ap.add_argument("--ranks-on-device", type=int, default=None, metavar="D",
                    help="put every rank on cuda:D instead of cuda:LOCAL_RANK (test rig for --dist-backend gloo; never for measurements)")
    ap.add_argument("--rank-report", default=None, metavar="DIR",
                    help="every rank writes DIR/rank<r>.json (its shard, input checksum, device) — evidence for the N > 1 tests")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args(argv)
    if args.math is not None:
        args.conv_math = args.math
        args.mlp_math = "f16x3" if args.math == "f16x3" else "fp32"
    if args.views == 0:
        args.views = 7 if (args.volume == "mlp" and args.workload != "warp_match_dot") else 8
    if args.planes == 0:
        args.planes = 96 if args.workload == "temporal" else 64
    return args


def shard_counts(global_batch: int, world: int):
    """Frames per rank for a global batch sharded by frame (ragged when world does not divide it)."""
    from implicit_depth_amd.dist import shard_range

    return [shard_range(global_batch, world, r)[1] - shard_range(global_batch, world, r)[0] for r in range(world)]


# ------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------
class WarpMatchDot:
    """BASELINE.json configs[1]: fused warp+match HIP kernel, 512x384 image -> 96x128 matching
    map, K source views, D planes, random-init N(0,1) matching features (NHWC, resident)."""

    name = "warp_match_dot"
    bound = "hbm"
    scaling = "strong"

    def __init__(self, args, device, rank):
        import implicit_depth_amd.synthetic as syn
        from implicit_depth_amd import _lib
        from implicit_depth_amd.cost_volume import to_nhwc

        self.L = _lib.lib()
        self._lib = _lib
        self.B, self.K, self.D = args.batch, args.views, args.planes
        self.H, self.W, self.C = args.height // 4, args.width // 4, 16
        inp = syn.cost_volume_inputs(self.B, self.K, self.C, self.H, self.W, seed=rank)
        self.host_inputs = inp
        d = {k: v.to(device) for k, v in inp.items()}
        self.cur = to_nhwc(d["cur_feats"])
        self.src = to_nhwc(d["src_feats"])
        self.Ks, self.E, self.invK = d["src_Ks"].contiguous(), d["src_extrinsics"].contiguous(), d["cur_invK"].contiguous()
        self.cost = torch.empty(self.B, self.D, self.H, self.W, device=device)
        self.lowest = torch.empty(self.B, self.H, self.W, device=device)
        self.planes = torch.empty(self.D, device=device)
        self.dominant_kernel = self.L.idh_cost_volume_dot_kernel_name(self.B, self.K, self.H, self.W, self.D).decode()
        # the drop-in (CostVolumeManager / HotPath) passes the optional arg-max scratch of idh_volume_opts: so does the bench
        from implicit_depth_amd.cost_volume import volume_opts

        self.opts, self._keep = volume_opts(self.B, self.K, self.C, self.H, self.W, self.D, dot_scratch_device=device)

    def input_checksum(self):
        """float64 sum of this rank's synthetic inputs (seed = rank: ranks must differ)"""
        t = self.host_l1 if getattr(self, "host_l1", None) is not None else self.host_inputs["src_feats"]
        return float(t.double().sum()) + float(self.host_inputs["src_extrinsics"].double().sum())

    def config(self):
        return {"workload": f"{self.name}: fused plane-sweep warp+match, {self.W * 4}x{self.H * 4} image, matching map {self.W}x{self.H}, "
                            f"K={self.K} source views, D={self.D} planes, C=16, fp32",
                "per_gpu_batch": self.B, "source_views": self.K, "depth_planes": self.D}

    def metric(self):
        return f"frames/sec (fused warp+match kernel, {self.W * 4}x{self.H * 4}, {self.D} planes, {self.K} views)"

    def step(self, ev=None):
        p, L = self._lib.ptr, self.L
        if ev is not None:
            ev[0].record()
        rc = L.idh_cost_volume_dot_ex_fwd(p(self.cur), p(self.src), p(self.Ks), p(self.E), p(self.invK), 0.25, 5.0,
                                          self.B, self.K, self.C, self.H, self.W, self.D, p(self.cost), 0, p(self.lowest),
                                          p(self.planes), self.opts, self._lib.stream_ptr())
        if ev is not None:
            ev[1].record()
        self._lib.check(rc, "idh_cost_volume_dot_ex_fwd")

    def frames_per_step(self):
        return self.B

    def algorithmic_bytes_per_launch(self):
        # compulsory traffic per frame = every input read once + every output
        # written once = 4*[C*N + K*C*N + D*N + N] + 64*(2K+1) bytes; one launch processes B frames.
        N = self.H * self.W
        per_frame = 4 * (self.C * N + self.K * self.C * N + self.D * N + N) + 64 * (2 * self.K + 1)
        return per_frame * self.B

    def kernel_roofline(self, iters=30):
        """HIP-event timing of `iters` back-to-back launches on the launch stream -> the `warp_match` object."""
        for _ in range(5):
            self.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            self.step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        alg = self.algorithmic_bytes_per_launch()
        gbs = alg / (ms * 1e-3) / 1e9
        out = {"kernel": self.dominant_kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "kernel_ms": ms, "algorithmic_bytes_per_launch": alg, "frames_per_launch": self.B, "source_views": self.K, "depth_planes": self.D,
               "frames_per_s": self.B / (ms * 1e-3), "traffic": None,
               "note": "achieved = compulsory bytes (every input read once, every output written once) / time of the launch (cv_dot_win_k + its arg-max pass cv_argmax_k); bound by "
                       "the on-chip gather + VALU work of D*K*N*4 bilinear taps, not by HBM; BASELINE.json's >= 0.5 of the HBM peak is unattainable under this accounting: the packed-fp32 VALU floor of the sampling arithmetic alone is ~11 us/frame = 0.12 (DESIGN.md 4.1)"}
        pmc = _pmc_traffic(f"warp_match_dot/b{self.B}") if (self.K, self.D) == (8, 64) else None
        if pmc is not None and pmc.get("kernel") == self.dominant_kernel:
            out["traffic"], out["traffic_source"] = pmc["traffic_bytes_per_launch"], "profiles/pmc_traffic.json"
        return out

    def metrics(self):
        # per-frame vector that is all-gathered (stand-in for the reference's per-frame
        # metric dict, test_bd.py:288-339): mean cost, mean arg-max depth
        return torch.stack([self.cost.mean((1, 2, 3)), self.lowest.mean((1, 2))], 1)

    def cpu_baseline(self, seconds):
        from oracle import cost_volume as ocv

        i = self.host_inputs
        one = {k: (v[:1] if v.shape[0] == self.B and v.ndim > 1 and k not in ("min_depth", "max_depth") else v) for k, v in i.items()}
        ocv.FAST_GATHER = True  # time the restatement with torch's own grid_sample primitive
        run = lambda: ocv.cost_volume_dot(one["cur_feats"], one["src_feats"], one["src_extrinsics"], one["src_Ks"], one["cur_invK"], 0.25, 5.0, self.D)
        with torch.inference_mode():
            run()  # untimed warm-up (thread pool, allocator, first-call costs)
            n, t0 = 0, time.perf_counter()
            while True:
                run()
                n += 1
                if time.perf_counter() - t0 > seconds and n >= 4:
                    break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} frames of the same workload through oracle/cost_volume.py (torch CPU fp32) after one untimed warm-up frame, {dt:.1f} s",
                "reference_cpu": _reference_cpu(f"CostVolumeManager_k{self.K}_d{self.D}")}


class HotPathWorkload:
    """The in-scope part of BDModel.forward on synthetic ScanNet-shaped tuples — 512x384 image, matching map
    128x96, K source views, D planes, P=8 query planes — starting at the matching backbone's layer1 map
    (B*(K+1) images, 64 channels at 1/4 resolution).  The third-party image encoder and ResNet18 stem are replaced by
    resident synthetic feature maps of the right shape (they are outside the hot path)."""

    name = "hot_path"
    bound = "mfma"
    scaling = "strong"
    use_prior = False
    P = 8

    def __init__(self, args, device, rank):
        import implicit_depth_amd.synthetic as syn
        from implicit_depth_amd import networks as net
        from implicit_depth_amd.cost_volume import CostVolumeManager, FeatureVolumeManager
        from implicit_depth_amd.pipeline import HotPath

        self.args = args
        self.B, self.K, self.D = args.batch, args.views, args.planes
        self.Hi, self.Wi = args.height, args.width
        self.H, self.W, self.C = args.height // 4, args.width // 4, 16
        self.head = not getattr(args, "no_head", False)
        enc_ch = [24, 48, 64, 160, 256]
        self.volume = getattr(args, "volume", "dot")
        if self.volume == "mlp":
            cv = FeatureVolumeManager(self.H, self.W, self.D, num_source_views=self.K)
            syn.fill_state_dict(cv.mlp, seed=99, gain=1.4)
        else:
            cv = CostVolumeManager(self.H, self.W, self.D)
        cve = net.CVEncoder(self.D, enc_ch[1:], [64, 128, 256, 384])
        dec = net.BDDecoderPP(enc_ch[:1] + cve.num_ch_enc)
        mlp = net.BinaryMLPNetwork(dec.num_ch_dec, mlp_size=128, use_prior=self.use_prior)
        mm = net.ResnetMatchingEncoder([torch.nn.Identity() for _ in range(5)], self.C) if self.head else None
        for i, m in enumerate((cve, dec, mlp) + ((mm,) if mm is not None else ())):
            syn.fill_state_dict(m, seed=100 + i)
        self.conv_math = getattr(args, "conv_math", "fp32")
        self.mlp_math = getattr(args, "mlp_math", "fp32")
        if self.volume == "mlp":
            cv.mlp_math = self.mlp_math
        mlp.mlp_math = self.mlp_math
        self.model = HotPath(cv, cve, dec, mlp, conv_math=self.conv_math, matching_model=mm).to(device)
        inp = syn.cost_volume_inputs(self.B, self.K, self.C, self.H, self.W, seed=rank)
        self.host_inputs = inp
        self.host_pyr = syn.encoder_pyramid(self.B, self.Hi, self.Wi, seed=rank)
        self.host_rd = syn.rendered_depth_planes(self.B, self.Hi // 2, self.Wi // 2, self.P)
        self.d = {k: v.to(device) for k, v in inp.items()}
        self.pyr = [t.to(device) for t in self.host_pyr]
        self.rd = self.host_rd.to(device)
        self.l1 = self.host_l1 = None
        if self.head:
            self.host_l1 = syn.layer1_maps(self.B, self.K, self.H, self.W, seed=rank)
            self.l1 = self.host_l1.to(device)
        self.out = None

    def input_checksum(self):
        """float64 sum of this rank's synthetic inputs (seed = rank: ranks must differ)"""
        t = self.host_l1 if getattr(self, "host_l1", None) is not None else self.host_inputs["src_feats"]
        return float(t.double().sum()) + float(self.host_inputs["src_extrinsics"].double().sum())

    def config(self):
        vol = "fused MLP feature volume (FeatureVolumeManager, implicit_depth.yaml)" if self.volume == "mlp" else "fused warp+match (dot, CostVolumeManager)"
        start = (f"layer1 map of the matching backbone ({self.K + 1} images/frame, 64ch @ {self.W}x{self.H}, NCHW) -> matching-encoder head "
                 "(1x1 conv, InstanceNorm, LeakyReLU, 3x3 conv, InstanceNorm; NHWC hand-off)") if self.head else "matching feats"
        return {"workload": f"{self.name}: {start} -> {vol} -> CVEncoder -> BDDecoderPP (UNet++) -> occlusion MLP x{self.P} planes; "
                            f"{self.Wi}x{self.Hi} image, matching map {self.W}x{self.H}, K={self.K} source views, D={self.D} planes, fp32; "
                            "image encoder / ResNet18 stem (third-party) replaced by resident synthetic feature maps",
                "per_gpu_batch": self.B, "source_views": self.K, "depth_planes": self.D, "query_planes": self.P, "volume": self.volume,
                "matching_head_in_timed_region": self.head, "conv_math": self.conv_math, "mlp_math": self.mlp_math}

    def metric(self):
        return (f"frames/sec (BDModel.forward hot path, {self.Wi}x{self.Hi}, {self.D} planes, {self.K + 1}-frame tuple = 1 current + {self.K} "
                f"source views; third-party backbones excluded)")

    def frames_per_step(self):
        return self.B

    def _forward(self, frames=None, **kw):
        """``frames``: a slice of the batch (the `parity` object re-runs single frames through the same modules)"""
        d, pyr, rd, l1 = self.d, self.pyr, self.rd, self.l1
        if frames is not None:
            d = {k: (v[frames] if v.shape[0] == self.B and v.dim() > 1 and k not in ("min_depth", "max_depth") else v) for k, v in d.items()}
            pyr, rd, l1 = [t[frames] for t in pyr], rd[frames], (l1[frames] if l1 is not None else None)
        if self.head:
            return self.model(None, None, pyr, d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"], rendered_depth=rd,
                              matching_layer1=l1, return_mask=True, **kw)  # return_mask=True: the reference's eval call, test_bd.py:200-208
        return self.model(d["cur_feats"], d["src_feats"], pyr, d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                          rendered_depth=rd, return_mask=True, **kw)

    def parity(self):
        """Numerical evidence for the TIMED plan (its kernel mix — F(4x4) / F(2x2) / direct by tile counts — and its buffer aliasing exist only at
        this batch size): frames 0, B/2 and B-1 of the timed output against the same frames run ONE AT A TIME through the same modules (a
        one-frame plan: no F(4x4) layer, no recycled buffer, other split-K factors; the one-frame plan is what tests/test_bdmodel_gpu.py pins to the
        reference's full-size goldens).  Scale-relative max error of the logits (bar: 1e-4, BASELINE.json), mismatch rates of the discrete
        outputs (arg-max depth, overall mask)."""
        from implicit_depth_amd import nhwc

        big = {k: v.clone() for k, v in self.out.items() if torch.is_tensor(v)}
        ops = [op for op in next(iter(self.model._plans.values()))["plan"].ops if op.kind == nhwc.OP_CONV]
        plan = next(iter(self.model._plans.values()))["plan"]
        res = {"frames": sorted({0, self.B // 2, self.B - 1}), "tolerance": 1e-4,
               "timed_plan": {"convs": len(ops), "wino4": sum(op.tile_m == nhwc.TILE_WINO4 for op in ops), "wino2": sum(op.tile_m == nhwc.TILE_WINO for op in ops),
                              "recycled_buffers": int(getattr(plan, "recycled", 0))}}
        rel, low, msk = [], [], []
        for b in res["frames"]:
            one = self._forward(frames=slice(b, b + 1))
            ref = one["pred_0"]
            rel.append(float((big["pred_0"][b:b + 1] - ref).abs().max() / ref.abs().max()))
            low.append(float(((big["lowest_cost_bhw"][b:b + 1] - one["lowest_cost_bhw"]).abs() > 1e-5).float().mean()))
            if one.get("overall_mask_bhw") is not None:
                msk.append(float((big["overall_mask_bhw"][b:b + 1] != one["overall_mask_bhw"]).float().mean()))
        res.update(frame0_vs_b1_rel=rel[0], worst_frame_vs_b1_rel=max(rel), lowest_mismatch_rate=max(low), mask_mismatch_rate=max(msk) if msk else None,
                   ok=bool(max(rel) < 1e-4 and max(low) < 5e-3 and (not msk or max(msk) < 2e-3)))
        return res

    def step(self, ev=None):
        if ev is not None:
            ev[0].record()
        self.out = self._forward()
        if ev is not None:
            ev[1].record()

    # roofline of the dominant kernel -------------------------------------------------------
    dominant_kernel = None  # set by conv_only_ms(): the conv kernel family with the most time per step, as rocprofv3 names it

    def _replay_ms(self, ops, iters=10, batches=3):
        """ms per pass of `ops` replayed alone between HIP events on the launch stream: median of
        `batches` batches of `iters` passes (a single batch right after the timed loop occasionally
        catches a clock / power-state transient: 0.74 vs 0.79 of peak for the same kernel)."""
        import ctypes as C

        from implicit_depth_amd import _lib, nhwc

        arr = (nhwc.Op * len(ops))(*ops)
        L = _lib.lib()
        for _ in range(3):
            _lib.check(L.idh_run_ops(C.cast(arr, C.c_void_p), len(ops), _lib.stream_ptr()), "idh_run_ops")
        times = []
        for _ in range(batches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                _lib.check(L.idh_run_ops(C.cast(arr, C.c_void_p), len(ops), _lib.stream_ptr()), "idh_run_ops")
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / iters)
        return sorted(times)[len(times) // 2]

    @staticmethod
    def _conv_flops(op):
        """algorithmic flops of a conv op: 2 * MAC of the direct convolution, no padding"""
        return sum(2 * op.N * op.Ho * op.Wo * op.Cout * s.Cin * s.ks * s.ks for s in op.src if s.in_)

    @staticmethod
    def _conv_executed_flops(op):
        """flops the kernel's MFMAs execute (= SQ_INSTS_MFMA x 2048 of the launch).  Winograd F(2x2,3x3): 16 multiplies per
        2x2 output tile, input channel (padded to the 8-channel K step) and output channel over whole 32 x 8 pixel tiles; the
        fused 1x1 source runs in 32-channel steps.  Direct kernels: the algorithmic count (channel padding of the odd
        layers aside)."""
        from implicit_depth_amd import nhwc

        if op.tile_m == nhwc.TILE_WINO4:
            # Winograd F(4x4,3x3): 36 multiplies per 4x4 output tile over whole 32 x 8 pixel tiles (= SQ_INSTS_MFMA x 2048: PMC 42.47 M
            # for 192->64 @192x256 x 32, profiles/r04/pmc_wino4_vs_wino2_192to64.txt)
            pix4 = op.N * (-(-op.Ho // 8) * 8) * (-(-op.Wo // 32) * 32)
            fl4 = 2 * (pix4 // 16) * 36 * (-(-op.src[0].Cin // 16) * 16) * op.Cout
            if op.src[1].in_:  # the fused 1x1 projection: pixel-domain MFMAs in 16-channel chunks
                fl4 += 2 * pix4 * (-(-op.src[1].Cin // 16) * 16) * op.Cout
            return fl4
        if op.tile_m != nhwc.TILE_WINO:
            return HotPathWorkload._conv_flops(op)
        pix = op.N * (-(-op.Ho // 8) * 8) * (-(-op.Wo // 32) * 32)
        fl = 2 * (pix // 4) * 16 * (-(-op.src[0].Cin // 8) * 8) * op.Cout
        if op.src[1].in_:
            fl += 2 * pix * (-(-op.src[1].Cin // 32) * 32) * op.Cout
        return fl

    @staticmethod
    def _kernel_family(op, grouped=False):
        """rocprofv3's name of the kernel a conv op of the plan launches, from its tile codes (idh_op.tile_m / tile_n);
        `grouped`: the op shares a persistent grid with other Winograd convs of its dependency level"""
        from implicit_depth_amd import nhwc

        if op.tile_m == nhwc.TILE_WINO4:
            # conv3x3_wino4_k<SRC2, RES>: fused 1x1 projection of a second tensor / residual added in the epilogue (never both)
            return f"conv3x3_wino4_k<{'true' if op.src[1].in_ else 'false'}, {'true' if (op.res and not op.src[1].in_) else 'false'}>"
        if op.tile_m == nhwc.TILE_WINO:
            return f"conv3x3_wino_{'group_' if grouped else ''}k<4, 2, 8, {'true' if op.src[1].in_ else 'false'}>"
        if op.tile_m in (10, 11):
            return "conv3x3_split_k<4, 2, 1, *>"
        if op.tile_m in (8, 9):
            rw, nj = (2 if op.tile_m == 8 else 1), (op.tile_n or 4)
            if op.tile_m == 9:
                return "conv3x3_lds_k<1, false, *, false, *> + conv3x3_lds_group_k<1, false, *> + level_k<*>"
            s2 = bool(op.src[1].in_) and op.src[1].ks == 3
            return f"conv3x3_lds_k<{rw}, false, {nj}, {'true' if op.src[0].norm else 'false'}, {'true' if s2 else 'false'}>"
        return "conv_mfma_k<*, *>"

    def conv_only_ms(self, iters=10):
        """HIP-event timing of (a) the launches of the dominant conv kernel — the kernel family (by the plan's tile codes) whose
        launches take the most time per step, each family replayed alone on the launch stream — and (b) every conv op of
        the step.  Returns two (ms_per_step, launches, algorithmic flops, executed flops)."""
        from implicit_depth_amd import nhwc

        ent = next(iter(self.model._plans.values()))
        p = ent["plan"]
        convs = [op for op in p.ops if op.kind == nhwc.OP_CONV]
        # idh_run_ops launches a run of consecutive Winograd convs that share a level's group id (and the kind of second source)
        # as one conv3x3_wino_group_k grid, at most 6 at a time (csrc/conv.hip: idh_run_ops, csrc/conv_wino.hip: kWinoMaxGroup)
        grouped, launches_of = {}, {}
        i = 0
        while i < len(p.ops):
            op = p.ops[i]
            j = i + 1
            if op.kind == nhwc.OP_CONV and op.tile_m == nhwc.TILE_WINO and op.group:
                while (j < len(p.ops) and j - i < 6 and p.ops[j].kind == nhwc.OP_CONV and p.ops[j].tile_m == nhwc.TILE_WINO and p.ops[j].group == op.group
                       and bool(p.ops[j].src[1].in_) == bool(op.src[1].in_)):
                    j += 1
            for k in range(i, j):
                grouped[id(p.ops[k])] = j - i > 1
            i = j
        fams = {}
        for op in convs:
            fams.setdefault(self._kernel_family(op, grouped.get(id(op), False)), []).append(op)
        # the two families with the most algorithmic flops are timed; the slower one is the dominant kernel
        cand = sorted(fams, key=lambda k: -sum(self._conv_flops(o) for o in fams[k]))[:2]
        timed = {k: self._replay_ms(fams[k], iters) for k in cand}
        self.dominant_kernel = max(timed, key=timed.get)
        dom = fams[self.dominant_kernel]
        def launches(ops):  # kernel launches of a replayed op list (idh_count_launches: same decisions as idh_run_ops)
            import ctypes as C

            from implicit_depth_amd import _lib

            arr = (nhwc.Op * len(ops))(*ops)
            return int(_lib.lib().idh_count_launches(C.cast(arr, C.c_void_p), len(ops)))

        dom_res = (timed[self.dominant_kernel], launches(dom), sum(self._conv_flops(o) for o in dom), sum(self._conv_executed_flops(o) for o in dom))
        all_res = (self._replay_ms(convs, iters), launches(convs), sum(self._conv_flops(o) for o in convs), sum(self._conv_executed_flops(o) for o in convs))
        return dom_res, all_res

    def fv_mlp_roofline(self, iters=10):
        """The feature-volume kernel (`fv_mlp_k<K>`, second largest of the step) replayed alone between HIP events on the launch
        stream, priced on the flops its MFMAs EXECUTE: per 16-voxel tile and plane 32 K (warped features) + 8 * 15 (metadata: 4
        four-slot blocks minus the structurally-zero last slot) + 256 (layer 2) v_mfma_f32_16x16x4 of 2048 flops, + 64 per pixel
        tile for the plane-independent pre-activation (csrc/feature_volume.hip; = SQ_INSTS_MFMA of the launch,
        profiles/r03/pmc_mfma_util_hot_path_mlp_b32.json).  The algorithmic count of the reference's MLP (2 * MAC of
        Linear(C (K+1) + 10 K + 4, 128), Linear(128, 128), Linear(128, 1), cost_volume.py:405-423) is reported beside it: the kernel
        folds the current-view terms into a per-pixel pre-activation and the pose terms into a per-frame bias."""
        if self.volume != "mlp" or self.K > 8 or self.C != 16:
            return None
        m = self.model
        ent = next(iter(m._plans.values()))
        d = self.d
        call = lambda: m.cost_volume.fused_into(ent["cv_in"], ent["state"], ent["feats"], d["src_extrinsics"], d["src_poses"], d["src_Ks"],
                                                d["cur_invK"], m.min_depth, m.max_depth, False)
        for _ in range(2):
            call()
        times = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / iters)
        ms = sorted(times)[1]
        vox = self.B * self.H * self.W * self.D
        # MFMAs issued per 16-voxel tile and plane: 32 per source view, 96 for the per-voxel metadata (12 slots per lane quarter; 104 at K = 8, where the
        # plane depth needs a fourth block), 256 for layer 2, + 64 per pixel tile for the plane-independent part (csrc/feature_volume.hip)
        executed = (vox // 16) * (32 * self.K + (96 if self.K < 8 else 104) + 256 + 64 / self.D) * 2048
        n_in = self.C * (self.K + 1) + 10 * self.K + 4
        algorithmic = 2 * vox * (n_in * 128 + 128 * 128 + 128)
        busy = busy_src = None
        try:  # the newest round's committed PMC pass (profiles/rNN/pmc_mfma_util_hot_path_mlp_b32.json)
            import glob

            busy_src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_mfma_util_hot_path_mlp_b32.json")))[-1]
            rows = json.load(open(busy_src))
            busy = next(r["mfma_util"] for r in rows if r["kernel"] == f"fv_mlp_k<{self.K}>") if (self.B, self.K, self.D) == (32, 7, 64) else None
            busy_src = os.path.relpath(busy_src, ROOT)
        except Exception:
            pass
        return {"kernel": f"fv_mlp_k<{self.K}>", "ms": ms, "bound": "mfma", "achieved": executed / (ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": executed / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "executed_flops_per_launch": executed,
                "algorithmic_flops_per_launch": algorithmic, "achieved_algorithmic": algorithmic / (ms * 1e-3) / 1e12,
                "mfma_busy": busy, "mfma_busy_source": f"{busy_src} (SQ_VALU_MFMA_BUSY_CYCLES pass)" if busy else None,
                "frames_per_launch": self.B}

    def metrics(self):
        """Per-frame metric rows (B, 120): PlaneEvaluator IoU / IoU+ / IoU- for 5 thresholds x 8 query
        planes against a synthetic ground-truth depth — the shape of the dict test_bd.py:288-339
        builds per frame, computed on the GPU (csrc/metrics.hip) and then all-gathered."""
        import implicit_depth_amd.synthetic as syn
        from implicit_depth_amd.metrics import PlaneEvaluator, metric_rows

        o = self.out
        B, P, H, W = o["pred_0"].shape
        gt = (1.0 + 3.5 * torch.sigmoid(syn.randn((B, 1, H, W), 7, "bench_gt"))).to(o["pred_0"].device)
        rows, self.metric_keys = metric_rows(PlaneEvaluator().compute_batch_scores(self.rd, gt, torch.sigmoid(o["pred_0"])))
        return rows

    def cpu_baseline(self, seconds):
        """oracle (torch CPU fp32 restatement) of the same path, one frame at a time."""
        from oracle import cost_volume as ocv
        from oracle import networks as onet

        i = self.host_inputs
        sd = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        w_cve, w_dec, w_mlp = sd(self.model.cost_volume_net), sd(self.model.depth_decoder), sd(self.model.binary_mlp)
        w_mm = sd(self.model.matching_model) if self.head else None
        ocv.FAST_GATHER = True  # time the restatement with torch's own grid_sample primitive

        def frame():
            cur_f, src_f = i["cur_feats"][:1], i["src_feats"][:1]
            if self.head:
                f = onet.matching_head(self.host_l1[0], w_mm)[None]
                cur_f, src_f = f[:, 0], f[:, 1:]
            if self.volume == "mlp":
                w_fv = {k: v.detach().cpu() for k, v in self.model.cost_volume.mlp.state_dict().items()}
                cvol = ocv.feature_volume(cur_f, src_f, i["src_extrinsics"][:1], i["src_poses"][:1], i["src_Ks"][:1],
                                          i["cur_invK"][:1], 0.25, 5.0, self.D, w_fv)[0]
            else:
                cvol, _, _ = ocv.cost_volume_dot(cur_f, src_f, i["src_extrinsics"][:1], i["src_Ks"][:1], i["cur_invK"][:1], 0.25, 5.0, self.D)
            pyr = [t[:1] for t in self.host_pyr]
            enc = onet.cv_encoder(cvol, pyr[1:], w_cve)
            dec = onet.unetpp_decoder([pyr[0]] + enc, w_dec, depth_head=False)
            prior = -torch.ones_like(self.host_rd[:1]) if self.use_prior else None
            onet.occlusion_logits(dec["feature_s0_b1hw"], self.host_rd[:1], w_mlp, prior)

        with torch.inference_mode():
            frame()  # untimed warm-up frame: thread pool, allocator and oneDNN primitive caches (the figure drifted 0.18 -> 0.12 without it)
            n, t0 = 0, time.perf_counter()
            while True:
                frame()
                n += 1
                if time.perf_counter() - t0 > seconds and n >= 4:
                    break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} frame(s) of the same workload through oracle/ (torch CPU fp32 restatement, grid_sample gather) after one untimed warm-up frame, {dt:.1f} s",
                "reference_cpu": _reference_cpu("BDModel_forward_mlp_k7_d64" if self.volume == "mlp" else "BDModel_forward_dot_k8_d64")}


class TemporalWorkload(HotPathWorkload):
    """BASELINE.json configs[4] as the reference runs it (inference/inference.py:139-157): 8-frame tuple, D=96 planes,
    prior-enabled occlusion MLP, ONE query plane at 2 m (the `plane_2.0` asset), the previous frame's sigmoid(pred_0) +
    cam_T_world carried as the prior.  Frames of a sequence are serially dependent, so a step = one frame of each of the
    S independent sequences a GPU runs side by side in one batch (``--sequences``, default 1 = the reference's loop; every
    sequence has its own camera track and its own carried prior), and GPUs run disjoint sets of sequences (weak scaling;
    SURVEY.md §8e)."""

    name = "temporal"
    scaling = "weak"
    use_prior = True
    P = 1

    def __init__(self, args, device, rank):
        a = copy.copy(args)
        self.S = max(1, int(getattr(args, "sequences", 1)))
        self.F = max(1, int(getattr(args, "frames_in_flight", 1)))  # consecutive frames of ONE sequence per step
        if self.F > 1 and self.S > 1:
            raise SystemExit("--frames-in-flight needs --sequences 1")
        a.batch = self.S * self.F  # S sequences per GPU, one frame of each per step (or F consecutive frames of one sequence)
        super().__init__(a, device, rank)
        import implicit_depth_amd.synthetic as syn

        S = self.S * self.F
        self.rd = torch.full((S, 1, self.Hi // 2, self.Wi // 2), 2.0, device=device)
        self.host_rd = self.rd.cpu()
        K0 = syn.intrinsics(self.Wi // 2, self.Hi // 2).float()
        self.K0, self.invK0 = K0[None].expand(S, 4, 4).contiguous().to(device), torch.linalg.inv(K0)[None].expand(S, 4, 4).contiguous().to(device)
        # sequence s drifts (5 + s/2) cm / frame along x and s mm / frame along y: the prior is re-projected with a real,
        # sequence-specific motion every frame
        self.poses = []
        for t in range(64):
            T = torch.eye(4).repeat(self.S, 1, 1)
            for q in range(self.S):
                T[q, 0, 3] = (0.05 + 0.005 * q) * t
                T[q, 1, 3] = 0.001 * q * t
            self.poses.append((T.to(device), torch.linalg.inv(T).to(device)))  # world_T_cam, cam_T_world
        self.t = 0
        self.prev = None

    def metric(self):
        return (f"frames/sec (temporal BDModel.forward hot path, {self.Wi}x{self.Hi}, {self.D} planes, {self.K + 1}-frame tuple, "
                f"prior carried frame to frame, {self.S} sequence(s) per GPU" + (f", {self.F} consecutive frames per step" if self.F > 1 else "") + ")")

    def step(self, ev=None):
        if ev is not None:
            ev[0].record()
        if self.F > 1:  # F consecutive frames of the one sequence: everything up to the decoder in one batch, the MLP / prior chain frame by frame
            ps = [self.poses[(self.t + f) % len(self.poses)] for f in range(self.F)]
            chain = {"world_T_cam_b44": torch.cat([p[0] for p in ps]), "cam_T_world_b44": torch.cat([p[1] for p in ps]), "K_s0_b44": self.K0, "invK_s0_b44": self.invK0,
                     "prior_prediction": self.prev[0] if self.prev is not None else None, "prior_cam_T_world": self.prev[1] if self.prev is not None else None}
            self.out = self._forward(frame_chain=chain)
            self.prev = (torch.sigmoid(self.out["pred_0"][-1:]), ps[-1][1])
            self.t += self.F
            if ev is not None:
                ev[1].record()
            return
        wTc, cTw = self.poses[self.t % len(self.poses)]
        prior_inputs = None
        if self.prev is not None:
            prior_inputs = {"prior_prediction": self.prev[0], "prior_cam_T_world": self.prev[1], "world_T_cam_b44": wTc,
                            "K_s0_b44": self.K0, "invK_s0_b44": self.invK0}
        self.out = self._forward(prior_inputs=prior_inputs)
        self.prev = (torch.sigmoid(self.out["pred_0"]), cTw)  # sigmoid_custom(x, 1.0), inference.py:154
        self.t += 1
        if ev is not None:
            ev[1].record()

    def metrics(self):
        o = self.out["pred_0"]
        return torch.stack([o.mean((1, 2, 3)), torch.sigmoid(o).mean((1, 2, 3))], 1)


class _RunOpts:
    """the fields of the reference's options object that BDModel.forward reads (options.py; implicit_depth.yaml)"""
    matching_scale = 1
    min_matching_depth = 0.25
    max_matching_depth = 5.0
    use_prior = False
    bd_edge_regularision = False
    cv_encoder_type = "multi_scale_encoder"


def _standin_bdmodel(K, H, W, D, volume, golden_weights=False):
    """An object with the attribute tree of the reference's BDModel (bd_model.py:41-141): stand-in backbones (syn.StubImageEncoder /
    syn.StubResnetStem - the third-party timm / antialiased_cnns networks are not in this image) and the hot-path modules as
    `dropin.convert` leaves them."""
    import implicit_depth_amd.synthetic as syn
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.cost_volume import CostVolumeManager, FeatureVolumeManager

    m = torch.nn.Module()
    m.encoder = syn.StubImageEncoder()
    m.cost_volume = FeatureVolumeManager(H, W, D, num_source_views=K) if volume == "mlp" else CostVolumeManager(H, W, D)
    stem = syn.StubResnetStem()
    m.matching_model = net.ResnetMatchingEncoder([stem.conv1, stem.bn1, stem.relu, stem.maxpool, stem.layer1], 16)
    m.cost_volume_net = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    m.depth_decoder = net.BDDecoderPP([24] + m.cost_volume_net.num_ch_enc)
    m.binary_mlp = net.BinaryMLPNetwork(m.depth_decoder.num_ch_dec, mlp_size=128, use_prior=False)
    m.run_opts = _RunOpts()
    m.thresholder = None
    syn.fill_state_dict(m, seed=30)  # (name-keyed: the tensors tests/golden/gen_golden.py gave the reference's BDModel)
    if volume == "mlp" and not golden_weights:
        syn.fill_state_dict(m.cost_volume.mlp, seed=99, gain=1.4)  # (the hot_path workload's feature-volume weights: O(1) volume values)
    return m


def _reference_shaped_forward(m, cur_data, src_data, return_mask=True):
    """The call sequence of the reference's BDModel.forward at test time (bd_model.py:175-311, run_mlp_val :412-442), module by module,
    on a model whose hot-path attributes were swapped by `dropin.convert` - what a reference user gets from the one-line module swap of
    INTEGRATION.md 1: NCHW tensors between the modules, torch.cat per query plane, 8 BinaryMLPNetwork calls."""
    o = m.run_opts
    ms = o.matching_scale
    cur_image, src_image = cur_data["image_b3hw"], src_data["image_b3hw"]
    src_K, cur_invK = src_data[f"K_s{ms}_b44"], cur_data[f"invK_s{ms}_b44"]
    src_cam_T_cur_cam = src_data["cam_T_world_b44"] @ cur_data["world_T_cam_b44"].unsqueeze(1)   # :196-204
    cur_cam_T_src_cam = cur_data["cam_T_world_b44"].unsqueeze(1) @ src_data["world_T_cam_b44"]
    cur_feats = list(m.encoder(cur_image))                                                      # :218
    frames = torch.cat([cur_image.unsqueeze(1), src_image], 1)                                  # compute_matching_feats :143-173 (batched form)
    feats = m.matching_model(frames.flatten(0, 1)).unflatten(0, frames.shape[:2])
    mc, msrc = feats[:, 0], feats[:, 1:].contiguous()
    min_depth = torch.tensor(o.min_matching_depth).type_as(src_K).view(1, 1, 1, 1)
    max_depth = torch.tensor(o.max_matching_depth).type_as(src_K).view(1, 1, 1, 1)
    cost_volume, lowest_cost, _, overall_mask = m.cost_volume(cur_feats=mc, src_feats=msrc, src_extrinsics=src_cam_T_cur_cam, src_poses=cur_cam_T_src_cam,
                                                              src_Ks=src_K, cur_invK=cur_invK, min_depth=min_depth, max_depth=max_depth,
                                                              return_mask=return_mask)                   # :235-245
    cv_feats = m.cost_volume_net(cost_volume, cur_feats[ms:])                                   # :253-258
    feature_outputs = m.depth_decoder(cur_feats[:ms] + cv_feats)                                # :261
    outputs = None
    rendered_depth = cur_data["rendered_depth"]
    features = feature_outputs["feature_s0_b1hw"]
    for idx in range(rendered_depth.shape[1]):                                                  # :293-304
        model_inputs = [torch.cat((rendered_depth[:, idx:idx + 1], features), 1).permute(0, 2, 3, 1)]   # run_mlp_val :415-436
        cur = {k: v.permute(0, 3, 1, 2) for k, v in m.binary_mlp(model_inputs, max_scale_only=True).items()}
        outputs = cur if outputs is None else {k: torch.cat((v, cur[k]), 1) for k, v in outputs.items()}
    outputs["lowest_cost_bhw"], outputs["overall_mask_bhw"] = lowest_cost, overall_mask
    return outputs


class FusedForwardWorkload:
    """What a reference user calls and the reference times (test_bd.py:196-212): `model("test", cur_data, src_data, return_mask=True)` on
    raw 512x384 image tuples - with `model.forward = dropin.fused_forward(model)` installed.  The stand-in backbones run
    as torch modules INSIDE the timed region (they are not the third-party networks: the figure shows the cost of the call shape - dict handling,
    pose products, the stem / encoder hand-off - not of EfficientNetV2-S)."""

    name = "fused_forward"
    bound = "mfma"
    scaling = "strong"
    P = 8
    mode = "fused"

    def __init__(self, args, device, rank):
        import implicit_depth_amd.synthetic as syn
        from implicit_depth_amd.dropin import convert, fused_forward

        self.B, self.K, self.D = args.batch, args.views, args.planes
        self.Hi, self.Wi = args.height, args.width
        self.volume = args.volume
        self.model = _standin_bdmodel(self.K, self.Hi // 4, self.Wi // 4, self.D, self.volume).to(device).eval()
        cur, src = syn.frame_tuple(self.B, self.K, self.Hi, self.Wi, seed=rank, P=self.P)
        self.cur = {k: v.to(device) for k, v in cur.items()}
        self.src = {k: v.to(device) for k, v in src.items()}
        self._checksum = float(cur["image_b3hw"].double().sum()) + float(src["cam_T_world_b44"].double().sum())
        if self.mode == "fused":
            self.fwd = fused_forward(self.model)
        else:
            convert(self.model)  # (idempotent: the attributes already are drop-ins)
            self.fwd = lambda phase, c, s_, return_mask=True: _reference_shaped_forward(self.model, c, s_, return_mask)
        self.out = None
        self.dominant_kernel = None

    def input_checksum(self):
        return self._checksum

    def frames_per_step(self):
        return self.B

    def config(self):
        what = ("dropin.fused_forward(model): BDModel.forward's signature, one fused HotPath pass behind the stand-in backbones" if self.mode == "fused" else
                "dropin.convert(model) + the reference's own forward sequence: module by module, NCHW between modules, 8 BinaryMLPNetwork calls")
        return {"workload": f"{self.name}: {what}; raw {self.Wi}x{self.Hi} images, K={self.K} source views, D={self.D} planes, {self.P} query planes, fp32; "
                            "stand-in image encoder / ResNet stem (torch) inside the timed region",
                "per_gpu_batch": self.B, "source_views": self.K, "depth_planes": self.D, "query_planes": self.P, "volume": self.volume}

    def metric(self):
        return (f"frames/sec (BDModel.forward call shape via {'dropin.fused_forward' if self.mode == 'fused' else 'dropin.convert (module swap)'}, "
                f"{self.Wi}x{self.Hi}, {self.D} planes, {self.K + 1}-frame tuple, stand-in backbones in the timed region)")

    def step(self, ev=None):
        if ev is not None:
            ev[0].record()
        self.out = self.fwd("test", self.cur, self.src, return_mask=True)
        if ev is not None:
            ev[1].record()

    def metrics(self):
        o = self.out["pred_0"]
        return torch.stack([o.mean((1, 2, 3)), torch.sigmoid(o).mean((1, 2, 3))], 1)

    def cpu_baseline(self, seconds):
        return None


class ModuleSwapWorkload(FusedForwardWorkload):
    name = "module_swap"
    mode = "swap"


WORKLOADS = {"warp_match_dot": WarpMatchDot, "hot_path": HotPathWorkload, "temporal": TemporalWorkload, "fused_forward": FusedForwardWorkload,
             "module_swap": ModuleSwapWorkload}


def _pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, tools/pmc_bench.sh):
    PMC counters cannot be read from inside the process."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def _reference_cpu(key):
    """The REFERENCE's own modules timed on the build container's CPU cores (tools/time_reference_cpu.py -> profiles/cpu_reference.json,
    committed): quoted beside the port figure, which is what can be timed on the GPU box (/root/reference does not exist there)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference.json")))
        t = d["timings"][key]
        return {"frames_per_s_best": 1.0 / t["best_s"], "frames_per_s_median": 1.0 / t["median_s"], "runs": t["runs"], "what": key,
                "cores": d["host"]["torch_threads"], "cpu_model": d["host"]["cpu_model"], "torch": d["host"]["torch"],
                "source": "profiles/cpu_reference.json (the reference's own code, build container; not measured in this run)"}
    except Exception:
        return None


def _median(xs):
    s = sorted(xs)
    n = len(s)
    return 0.0 if n == 0 else (s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2]))


# ------------------------------------------------------------------------------------------
def _free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: start the N ranks ourselves — one process per GPU under
    torch.distributed.run, the reference's process-per-GPU model (train.py:124,135) — and pass rank 0's JSON line through as the
    only stdout line; everything else the ranks print goes to stderr.  Refuses (rc != 0) only when the box has fewer than N GPUs
    (unless --ranks-on-device puts every rank on one device, the gloo test rig)."""
    import subprocess

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if args.ranks_on_device is None and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box exposes {have} GPU(s) to torch; one process per GPU needs {args.gpus}")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or args.gpus) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench.py] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=ROOT)
    line = None
    for ln in proc.stdout:
        t = ln.strip()
        if line is None and t.startswith("{") and t.endswith("}") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0 or line is None:
        raise SystemExit(f"bench.py self-launch of {args.gpus} ranks failed (rc {rc}, {'no JSON line' if line is None else 'JSON line seen'})")
    rec = json.loads(line)
    rec["launcher"] = "self (python -m torch.distributed.run, started by bench.py)"
    _emit(json.dumps(rec))


_JSON_FD = None


def _claim_stdout():
    """stdout carries exactly ONE line: the JSON record.  Native libraries write there too (RCCL prints its version banner to the C stdout when
    the first communicator is created), so fd 1 is pointed at stderr for the life of the process and the record goes to a private duplicate of
    the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(line: str):
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def main():
    args = parse()
    _claim_stdout()
    under_launcher = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.gpus > 1 and not under_launcher:
        return _self_launch(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.ranks_on_device is not None and args.dist_backend == "nccl" and world > 1:
        raise SystemExit("--ranks-on-device with more than one rank needs --dist-backend gloo (RCCL refuses duplicate devices)")
    dev_index = local_rank if args.ranks_on_device is None else args.ranks_on_device
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: cuda:{dev_index} does not exist ({torch.cuda.device_count()} GPU(s) visible)")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # weak scaling (default): --batch frames per rank (every rank its own 32 ScanNet-shaped tuples, seed = rank); strong: ONE global batch sharded by frame
    counts = [args.batch] * world if args.scaling == "weak" else shard_counts(args.batch, world)
    args.batch = counts[rank]  # frames of THIS rank
    # A process group always exists (a 1-rank RCCL group at N = 1 without a launcher: the N = 1 record then exercises the path's one
    # collective, ncclAllGather, too) unless --no-process-group asks for the bare single process.
    use_dist = under_launcher or not args.no_process_group
    dist_error = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not under_launcher:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        try:
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)  # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)  # host-side collectives (tensors are staged through the CPU below)
        except Exception as e:  # a lone process can still be measured; N > 1 cannot
            if world > 1:
                raise
            use_dist, dist_error = False, f"{type(e).__name__}: {e}"
    host_coll = use_dist and args.dist_backend == "gloo"

    wl = WORKLOADS[args.workload](args, device, rank)
    if wl.name == "temporal":  # one set of sequences per GPU: weak by construction (frames of a sequence are serially dependent)
        counts = [wl.frames_per_step()] * world
    else:
        wl.scaling = args.scaling

    def barrier():
        if use_dist:
            dist.barrier()

    def timed(w):
        """W warmup steps, then exactly K steps between barrier + synchronize pairs; MAX over ranks."""
        with torch.inference_mode():
            for _ in range(args.warmup):
                w.step()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            t0 = time.perf_counter()
            for i in range(args.steps):
                w.step(evs[i])
            torch.cuda.synchronize()
            barrier()
            el = time.perf_counter() - t0
            torch.cuda.synchronize()
        per_step = [a.elapsed_time(b) for a, b in evs]
        t = torch.tensor([el], device="cpu" if host_coll else device, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), per_step

    elapsed, per_step = timed(wl)
    kernel_ms = sum(per_step) / max(len(per_step), 1)
    parity = None
    if wl.name == "hot_path" and not args.no_parity and rank == 0:
        with torch.inference_mode():
            parity = wl.parity()
    frames_per_step_total = sum(counts)

    # Secondary line (same run, same inputs and weights): the hot path with the fp32-equivalent
    # split-precision kernels (f16x3 convs + MLPs).  `value` above stays the fp32-MFMA path.
    split = None
    if wl.name == "hot_path" and args.conv_math == "fp32" and args.mlp_math == "fp32" and args.split_line:
        a2 = copy.copy(args)
        a2.conv_math, a2.mlp_math = "f16x3", "f16x3"
        wl2 = HotPathWorkload(a2, device, rank)
        el2, _ = timed(wl2)
        ref_o, got_o = wl.out["pred_0"], wl2.out["pred_0"]
        split = {"math": "f16x3", "value": frames_per_step_total * args.steps / el2, "unit": "frames/s", "ms_per_step": el2 / args.steps * 1e3,
                 "max_abs_diff_vs_fp32_path_over_max_abs": float((ref_o - got_o).abs().max() / ref_o.abs().max()),
                 "note": "NOT the headline: 3x3 convs, feature-volume MLP and BinaryMLP on the f16 matrix cores with fp32 operands split into two "
                         "power-of-two-scaled f16 pieces, 3 products, fp32 accumulate (narrower operand arithmetic than the reference's fp32); same "
                         "goldens / tolerances as the fp32-MFMA path (tests/test_*split*_gpu.py)"}
        del wl2
        torch.cuda.empty_cache()

    # the path's only collective: all-gather of per-frame metric vectors (RCCL over xGMI)
    from implicit_depth_amd.dist import all_gather_metrics

    local_rows = wl.metrics().float().contiguous()
    send = local_rows.cpu() if host_coll else local_rows
    m = all_gather_metrics(send, counts=counts)
    # evidence of the collective: wall time of the metric all-gather (median of 20 after the first, which carries RCCL's lazy
    # communicator setup), the world size the process group itself reports, and one device record per rank gathered through it
    allgather_us = first_allgather_us = ranks_reported = devices = None
    if use_dist:
        def one_gather():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            all_gather_metrics(send, counts=counts)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6

        first_allgather_us = one_gather()
        allgather_us = _median([one_gather() for _ in range(20)])
        ranks_reported = dist.get_world_size()
        pr = torch.cuda.get_device_properties(device)
        mine = {"rank": rank, "device": str(device), "name": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None),
                "uuid": str(getattr(pr, "uuid", "")) or None, "frames": counts[rank], "pid": os.getpid()}
        devices = [None] * ranks_reported
        dist.all_gather_object(devices, mine)
    frames_total = frames_per_step_total * args.steps
    if args.rank_report:
        os.makedirs(args.rank_report, exist_ok=True)
        lo = sum(counts[:rank])
        with open(os.path.join(args.rank_report, f"rank{rank}.json"), "w") as f:
            json.dump({"rank": rank, "world": world, "device": str(device), "backend": args.dist_backend if use_dist else None,
                       "frames": [lo, lo + counts[rank]], "input_checksum": wl.input_checksum(), "local_metric_rows": int(local_rows.shape[0]),
                       "gathered_metric_rows": int(m.shape[0]),
                       # (metric rows hold NaN where a score is undefined, as the reference's do: compare with NaN == NaN)
                       "gathered_rows_match_local": bool(torch.equal(torch.nan_to_num(m[lo:lo + counts[rank]].cpu(), nan=-7.0),
                                                                      torch.nan_to_num(local_rows.cpu(), nan=-7.0)))}, f)

    if rank == 0:
        if not hasattr(wl, "conv_only_ms") and wl.bound != "hbm":
            # call-shape workloads (fused_forward / module_swap): the kernels are the hot path's; its line carries the roofline
            roof = {"bound": "mfma", "achieved": None, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                    "note": "same kernels as --workload hot_path, whose line prices the dominant one; this workload measures the call shape"}
        elif wl.bound == "hbm":
            rl_alg = wl.algorithmic_bytes_per_launch()
            achieved = rl_alg / (kernel_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "kernel": wl.dominant_kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": rl_alg}
        else:
            # dominant kernel = conv3x3_lds_k<2> (8-row LDS-staged 3x3 conv, ~2/3 of a step): replay
            # ONLY its launches between two HIP events on the launch stream; the same for all conv
            # launches of the step as a secondary figure
            (dom_ms, dom_n, dom_fl, dom_ex), (all_ms, all_n, all_fl, all_ex) = wl.conv_only_ms()
            math = getattr(wl, "conv_math", "fp32")
            # split-precision convs execute 3 f16 MFMA products per fp32-equivalent
            # MAC: their roofline is the dense 16-bit MFMA peak divided by that count
            peak = {"fp32": MFMA_F32_PEAK_TFLOPS, "f16x3": MFMA_16BIT_PEAK_TFLOPS / 3}[math]
            # `achieved` / `frac`: flops the kernel's MFMAs EXECUTE per second (<= peak by construction).  The Winograd kernel
            # needs 2.25x fewer multiplies than the direct convolution it replaces, so the algorithmic 2*MAC rate of
            # SURVEY.md 8(d) is reported beside it and may exceed the peak.
            achieved = dom_ex / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": None, "kernel": wl.dominant_kernel, "kernel_ms": dom_ms / dom_n, "launches_per_step": dom_n,
                    "kernel_ms_per_step": dom_ms, "executed_flops_per_launch": dom_ex / dom_n, "algorithmic_flops_per_launch": dom_fl / dom_n,
                    "achieved_algorithmic": dom_fl / (dom_ms * 1e-3) / 1e12, "frac_algorithmic": dom_fl / (dom_ms * 1e-3) / 1e12 / peak,
                    "all_conv_kernels": {"achieved": all_ex / (all_ms * 1e-3) / 1e12, "frac": all_ex / (all_ms * 1e-3) / 1e12 / peak,
                                         "achieved_algorithmic": all_fl / (all_ms * 1e-3) / 1e12, "launches_per_step": all_n, "ms_per_step": all_ms,
                                         "executed_flops_per_step": all_ex, "algorithmic_flops_per_step": all_fl},
                    "step_ms_hip_events": kernel_ms}
            if math != "fp32":
                roof["peak_note"] = (f"fp32-equivalent flops; peak = {MFMA_16BIT_PEAK_TFLOPS:.0f} TFLOP/s dense 16-bit MFMA / "
                                     "3 products per MAC; the fp32-MFMA peak is 157.3")
        key = f"{wl.name}/{getattr(wl, 'volume', 'dot')}/b{wl.B}" if wl.name != "warp_match_dot" else f"{wl.name}/b{wl.B}"
        if getattr(wl, "conv_math", "fp32") != "fp32":
            key += "/" + wl.conv_math
        pmc = _pmc_traffic(key)
        if pmc is not None and pmc.get("kernel") == wl.dominant_kernel and (wl.name != "warp_match_dot" or (wl.K, wl.D) == (8, 64)):
            roof["traffic"] = pmc["traffic_bytes_per_launch"]
            roof["traffic_source"] = "profiles/pmc_traffic.json"
        out = {
            "metric": wl.metric(),
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median_hip_events": _median(per_step),
            "ms_per_step_min_hip_events": min(per_step) if per_step else 0.0,
            "higher_is_better": True,
            "scaling": wl.scaling,
            "vs_baseline": None,
            "dtype": {"fp32": "f32",
                      "f16x3": "f32 (3x3 convs: scaled f16x2-split operands, 3 products, f32 accumulate)"}[getattr(wl, "conv_math", "fp32")],
            "data": "synthetic",
            "config": dict(wl.config(), global_batch=frames_per_step_total, parallelism=f"dp{world}: frames sharded over {world} rank(s), one process per GPU, weights replicated, no data-path collective"),
            "roofline": roof,
            "gathered_metric_rows": int(m.shape[0]),
            # the process group behind the barriers / MAX all-reduce / metric all-gather of this run ("nccl" = RCCL)
            "dist_backend": (args.dist_backend if use_dist else None),
            "ranks": ranks_reported,
            "devices": devices,
            "allgather_us": allgather_us,
            "allgather_first_call_us": first_allgather_us,
            "allgather_payload_bytes_per_rank": int(local_rows.numel() * 4),
            "launcher": ("torch.distributed.run" if under_launcher else "none (single process" + (", 1-rank process group)" if use_dist else ")")),
        }
        if dist_error is not None:
            out["dist_init_error"] = dist_error
        if parity is not None:
            out["parity"] = parity
        if args.ranks_on_device is not None:
            out["note_ranks_on_device"] = (f"all {world} ranks ran on cuda:{args.ranks_on_device} (code-path exercise of the N > 1 branch on a 1-GPU box; "
                                           "the rate is NOT a multi-GPU measurement)")
        if split is not None:
            out["split_precision"] = split
        if wl.name == "hot_path" and getattr(wl, "mlp_math", "fp32") == "fp32":
            with torch.inference_mode():
                fv = wl.fv_mlp_roofline()
            if fv is not None:
                out["fv_mlp"] = fv
        if wl.name == "hot_path" and not args.no_extras:
            # the north-star kernel on its own, same per-GPU batch (BASELINE.json configs[1]: K=8, D=64)
            a3 = copy.copy(args)
            a3.views, a3.planes = 8, 64
            with torch.inference_mode():
                out["warp_match"] = WarpMatchDot(a3, device, rank).kernel_roofline()
            torch.cuda.empty_cache()
            if world == 1:  # BASELINE.json configs[4]: the temporal loop, D=96, prior carried over 48 frames
                def temporal_run(S, F=1):
                    a4 = copy.copy(args)
                    a4.planes, a4.views, a4.volume, a4.sequences, a4.frames_in_flight = 96, 7, "mlp", S, F
                    tw = TemporalWorkload(a4, device, rank)
                    with torch.inference_mode():
                        for _ in range(8):
                            tw.step()
                        torch.cuda.synchronize()
                        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(48)]
                        t0 = time.perf_counter()
                        for e in evs:
                            tw.step(e)
                        torch.cuda.synchronize()
                        el = time.perf_counter() - t0
                    ts = [a.elapsed_time(b) for a, b in evs]
                    cfg = tw.config()["workload"]
                    del tw
                    torch.cuda.empty_cache()
                    return {"value": S * F * len(evs) / el, "unit": "frames/s", "sequences": S, "frames": S * F * len(evs), "ms_per_step_median_hip_events": _median(ts),
                            **({"frames_per_step": F, "ms_per_frame": _median(ts) / F} if F > 1 else {})}, cfg

                one, cfg = temporal_run(1)  # the reference's loop: one sequence, one frame at a time
                out["temporal"] = {"value": one["value"], "unit": "frames/s", "frames": one["frames"],
                                   "ms_per_frame_median_hip_events": one["ms_per_step_median_hip_events"], "config": cfg,
                                   # S independent sequences per GPU in one batch, each carrying its own prior (how the path shards
                                   # "by sequence"): frames/s of all S sequences together
                                   "sequences_per_gpu": {"1": one, "4": temporal_run(4)[0], "8": temporal_run(8)[0]},
                                   # ONE sequence, F consecutive frames per step: head, volume, CVEncoder and decoder do not depend on the previous
                                   # frame and run once for the F frames; the occlusion MLP with its carried prior stays frame by frame
                                   # (HotPath.forward(frame_chain=...): same outputs as F single-frame steps, tests/test_temporal_gpu.py).  Throughput of a
                                   # recorded scan (the reference's inference.py), at a latency of one step per frame
                                   "frames_in_flight": {"2": temporal_run(1, 2)[0], "4": temporal_run(1, 4)[0]}}
            if world == 1:
                # What a reference user calls (test_bd.py:196-212): model("test", cur_data, src_data, return_mask=True) from raw images, with the
                # stand-in backbones inside the timed region - through dropin.fused_forward and through the one-line module swap (dropin.convert +
                # the reference's own forward sequence).  Same weights (name-keyed seeds) and inputs on both: their logits must agree.
                def call_shape(cls):
                    w = cls(copy.copy(args), device, rank)
                    with torch.inference_mode():
                        for _ in range(3):
                            w.step()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(10):
                            w.step()
                        torch.cuda.synchronize()
                        el_ = (time.perf_counter() - t0) / 10
                    return w, {"value": w.B / el_, "unit": "frames/s", "ms_per_step": el_ * 1e3, "steps": 10, "per_gpu_batch": w.B,
                               "fraction_of_hot_path_rate": (w.B / el_) / (frames_total / elapsed), "config": w.config()["workload"]}

                wf, rf = call_shape(FusedForwardWorkload)
                pf = wf.out["pred_0"].clone()
                del wf
                torch.cuda.empty_cache()
                wm, rm = call_shape(ModuleSwapWorkload)
                rm["logits_max_abs_diff_vs_fused_forward_over_max_abs"] = float((wm.out["pred_0"] - pf).abs().max() / pf.abs().max())
                del wm, pf
                torch.cuda.empty_cache()
                out["extra"] = {"fused_forward": rf, "module_swap": rm}
            if args.volume == "mlp" and args.views != 8:
                # BASELINE.json's literal "8 source views" through the whole path (the headline is the reference-native 8-frame
                # tuple = 7 source views): same batch, D, head and query planes, K = 8
                a5 = copy.copy(args)
                a5.views = 8
                w8 = HotPathWorkload(a5, device, rank)
                with torch.inference_mode():
                    for _ in range(3):
                        w8.step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        w8.step()
                    torch.cuda.synchronize()
                    el8 = (time.perf_counter() - t0) / 10
                out["k8"] = {"value": frames_per_step_total / el8, "unit": "frames/s", "ms_per_step": el8 * 1e3, "source_views": 8,
                             "depth_planes": a5.planes, "per_gpu_batch": w8.B, "volume": "mlp", "steps": 10,
                             "note": "whole hot path with K = 8 source views (BASELINE.json's literal configuration); rank-0 rate x ranks"}
                del w8
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            cb = wl.cpu_baseline(args.cpu_seconds)
            if cb is not None:
                out["cpu_baseline"] = cb
        _emit(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
