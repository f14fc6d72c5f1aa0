#!/usr/bin/env python
"""bench.py — frames/s of the MI355X-native cost-volume hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--batch B]

One "step" = one pass of the hot path over the GLOBAL batch of 32 synthetic frames (BASELINE.json
configs[3]), sharded by frame over the GPUs (strong scaling: total work fixed, 32/N frames per GPU).  Inputs are generated once and are resident in HBM before the
timed region.  For N>1 launch with torch.distributed.run (one rank per GPU, RCCL); the batch
dimension is sharded, there is no data-path collective, and the only message is an
all-gather of per-frame metric vectors after the timed region.

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline     — dominant kernel, algorithmic bytes(flops)/launch ÷ HIP-event-measured
                 average launch time vs the gfx950 peak
  cpu_baseline — the oracle (CPU restatement) timed on this host's cores on a bounded
                 sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="auto", help="auto | warp_match_dot | hot_path")
    ap.add_argument("--batch", type=int, default=32, help="GLOBAL frames per step (BASELINE.json configs[3]: batch_size=32), sharded over the GPUs")
    ap.add_argument("--views", type=int, default=0, help="source views K; 0 = 7 for --volume mlp (reference-native 8-frame tuple = 1 cur + 7 src), 8 for --volume dot (BASELINE.json literal)")
    ap.add_argument("--volume", default="mlp", choices=["mlp", "dot"], help="mlp = FeatureVolumeManager (every shipped BDModel config), dot = CostVolumeManager")
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--conv-math", default="fp32", choices=["fp32", "bf16x6", "f16x3"],
                    help="arithmetic of the 3x3 stride-1 convs: fp32 MFMA (default) or the fp32-equivalent split-precision kernels")
    ap.add_argument("--mlp-math", default="fp32", choices=["fp32", "f16x3"], help="arithmetic of the MLP kernels (feature volume)")
    ap.add_argument("--math", default=None, choices=["fp32", "bf16x6", "f16x3"],
                    help="shorthand: sets --conv-math, and --mlp-math f16x3 when f16x3")
    ap.add_argument("--no-split-line", action="store_true", help="skip the secondary split-precision measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.math is not None:
        args.conv_math = args.math
        args.mlp_math = "f16x3" if args.math == "f16x3" else "fp32"
    return args


# ------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------
class WarpMatchDot:
    """BASELINE.json configs[1]: fused warp+match HIP kernel, 512x384 image -> 96x128 matching
    map, K source views, D planes, random-init N(0,1) matching features (NHWC, resident)."""

    name = "warp_match_dot"
    dominant_kernel = "cv_dot_quad_k"
    bound = "hbm"

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
        self.kernel_ms = []

    def config(self):
        return {"workload": f"{self.name}: fused plane-sweep warp+match, {self.W * 4}x{self.H * 4} image, matching map {self.W}x{self.H}, "
                            f"K={self.K} source views, D={self.D} planes, C=16, fp32",
                "per_gpu_batch": self.B, "source_views": self.K, "depth_planes": self.D}

    def step(self, ev=None):
        p, L = self._lib.ptr, self.L
        if ev is not None:
            ev[0].record()
        rc = L.idh_cost_volume_dot_fwd(p(self.cur), p(self.src), p(self.Ks), p(self.E), p(self.invK), 0.25, 5.0,
                                       self.B, self.K, self.C, self.H, self.W, self.D, p(self.cost), 0, p(self.lowest),
                                       p(self.planes), self._lib.stream_ptr())
        if ev is not None:
            ev[1].record()
        self._lib.check(rc, "idh_cost_volume_dot_fwd")

    def algorithmic_bytes_per_launch(self):
        # compulsory traffic per frame = every input read once + every output
        # written once = 4*[C*N + K*C*N + D*N + N] + 64*(2K+1) bytes; one launch processes B frames.
        N = self.H * self.W
        per_frame = 4 * (self.C * N + self.K * self.C * N + self.D * N + N) + 64 * (2 * self.K + 1)
        return per_frame * self.B

    def metrics(self):
        # per-frame vector that is all-gathered (stand-in for the reference's per-frame
        # metric dict, test_bd.py:288-339): mean cost, mean arg-max depth
        return torch.stack([self.cost.mean((1, 2, 3)), self.lowest.mean((1, 2))], 1)

    def cpu_baseline(self, seconds):
        from oracle import cost_volume as ocv

        i = self.host_inputs
        one = {k: (v[:1] if v.shape[0] == self.B and v.ndim > 1 and k not in ("min_depth", "max_depth") else v) for k, v in i.items()}
        n, t0 = 0, time.perf_counter()
        ocv.FAST_GATHER = True  # time the restatement with torch's own grid_sample primitive
        with torch.inference_mode():
            while True:
                ocv.cost_volume_dot(one["cur_feats"], one["src_feats"], one["src_extrinsics"], one["src_Ks"], one["cur_invK"], 0.25, 5.0, self.D)
                n += 1
                if time.perf_counter() - t0 > seconds and n >= 2:
                    break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} frames of the same workload through oracle/cost_volume.py (torch CPU fp32), {dt:.1f} s"}


def make_workload(args, device, rank):
    name = args.workload
    if name == "auto":
        try:
            from implicit_depth_amd import pipeline  # noqa: F401

            name = "hot_path"
        except Exception:
            name = "warp_match_dot"
    if name == "warp_match_dot":
        return WarpMatchDot(args, device, rank)
    if name == "hot_path":
        from implicit_depth_amd.pipeline import HotPathWorkload

        return HotPathWorkload(args, device, rank)
    raise SystemExit(f"unknown workload {name}")


# ------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.views == 0:
        args.views = 7 if (args.volume == "mlp" and args.workload != "warp_match_dot") else 8
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # strong scaling: the global batch is fixed (32 ScanNet-shaped tuples) and sharded by frame
    from implicit_depth_amd.dist import shard_range

    global_batch = args.batch
    lo, hi = shard_range(global_batch, world, rank)
    args.batch = hi - lo  # frames of THIS rank
    counts = [shard_range(global_batch, world, r)[1] - shard_range(global_batch, world, r)[0] for r in range(world)]
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun: exercise RCCL even at N=1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm

    wl = make_workload(args, device, rank)

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
        k_ms = sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)
        t = torch.tensor([el], device=device, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), k_ms

    elapsed, kernel_ms = timed(wl)

    # Secondary line (same run, same inputs and weights): the hot path with the fp32-equivalent
    # split-precision kernels (f16x3 convs + MLPs).  `value` above stays the fp32-MFMA path.
    split = None
    if wl.name == "hot_path" and args.conv_math == "fp32" and args.mlp_math == "fp32" and not args.no_split_line:
        import copy

        a2 = copy.copy(args)
        a2.conv_math, a2.mlp_math = "f16x3", "f16x3"
        wl2 = make_workload(a2, device, rank)
        el2, _ = timed(wl2)
        ref_o, got_o = wl.out["pred_0"], wl2.out["pred_0"]
        split = {"math": "f16x3", "value": global_batch * args.steps / el2, "unit": "frames/s", "ms_per_step": el2 / args.steps * 1e3,
                 "max_abs_diff_vs_fp32_path_over_max_abs": float((ref_o - got_o).abs().max() / ref_o.abs().max()),
                 "note": "3x3 convs, feature-volume MLP and BinaryMLP on the f16 matrix cores: fp32 operands split into two power-of-two-scaled "
                         "f16 pieces, 3 products, fp32 accumulate; same goldens / tolerances as the fp32-MFMA path (tests/test_*split*_gpu.py)"}
        del wl2
        torch.cuda.empty_cache()

    # the path's only collective: all-gather of per-frame metric vectors (RCCL over xGMI)
    from implicit_depth_amd.dist import all_gather_metrics

    m = all_gather_metrics(wl.metrics().float().contiguous(), counts=counts)
    frames_total = global_batch * args.steps

    if rank == 0:
        if wl.bound == "hbm":
            rl_alg = wl.algorithmic_bytes_per_launch()
            achieved = rl_alg / (kernel_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "kernel": wl.dominant_kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": rl_alg}
        else:
            # dominant kernel = conv3x3_lds_k<2> (8-row LDS-staged 3x3 conv, ~1/3 of a step): replay
            # ONLY its launches between two HIP events on the launch stream; the same for all conv
            # launches of the step as a secondary figure
            (dom_ms, dom_n, dom_fl), (all_ms, all_n, all_fl) = wl.conv_only_ms()
            achieved = dom_fl / (dom_ms * 1e-3) / 1e12
            math = getattr(wl, "conv_math", "fp32")
            # split-precision convs execute 6 (bf16x6) / 3 (f16x3) 16-bit MFMA products per fp32-equivalent
            # MAC: their roofline is the dense 16-bit MFMA peak divided by that count
            peak = {"fp32": MFMA_F32_PEAK_TFLOPS, "bf16x6": MFMA_16BIT_PEAK_TFLOPS / 6, "f16x3": MFMA_16BIT_PEAK_TFLOPS / 3}[math]
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": None, "kernel": wl.dominant_kernel, "kernel_ms": dom_ms / dom_n, "launches_per_step": dom_n,
                    "kernel_ms_per_step": dom_ms, "algorithmic_flops_per_launch": dom_fl / dom_n,
                    "all_conv_kernels": {"achieved": all_fl / (all_ms * 1e-3) / 1e12, "frac": all_fl / (all_ms * 1e-3) / 1e12 / peak,
                                         "launches_per_step": all_n, "ms_per_step": all_ms, "algorithmic_flops_per_step": all_fl},
                    "step_ms_hip_events": kernel_ms}
            if math != "fp32":
                roof["peak_note"] = (f"fp32-equivalent flops; peak = {MFMA_16BIT_PEAK_TFLOPS:.0f} TFLOP/s dense 16-bit MFMA / "
                                     f"{6 if math == 'bf16x6' else 3} products per MAC; the fp32-MFMA peak is 157.3")
        # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside the
        # process, so the figure comes from the committed rocprofv3 --pmc passes of this same command
        # (profiles/pmc_traffic.json, tools/pmc_bench.sh) when the configuration matches, else null.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = f"{wl.name}/{getattr(wl, 'volume', 'dot')}/b{wl.B}" if wl.name == "hot_path" else f"{wl.name}/b{wl.B}"
            if getattr(wl, "conv_math", "fp32") != "fp32":
                key += "/" + wl.conv_math
            if key in pmc and (wl.name != "warp_match_dot" or (wl.K, wl.D) == (8, 64)):
                roof["traffic"] = pmc[key]["traffic_bytes_per_launch"]
                roof["traffic_source"] = "profiles/pmc_traffic.json"
        except Exception:
            pass
        out = {
            "metric": "frames/sec (BDModel.forward, 512x384, 64 planes, 8 views)",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x6": "f32 (3x3 convs: bf16x3-split operands, 6 products, f32 accumulate)",
                      "f16x3": "f32 (3x3 convs: scaled f16x2-split operands, 3 products, f32 accumulate)"}[getattr(wl, "conv_math", "fp32")],
            "data": "synthetic",
            "config": dict(wl.config(), global_batch=global_batch),
            "roofline": roof,
            "gathered_metric_rows": int(m.shape[0]),
        }
        if split is not None:
            out["split_precision"] = split
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
