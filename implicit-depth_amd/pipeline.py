"""The hot path of ``BDModel.forward`` / ``DepthModel.forward`` as one NHWC pipeline.

Covers reference experiment_modules/bd_model.py:231-311 (and depth_model.py:378-433) from
the matching features onwards:

    matching feats --(layout)--> fused warp+match (cost volume, NHWC out)
        --> CVEncoder --> UNet++ decoder --> per-pixel occlusion MLP over all query planes
                                         \\-> (DepthModel) 1x1 log-depth heads, exp

Everything between the NCHW inputs and the NCHW outputs stays channels-last in HBM; the only
layout conversions are the imports of the caller's NCHW tensors.  The module-level drop-ins
in cost_volume.py / networks.py do the same work one module at a time (with a conversion at
every module boundary) for callers that only swap attributes.

The image encoder (timm EfficientNetV2-S) and the ResNet18 stem of the matching encoder are
third-party code that is neither in the reference tree nor in scope: their
outputs are inputs of this pipeline.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _lib, nhwc
from .cost_volume import CostVolumeManager, ZeroCostVolumeManager
from .mlp import occlusion_logits


class HotPath(nn.Module):
    """Owns (or shares) the four hot-path modules of a BDModel / DepthModel."""

    def __init__(self, cost_volume: nn.Module, cost_volume_net: nn.Module, depth_decoder: nn.Module,
                 binary_mlp: Optional[nn.Module] = None, min_depth: float = 0.25, max_depth: float = 5.0,
                 conv_math: Optional[str] = None):
        super().__init__()
        self.conv_math = conv_math  # None = nhwc.DEFAULT_MATH ("fp32"); "bf16x6" / "f16x3": see nhwc.MATH_MODES
        self.cost_volume = cost_volume
        self.cost_volume_net = cost_volume_net
        self.depth_decoder = depth_decoder
        self.binary_mlp = binary_mlp
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.thresholder = None  # like BDModel.thresholder (bd_model.py:141): per-depth thresholds of the infer_depth search
        self._plans: Dict = {}

    # ------------------------------------------------------------------------------------
    def _plan(self, B, K, C, H, W, enc_shapes: Sequence[Sequence[int]], device):
        key = (B, K, C, H, W, tuple(tuple(s) for s in enc_shapes), str(device), self.conv_math,
               nhwc._param_key(self.cost_volume_net), nhwc._param_key(self.depth_decoder))
        ent = self._plans.get(key)
        if ent is not None:
            return ent
        self._plans.clear()
        D = self.cost_volume.num_depth_bins
        p = nhwc.Plan(device, math=self.conv_math)
        st = {"cur_n": torch.empty(B, H, W, C, device=device), "src_n": torch.empty(B, K, H, W, C, device=device),
              "lowest": None, "planes": torch.empty(D, device=device)}
        cv_in = p.buffer(B, H, W, D)
        v0 = p.buffer(B, enc_shapes[0][2], enc_shapes[0][3], enc_shapes[0][1])
        i_enc = [p.import_nchw(enc_shapes[0], v0)]
        outs, i_img = nhwc.build_cv_encoder(p, self.cost_volume_net, cv_in, enc_shapes[1:])
        i_enc += i_img
        final = nhwc.build_any_decoder(p, self.depth_decoder, [v0] + outs)
        ent = {"plan": p, "state": st, "cv_in": cv_in, "i_enc": i_enc, "final": final, "heads": {}}
        if getattr(self.depth_decoder, "depth_head", False):
            for i, v in final.items():
                ent["heads"][i] = p.head(v, self.depth_decoder.convs[f"output_{i}"][1], torch.empty(1, device=device))
        elif hasattr(self.depth_decoder, "out1"):  # SkipDecoderRegression
            for i, (hv, last) in nhwc.build_regression_heads(p, self.depth_decoder, final).items():
                ent["heads"][i] = p.head(hv, last, torch.empty(1, device=device))
        p.schedule()
        self._plans[key] = ent
        return ent

    # ------------------------------------------------------------------------------------
    def forward(self, matching_cur_feats: torch.Tensor, matching_src_feats: torch.Tensor, cur_feats: List[torch.Tensor],
                src_cam_T_cur_cam: torch.Tensor, cur_cam_T_src_cam: torch.Tensor, src_K: torch.Tensor, cur_invK: torch.Tensor,
                rendered_depth: Optional[torch.Tensor] = None, prior: Optional[torch.Tensor] = None,
                return_mask: bool = False, return_features: bool = False,
                prior_inputs: Optional[Dict[str, torch.Tensor]] = None, infer_depth: bool = False) -> Dict[str, torch.Tensor]:
        """``prior``: an already-warped prior channel (B,P,H/2,W/2), or ``prior_inputs`` = the
        reference's temporal inputs {"prior_prediction", "prior_cam_T_world", "world_T_cam_b44",
        "K_s0_b44", "invK_s0_b44"} (bd_model.py:420-431) to warp it here; with neither, a
        prior-enabled MLP sees the constant -1 (bd_model.py:433-434)."""
        _lib.require_cuda_f32(matching_cur_feats, matching_src_feats, src_cam_T_cur_cam, src_K, cur_invK, rendered_depth, prior, *cur_feats)
        B, K, C, H, W = matching_src_feats.shape
        dev = matching_cur_feats.device
        cur_feats = [f.contiguous() for f in cur_feats]
        ent = self._plan(B, K, C, H, W, [f.shape for f in cur_feats], dev)
        p, st = ent["plan"], ent["state"]
        L = _lib.lib()
        sp = _lib.stream_ptr()
        D = self.cost_volume.num_depth_bins
        out: Dict[str, torch.Tensor] = {}

        # 1. cost volume, written NHWC straight into the CVEncoder's input buffer
        lowest = torch.empty(B, H, W, device=dev)
        mask = None
        if isinstance(self.cost_volume, ZeroCostVolumeManager):
            ent["cv_in"].buf.zero_()
            planes = self.cost_volume.generate_depth_planes(B, torch.tensor(self.min_depth, device=dev).view(1, 1, 1, 1),
                                                            torch.tensor(self.max_depth, device=dev).view(1, 1, 1, 1))
            lowest = planes[:, 0]
        elif type(self.cost_volume) is CostVolumeManager:
            mc, ms = matching_cur_feats.contiguous(), matching_src_feats.contiguous()
            _lib.check(L.idh_nchw_to_nhwc_f32(mc.data_ptr(), st["cur_n"].data_ptr(), B, C, H * W, sp), "idh_nchw_to_nhwc_f32")
            _lib.check(L.idh_nchw_to_nhwc_f32(ms.data_ptr(), st["src_n"].data_ptr(), B * K, C, H * W, sp), "idh_nchw_to_nhwc_f32")
            Ks_c, E_c, iK_c = src_K.contiguous(), src_cam_T_cur_cam.contiguous(), cur_invK.contiguous()  # alive until enqueued
            _lib.check(L.idh_cost_volume_dot_fwd(st["cur_n"].data_ptr(), st["src_n"].data_ptr(), Ks_c.data_ptr(),
                                                 E_c.data_ptr(), iK_c.data_ptr(),
                                                 self.min_depth, self.max_depth, B, K, C, H, W, D, ent["cv_in"].ptr, ent["cv_in"].cs,
                                                 lowest.data_ptr(), st["planes"].data_ptr(), sp), "idh_cost_volume_dot_fwd")
        else:
            lowest, mask = self.cost_volume.fused_into(ent["cv_in"], st, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam,
                                                       cur_cam_T_src_cam, src_K, cur_invK, self.min_depth, self.max_depth, return_mask)

        # 2. CVEncoder + UNet++ decoder: one idh_run_ops call
        for idx, f in zip(ent["i_enc"], cur_feats):
            p.set_in(idx, f)
        final = ent["final"]
        if ent["heads"]:
            for i, idx in ent["heads"].items():
                v = final[i]
                t = torch.empty(B, 1, v.H, v.W, device=dev)
                p.set_out(idx, t)
                out[f"log_depth_pred_s{i}_b1hw"] = t
        p.run()
        if ent["heads"]:
            for i in ent["heads"]:
                out[f"depth_pred_s{i}_b1hw"] = torch.exp(out[f"log_depth_pred_s{i}_b1hw"])  # depth_model.py:425-433

        # 3. occlusion MLP over every query plane (BDModel only)
        if self.binary_mlp is not None and rendered_depth is not None:
            if prior is None and prior_inputs is not None and prior_inputs.get("prior_prediction") is not None:
                from .mlp import sample_prior

                prior = sample_prior(rendered_depth, prior_inputs["prior_prediction"], prior_inputs["world_T_cam_b44"],
                                     prior_inputs["prior_cam_T_world"], prior_inputs["K_s0_b44"], prior_inputs["invK_s0_b44"])
                out["prior_mask"] = prior
            f0 = final[0]
            if infer_depth:  # bd_model.py:273-292: 12-step per-pixel binary search, one launch
                from .mlp import infer_depth as _search

                out["search_depths"], out["pred_0"] = _search(self.binary_mlp, f0.buf, f0.c0, f0.C, prior[:, :1] if prior is not None else None,
                                                              thresholder=self.thresholder)
            else:
                out["pred_0"] = occlusion_logits(self.binary_mlp, f0.buf, f0.c0, f0.C, rendered_depth, prior)
        if return_features:
            for i, v in final.items():
                out[f"feature_s{i}_b1hw"] = _export(v)
        out["lowest_cost_bhw"] = lowest
        out["overall_mask_bhw"] = mask
        return out


def _export(v: nhwc.View) -> torch.Tensor:
    t = torch.empty(v.N, v.C, v.H, v.W, device=v.buf.device)
    p = nhwc.Plan(v.buf.device)
    p.export_nchw(v, t)
    p.run()
    return t


# --------------------------------------------------------------------------------------------
# bench workload
# --------------------------------------------------------------------------------------------
class HotPathWorkload:
    """bench.py workload: the in-scope part of BDModel.forward on synthetic ScanNet-shaped
    tuples — 512x384 image, matching map 128x96, K source views, D planes, P=8 query planes.
    The third-party image / matching backbones are replaced by resident synthetic feature
    maps of the right shape (they are outside the hot path)."""

    name = "hot_path"
    bound = "mfma"

    def __init__(self, args, device, rank):
        import implicit_depth_amd.synthetic as syn
        from . import networks as net

        self.args = args
        self.B, self.K, self.D = args.batch, args.views, args.planes
        self.Hi, self.Wi = args.height, args.width
        self.H, self.W, self.C = args.height // 4, args.width // 4, 16
        self.P = 8
        enc_ch = [24, 48, 64, 160, 256]
        self.volume = getattr(args, "volume", "dot")
        if self.volume == "mlp":
            from .cost_volume import FeatureVolumeManager

            cv = FeatureVolumeManager(self.H, self.W, self.D, num_source_views=self.K)
            syn.fill_state_dict(cv.mlp, seed=99, gain=1.4)
        else:
            cv = CostVolumeManager(self.H, self.W, self.D)
        cve = net.CVEncoder(self.D, enc_ch[1:], [64, 128, 256, 384])
        dec = net.BDDecoderPP(enc_ch[:1] + cve.num_ch_enc)
        mlp = net.BinaryMLPNetwork(dec.num_ch_dec, mlp_size=128, use_prior=False)
        for i, m in enumerate((cve, dec, mlp)):
            syn.fill_state_dict(m, seed=100 + i)
        self.conv_math = getattr(args, "conv_math", "fp32")
        self.mlp_math = getattr(args, "mlp_math", "fp32")
        if self.volume == "mlp":
            cv.mlp_math = self.mlp_math
        mlp.mlp_math = self.mlp_math
        self.model = HotPath(cv, cve, dec, mlp, conv_math=self.conv_math).to(device)
        inp = syn.cost_volume_inputs(self.B, self.K, self.C, self.H, self.W, seed=rank)
        self.host_inputs = inp
        self.host_pyr = syn.encoder_pyramid(self.B, self.Hi, self.Wi, seed=rank)
        self.host_rd = syn.rendered_depth_planes(self.B, self.Hi // 2, self.Wi // 2, self.P)
        self.d = {k: v.to(device) for k, v in inp.items()}
        self.pyr = [t.to(device) for t in self.host_pyr]
        self.rd = self.host_rd.to(device)
        self.out = None

    def config(self):
        vol = "fused MLP feature volume (FeatureVolumeManager, implicit_depth.yaml)" if self.volume == "mlp" else "fused warp+match (dot, CostVolumeManager)"
        return {"workload": f"{self.name}: matching feats -> {vol} -> CVEncoder -> BDDecoderPP (UNet++) -> occlusion MLP x{self.P} planes; "
                            f"{self.Wi}x{self.Hi} image, matching map {self.W}x{self.H}, K={self.K} source views, D={self.D} planes, fp32; "
                            "image/matching backbones (third-party) replaced by resident synthetic feature maps",
                "per_gpu_batch": self.B, "source_views": self.K, "depth_planes": self.D, "query_planes": self.P, "volume": self.volume,
                "conv_math": self.conv_math, "mlp_math": self.mlp_math}

    def step(self, ev=None):
        d = self.d
        if ev is not None:
            ev[0].record()
        self.out = self.model(d["cur_feats"], d["src_feats"], self.pyr, d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                              rendered_depth=self.rd)
        if ev is not None:
            ev[1].record()

    # roofline of the dominant kernel -------------------------------------------------------
    dominant_kernel = "conv3x3_lds_k<2>"

    def _replay_ms(self, ops, iters=10, batches=3):
        """ms per pass of `ops` replayed alone between HIP events on the launch stream: median of
        `batches` batches of `iters` passes (a single batch right after the timed loop occasionally
        catches a clock / power-state transient: 0.74 vs 0.79 of peak for the same kernel)."""
        import ctypes as C

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
        return sum(2 * op.N * op.Ho * op.Wo * op.Cout * s.Cin * s.ks * s.ks for s in op.src if s.in_)

    def conv_only_ms(self, iters=10):
        """HIP-event timing of (a) the launches of the dominant kernel — the 8-row LDS-staged
        3x3 conv, one launch per op with tile code 8 — and (b) every conv op of the step, each
        set replayed alone on the launch stream.  Returns two (ms_per_step, launches, flops)."""
        ent = next(iter(self.model._plans.values()))
        p = ent["plan"]
        convs = [op for op in p.ops if op.kind == nhwc.OP_CONV]
        dom = [op for op in convs if op.tile_m == 8]
        if self.conv_math != "fp32":
            dom = [op for op in convs if op.tile_m in (10, 11)]
            self.dominant_kernel = "conv3x3_split_k<8, 1, 0, *>" if self.conv_math == "bf16x6" else "conv3x3_split_k<4, 2, 1, *>"
        if not dom:  # small batches: every layer runs on the 4-row tile variant
            dom = [op for op in convs if op.tile_m == 9]
            self.dominant_kernel = "conv3x3_lds_k<1> + conv3x3_lds_group_k<1>"
        dom_res = (self._replay_ms(dom, iters), len(dom), sum(self._conv_flops(o) for o in dom))
        all_res = (self._replay_ms(convs, iters), len(convs), sum(self._conv_flops(o) for o in convs))
        return dom_res, all_res

    def metrics(self):
        """Per-frame metric rows (B, 120): PlaneEvaluator IoU / IoU+ / IoU- for 5 thresholds x 8 query
        planes against a synthetic ground-truth depth — the shape of the dict test_bd.py:288-339
        builds per frame, computed on the GPU (csrc/metrics.hip) and then all-gathered."""
        import implicit_depth_amd.synthetic as syn
        from .metrics import PlaneEvaluator, metric_rows

        o = self.out
        B, P, H, W = o["pred_0"].shape
        gt = (1.0 + 3.5 * torch.sigmoid(syn.randn((B, 1, H, W), 7, "bench_gt"))).to(o["pred_0"].device)
        rows, self.metric_keys = metric_rows(PlaneEvaluator().compute_batch_scores(self.rd, gt, torch.sigmoid(o["pred_0"])))
        return rows

    def cpu_baseline(self, seconds):
        """oracle (torch CPU fp32 restatement) of the same path, one frame at a time."""
        import time

        from oracle import cost_volume as ocv
        from oracle import networks as onet

        i = self.host_inputs
        sd = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        w_cve, w_dec, w_mlp = sd(self.model.cost_volume_net), sd(self.model.depth_decoder), sd(self.model.binary_mlp)
        n, t0 = 0, time.perf_counter()
        ocv.FAST_GATHER = True  # time the restatement with torch's own grid_sample primitive
        with torch.inference_mode():
            while True:
                if self.volume == "mlp":
                    w_fv = {k: v.detach().cpu() for k, v in self.model.cost_volume.mlp.state_dict().items()}
                    cvol = ocv.feature_volume(i["cur_feats"][:1], i["src_feats"][:1], i["src_extrinsics"][:1], i["src_poses"][:1], i["src_Ks"][:1],
                                              i["cur_invK"][:1], 0.25, 5.0, self.D, w_fv)[0]
                else:
                    cvol, _, _ = ocv.cost_volume_dot(i["cur_feats"][:1], i["src_feats"][:1], i["src_extrinsics"][:1], i["src_Ks"][:1], i["cur_invK"][:1], 0.25, 5.0, self.D)
                pyr = [t[:1] for t in self.host_pyr]
                enc = onet.cv_encoder(cvol, pyr[1:], w_cve)
                dec = onet.unetpp_decoder([pyr[0]] + enc, w_dec, depth_head=False)
                onet.occlusion_logits(dec["feature_s0_b1hw"], self.host_rd[:1], w_mlp)
                n += 1
                if time.perf_counter() - t0 > seconds:
                    break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n} frame(s) of the same workload through oracle/ (torch CPU fp32 restatement, grid_sample gather), {dt:.1f} s"}
