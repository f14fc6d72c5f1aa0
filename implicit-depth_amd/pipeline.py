"""The hot path of ``BDModel.forward`` / ``DepthModel.forward`` as one NHWC pipeline.

Covers reference experiment_modules/bd_model.py:231-311 (and depth_model.py:378-433) from
the matching features onwards:

    [matching backbone layer1 map --> encoder head (1x1 conv, InstanceNorm, LeakyReLU, 3x3 conv, InstanceNorm)]
    matching feats (NHWC) --> fused warp+match (cost volume, NHWC out)
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
from .cost_volume import CostVolumeManager, ZeroCostVolumeManager, volume_opts
from .mlp import occlusion_logits


class HotPath(nn.Module):
    """Owns (or shares) the hot-path modules of a BDModel / DepthModel: the cost volume, the CVEncoder, the
    depth decoder, the occlusion MLP and — optionally — the matching encoder, whose head then runs inside the
    same plan and hands its features to the volume kernel channels-last without a copy."""

    def __init__(self, cost_volume: nn.Module, cost_volume_net: nn.Module, depth_decoder: nn.Module,
                 binary_mlp: Optional[nn.Module] = None, min_depth: float = 0.25, max_depth: float = 5.0,
                 conv_math: Optional[str] = None, matching_model: Optional[nn.Module] = None):
        super().__init__()
        self.conv_math = conv_math  # None = nhwc.DEFAULT_MATH ("fp32"); "f16x3": see nhwc.MATH_MODES
        self.cost_volume = cost_volume
        self.cost_volume_net = cost_volume_net
        self.depth_decoder = depth_decoder
        self.binary_mlp = binary_mlp
        self.matching_model = matching_model  # ResnetMatchingEncoder (drop-in or reference): its net[5:] head runs here
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.thresholder = None  # like BDModel.thresholder (bd_model.py:141): per-depth thresholds of the infer_depth search
        self._plans = nhwc.PlanCache()  # LRU: a ragged last batch / alternating shapes replay instead of rebuilding
        _lib.watch_state_dict_loads(self)

    # ------------------------------------------------------------------------------------
    def _plan(self, B, K, C, H, W, enc_shapes: Sequence[Sequence[int]], device, head: Optional[str] = None, head_ch: int = 0):
        """``head``: None = matching features come in finished (NCHW); "nchw" / "nhwc" = the plan starts at the
        matching backbone's layer1 map (B*(K+1), head_ch, H, W) in that physical layout and runs the encoder head."""
        key = (B, K, C, H, W, tuple(tuple(s) for s in enc_shapes), str(device), self.conv_math, head, head_ch,
               nhwc._param_key(self.cost_volume_net), nhwc._param_key(self.depth_decoder),
               nhwc.ParamKey(nhwc._param_key(self.matching_model.net[5]) + nhwc._param_key(self.matching_model.net[8])) if head else None)
        ent = self._plans.get(key)
        if ent is not None:
            return ent
        D = self.cost_volume.num_depth_bins
        p = nhwc.Plan(device, math=self.conv_math)
        st = {"lowest": None, "planes": torch.empty(D, device=device)}
        ent = {"plan": p, "state": st, "heads": {}, "i_l1": None}
        if head is None:
            st["cur_n"] = torch.empty(B, H, W, C, device=device)
            st["src_n"] = torch.empty(B, K, H, W, C, device=device)
            ent["feats"] = (st["cur_n"].data_ptr(), st["src_n"].data_ptr(), (B, K, C, H, W), 0, 0)
        else:
            # matching-encoder head over all B*(K+1) images at once (the reference loops image by image,
            # bd_model.py:149-160; InstanceNorm statistics are per image either way).  Image order = frame b's
            # current view then its K source views, so the result is ONE (B, K+1, H, W, C) buffer that the volume
            # kernel addresses with batch strides.
            M = B * (K + 1)
            n0 = len(p.ops)
            if head == "nchw" and nhwc.FUSE_HEAD_IMPORT and nhwc.Plan.pointwise_nchw_eligible(self.matching_model.net[5]):
                # the first 1x1 conv reads the backbone's NCHW map in place: no layout-import pass
                y, ent["i_l1"] = nhwc.build_matching_head(p, self.matching_model, None, nchw_shape=(M, head_ch, H, W))
            elif head == "nchw":
                x = p.buffer(M, H, W, head_ch)
                ent["i_l1"] = p.import_nchw((M, head_ch, H, W), x)
                y = nhwc.build_matching_head(p, self.matching_model, x)
            else:  # channels-last producer: the first conv reads the caller's tensor in place (pointer patched per call)
                x = nhwc.View(torch.empty(1, device=device).expand(M, H, W, head_ch), 0, head_ch)
                y = nhwc.build_matching_head(p, self.matching_model, x)
                ent["i_l1"] = n0  # the 1x1 conv added first by build_matching_head
            if y.C != C:
                raise _lib.IdhError(f"matching head produces {y.C} channels, cost volume expects {C}")
            hw = H * W * C
            ent["feats"] = (y.ptr, y.ptr + 4 * hw, (B, K, C, H, W), (K + 1) * hw, (K + 1) * hw)
            ent["match_view"] = y
        n_pre = len(p.ops)
        cv_in = p.buffer(B, H, W, D)
        v0 = p.buffer(B, enc_shapes[0][2], enc_shapes[0][3], enc_shapes[0][1])
        i_enc = [p.import_nchw(enc_shapes[0], v0)]
        outs, i_img = nhwc.build_cv_encoder(p, self.cost_volume_net, cv_in, enc_shapes[1:])
        i_enc += i_img
        final = nhwc.build_any_decoder(p, self.depth_decoder, [v0] + outs)
        ent.update(cv_in=cv_in, i_enc=i_enc, final=final)
        if getattr(self.depth_decoder, "depth_head", False):
            for i, v in final.items():
                ent["heads"][i] = p.head(v, self.depth_decoder.convs[f"output_{i}"][1], torch.empty(1, device=device))
        elif hasattr(self.depth_decoder, "out1"):  # SkipDecoderRegression
            for i, (hv, last) in nhwc.build_regression_heads(p, self.depth_decoder, final).items():
                ent["heads"][i] = p.head(hv, last, torch.empty(1, device=device))
        # the volume kernel runs between the matching head and the CVEncoder: two replay segments of one plan
        ent["n_head_ops"] = p.schedule_segments(n_pre)
        return self._plans.put(key, ent)

    # ------------------------------------------------------------------------------------
    def forward(self, matching_cur_feats: Optional[torch.Tensor], matching_src_feats: Optional[torch.Tensor], cur_feats: List[torch.Tensor],
                src_cam_T_cur_cam: torch.Tensor, cur_cam_T_src_cam: torch.Tensor, src_K: torch.Tensor, cur_invK: torch.Tensor,
                rendered_depth: Optional[torch.Tensor] = None, prior: Optional[torch.Tensor] = None,
                return_mask: bool = False, return_features: bool = False,
                prior_inputs: Optional[Dict[str, torch.Tensor]] = None, infer_depth: bool = False,
                matching_layer1: Optional[torch.Tensor] = None, return_matching_feats: bool = False,
                frame_chain: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """``matching_layer1`` (B, K+1, 64, H, W): output of the matching backbone (conv1..layer1 of the ResNet18,
        third-party, run by the caller) for frame b's current image followed by its K source images — the order
        reference bd_model.py:149-160 builds; contiguous or channels-last per image
        (``x.view(-1, 64, H, W)`` in ``torch.channels_last``).  With it the encoder head (networks.py:279-283) runs
        inside this call and ``matching_cur_feats`` / ``matching_src_feats`` must be None.
        ``prior``: an already-warped prior channel (B,P,H/2,W/2), or ``prior_inputs`` = the
        reference's temporal inputs {"prior_prediction", "prior_cam_T_world", "world_T_cam_b44",
        "K_s0_b44", "invK_s0_b44"} (bd_model.py:420-431) to warp it here; with neither, a
        prior-enabled MLP sees the constant -1 (bd_model.py:433-434).
        ``frame_chain``: the B batch entries are CONSECUTIVE FRAMES OF ONE SEQUENCE (the temporal loop of
        inference/inference.py:139-157 over a recorded scan).  Everything up to the decoder does not depend on the previous
        frame, so it runs once for all B frames; only the occlusion MLP is evaluated frame by frame, frame b taking
        sigmoid(pred_0) of frame b-1 - warped by sample_prior with frame b-1's cam_T_world - as its prior, exactly as B
        separate calls would.  Keys: "world_T_cam_b44", "cam_T_world_b44" (B,4,4: the frames' poses), "K_s0_b44",
        "invK_s0_b44" (B,4,4), and the chain's start "prior_prediction" (1,1,H/2,W/2) + "prior_cam_T_world" (1,4,4) (or None
        for the first frame of a sequence).  Returns the usual dictionary with pred_0 of all B frames (the last one's
        sigmoid and pose are the next call's chain start)."""
        _lib.require_cuda_f32(matching_cur_feats, matching_src_feats, matching_layer1, src_cam_T_cur_cam, src_K, cur_invK, rendered_depth, prior, *cur_feats)
        head = None
        head_ch = 0
        if matching_layer1 is not None:
            if matching_cur_feats is not None or matching_src_feats is not None:
                raise _lib.IdhError("pass either finished matching features or matching_layer1, not both")
            if self.matching_model is None:
                raise _lib.IdhError("matching_layer1 needs HotPath(matching_model=...)")
            if matching_layer1.dim() != 5:
                raise _lib.IdhError(f"matching_layer1 must be (B, K+1, C, H, W), got {tuple(matching_layer1.shape)}")
            B, K1, head_ch, H, W = matching_layer1.shape
            K, C = K1 - 1, self.matching_model.net[8].out_channels
            l1 = matching_layer1.reshape(B * K1, head_ch, H, W)
            if l1.is_contiguous():
                head = "nchw"
            elif l1.is_contiguous(memory_format=torch.channels_last):
                head = "nhwc"
            else:
                l1, head = l1.contiguous(), "nchw"
        else:
            B, K, C, H, W = matching_src_feats.shape
        dev = src_K.device
        cur_feats = [f if f.is_contiguous() else f.contiguous() for f in cur_feats]
        ent = self._plan(B, K, C, H, W, [f.shape for f in cur_feats], dev, head, head_ch)
        p, st = ent["plan"], ent["state"]
        L = _lib.lib()
        sp = _lib.stream_ptr()
        D = self.cost_volume.num_depth_bins
        out: Dict[str, torch.Tensor] = {}

        # 0. matching features: the encoder head (first segment of the plan) or a layout import of finished features
        zero_volume = isinstance(self.cost_volume, ZeroCostVolumeManager)
        if head is not None:
            p.set_in(ent["i_l1"], l1)
            p.run(0, ent["n_head_ops"])
        elif not zero_volume:
            mc = matching_cur_feats if matching_cur_feats.is_contiguous() else matching_cur_feats.contiguous()
            ms = matching_src_feats if matching_src_feats.is_contiguous() else matching_src_feats.contiguous()
            _lib.check(L.idh_nchw_to_nhwc_f32(mc.data_ptr(), st["cur_n"].data_ptr(), B, C, H * W, sp), "idh_nchw_to_nhwc_f32")
            _lib.check(L.idh_nchw_to_nhwc_f32(ms.data_ptr(), st["src_n"].data_ptr(), B * K, C, H * W, sp), "idh_nchw_to_nhwc_f32")
        cur_ptr, src_ptr, dims, cbs, sbs = ent["feats"]

        # 1. cost volume, written NHWC straight into the CVEncoder's input buffer
        lowest = torch.empty(B, H, W, device=dev)
        mask = None
        if zero_volume:
            ent["cv_in"].buf.zero_()
            planes = self.cost_volume.generate_depth_planes(B, torch.tensor(self.min_depth, device=dev).view(1, 1, 1, 1),
                                                            torch.tensor(self.max_depth, device=dev).view(1, 1, 1, 1))
            lowest = planes[:, 0]
        elif type(self.cost_volume) is CostVolumeManager:
            mats = [t if t.is_contiguous() else t.contiguous() for t in (src_K, src_cam_T_cur_cam, cur_invK)]  # alive until enqueued
            opts, _keep = volume_opts(B, K, C, H, W, D, None, cbs, sbs, kernel=self.cost_volume.__dict__.get("kernel", 0), dot_scratch_device=dev)
            _lib.check(L.idh_cost_volume_dot_ex_fwd(cur_ptr, src_ptr, mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr(),
                                                    self.min_depth, self.max_depth, B, K, C, H, W, D, ent["cv_in"].ptr, ent["cv_in"].cs,
                                                    lowest.data_ptr(), st["planes"].data_ptr(), opts, sp), "idh_cost_volume_dot_ex_fwd")
        else:
            lowest, mask = self.cost_volume.fused_into(ent["cv_in"], st, ent["feats"], src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK,
                                                       self.min_depth, self.max_depth, return_mask)

        # 2. CVEncoder + UNet++ decoder: one idh_run_ops call
        for idx, f in zip(ent["i_enc"], cur_feats):
            p.set_in(idx, f)
        final = ent["final"]
        if ent["heads"]:
            for i, idx in ent["heads"].items():
                v = final[i]
                t = torch.empty(B, 1, v.H, v.W, device=dev)
                e = torch.empty(B, 1, v.H, v.W, device=dev)
                p.set_out(idx, t, exp_out=e)  # the head kernel writes log-depth and exp(log-depth) (depth_model.py:425-433)
                out[f"log_depth_pred_s{i}_b1hw"] = t
                out[f"depth_pred_s{i}_b1hw"] = e
        p.run(ent["n_head_ops"])

        # 3. occlusion MLP over every query plane (BDModel only)
        if self.binary_mlp is not None and rendered_depth is not None and frame_chain is not None:
            from .mlp import sample_prior

            if infer_depth or prior is not None or prior_inputs is not None:
                raise _lib.IdhError("frame_chain excludes prior / prior_inputs / infer_depth")
            if not getattr(self.binary_mlp, "use_prior", False):
                raise _lib.IdhError("frame_chain needs an occlusion MLP built with use_prior=True")
            for k in ("world_T_cam_b44", "cam_T_world_b44", "K_s0_b44", "invK_s0_b44"):
                if k not in frame_chain or tuple(frame_chain[k].shape) != (B, 4, 4):
                    raise _lib.IdhError(f"frame_chain[{k!r}] must be ({B}, 4, 4): one pose / intrinsics matrix per frame of the batch")
            if (frame_chain.get("prior_prediction") is None) != (frame_chain.get("prior_cam_T_world") is None):
                raise _lib.IdhError("frame_chain: prior_prediction and prior_cam_T_world come together (both None at the first frame of a sequence)")
            _lib.require_cuda_f32(*[t for t in frame_chain.values() if t is not None])
            f0 = final[0]
            prev_pred, prev_cTw = frame_chain.get("prior_prediction"), frame_chain.get("prior_cam_T_world")
            pred = torch.empty(B, rendered_depth.shape[1], f0.H, f0.W, device=dev)
            priors = []
            for b in range(B):
                pb = None
                if prev_pred is not None:
                    pb = sample_prior(rendered_depth[b:b + 1], prev_pred, frame_chain["world_T_cam_b44"][b:b + 1], prev_cTw,
                                      frame_chain["K_s0_b44"][b:b + 1], frame_chain["invK_s0_b44"][b:b + 1])
                priors.append(pb)
                occlusion_logits(self.binary_mlp, f0.buf[b:b + 1], f0.c0, f0.C, rendered_depth[b:b + 1], pb, out=pred[b:b + 1])
                prev_pred, prev_cTw = torch.sigmoid(pred[b:b + 1]), frame_chain["cam_T_world_b44"][b:b + 1]  # sigmoid_custom(x, 1.0), inference.py:154
            out["pred_0"] = pred
            if all(t is not None for t in priors):
                out["prior_mask"] = torch.cat(priors, 0)
        elif self.binary_mlp is not None and rendered_depth is not None:
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
        if return_matching_feats and head is not None:
            y = ent["match_view"]
            feats = _export(y).view(B, K + 1, C, H, W)
            out["matching_cur_feats"], out["matching_src_feats"] = feats[:, 0], feats[:, 1:]
        out["lowest_cost_bhw"] = lowest
        out["overall_mask_bhw"] = mask
        return out


def _export(v: nhwc.View) -> torch.Tensor:
    t = torch.empty(v.N, v.C, v.H, v.W, device=v.buf.device)
    p = nhwc.Plan(v.buf.device)
    p.export_nchw(v, t)
    p.run()
    return t
