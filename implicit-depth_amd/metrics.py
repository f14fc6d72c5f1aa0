"""GPU-side evaluation metrics with the reference's interfaces.

``PlaneEvaluator`` mirrors utils/binary_metrics_utils.py:55-192 (same method names, arguments,
score-dict keys), ``compute_depth_metrics_batched`` mirrors utils/metrics_utils.py:52-120; both run
one counting / summing kernel (csrc/metrics.hip) instead of ~10 NaN-masked full-size temporaries.
``metric_rows`` packs a score dict into the per-frame (B, M) matrix that is all-gathered across
ranks (implicit_depth_amd.dist.all_gather_metrics)."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib

DEPTH_METRIC_KEYS = ("abs_diff", "abs_rel", "sq_rel", "rmse", "rmse_log", "a5", "a10", "a25", "a0", "a1", "a2", "a3")


def _ws(B, D, N, T, device):
    n = _lib.lib().idh_metrics_workspace_bytes(B, D, N, T)
    return torch.empty((n + 7) // 8, device=device, dtype=torch.float64), n


def plane_iou(query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, thresholds: Sequence[float], bins: Optional[torch.Tensor] = None,
              bin_thresholds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, D, T, 3) = [harmonic IoU, positive IoU, negative IoU]."""
    _lib.require_cuda_f32(query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, bins, bin_thresholds)
    B, D = query_depth_bdhw.shape[:2]
    N = query_depth_bdhw[0, 0].numel()
    dev = query_depth_bdhw.device
    q, g, p = query_depth_bdhw.contiguous(), gt_depth_b1hw.contiguous(), prediction_bdhw.contiguous()
    if g.shape[0] != B or g[0].numel() != N or tuple(p.shape) != tuple(q.shape):
        raise _lib.IdhError("plane_iou: shape mismatch between query, gt and prediction")
    if bins is not None:
        thr, T, bins_c = bin_thresholds.contiguous(), 1, bins.contiguous()
        if thr.numel() != bins_c.numel():
            raise _lib.IdhError("Thresholder needs one threshold per bin")
    else:
        thr, T, bins_c = torch.tensor(list(thresholds), device=dev, dtype=torch.float32), len(thresholds), None
    out = torch.empty(B, D, T, 3, device=dev)
    ws, nbytes = _ws(B, D, N, T, dev)
    _lib.check(_lib.lib().idh_plane_iou_fwd(q.data_ptr(), g.data_ptr(), p.data_ptr(), thr.data_ptr(), T, _lib.ptr(bins_c),
                                            0 if bins_c is None else bins_c.numel(), B, D, N, out.data_ptr(), ws.data_ptr(), nbytes,
                                            _lib.stream_ptr()), "idh_plane_iou_fwd")
    return out


class Thresholder:
    """utils/binary_metrics_utils.py:42-52 (per-depth thresholds via bucketize)."""

    def __init__(self, planes: torch.Tensor, thresholds: torch.Tensor):
        self.bins = torch.zeros_like(planes)
        self.bins[:-1] = (planes[1:] + planes[:-1]) / 2
        self.bins[-1] = 100.0
        self.thresholds = thresholds


class PlaneEvaluator:
    def __init__(self, thresholds=np.linspace(0.3, 0.7, 5)):
        self.thresholds = thresholds

    def _scores(self, iou_bdt3, names, is_rendering, tag, depth_planes):
        scores: Dict[str, torch.Tensor] = {}
        pre = "" if tag is None else f"{tag}_"
        for t, tname in enumerate(names):
            for d in range(iou_bdt3.shape[1]):
                dp = -1 if is_rendering else depth_planes[d]
                for j, kind in enumerate(("iou", "iou_pos", "iou_neg")):
                    scores[f"{pre}{kind}_{tname}d_{dp:.1f}"] = iou_bdt3[:, d, t, j]
        return scores

    def compute_batch_scores(self, query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, is_rendering=False, tag=None,
                             depth_planes=tuple(1.5 + x * 0.5 for x in range(8))):
        out = plane_iou(query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, [float(t) for t in self.thresholds])
        return self._scores(out, [f"{float(t):.1f}_" for t in self.thresholds], is_rendering, tag, depth_planes)

    def compute_batch_scores_test(self, query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, thresholder, is_rendering=False, tag=None,
                                  depth_planes=tuple(1.5 + x * 0.5 for x in range(8))):
        if thresholder is None:
            return self.compute_batch_scores(query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, is_rendering, tag, depth_planes)
        dev = query_depth_bdhw.device
        out = plane_iou(query_depth_bdhw, gt_depth_b1hw, prediction_bdhw, [], bins=thresholder.bins.to(dev).float(),
                        bin_thresholds=thresholder.thresholds.to(dev).float())
        return self._scores(out, [""], is_rendering, tag, depth_planes)


def compute_depth_metrics_batched(gt_bN: torch.Tensor, pred_bN: torch.Tensor, valid_masks_bN: torch.Tensor, mult_a: bool = False) -> Dict[str, torch.Tensor]:
    _lib.require_cuda_f32(gt_bN, pred_bN)
    B, N = gt_bN.shape
    g, p = gt_bN.contiguous(), pred_bN.contiguous()
    v = valid_masks_bN.to(torch.uint8).contiguous()
    out = torch.empty(B, len(DEPTH_METRIC_KEYS), device=g.device)
    ws, nbytes = _ws(B, 1, N, 1, g.device)
    _lib.check(_lib.lib().idh_depth_metrics_fwd(g.data_ptr(), p.data_ptr(), v.data_ptr(), B, N, int(mult_a), out.data_ptr(), ws.data_ptr(), nbytes,
                                                _lib.stream_ptr()), "idh_depth_metrics_fwd")
    return {k: out[:, i] for i, k in enumerate(DEPTH_METRIC_KEYS)}


def metric_rows(scores: Dict[str, torch.Tensor], keys: Optional[Sequence[str]] = None):
    """Pack a score dict of (B,) tensors into (B, M) rows + the column names (sorted for a stable
    layout across ranks) — the payload of the path's single collective."""
    keys = sorted(scores) if keys is None else list(keys)
    return torch.stack([scores[k].float() for k in keys], 1), keys
