"""ORACLE (test infrastructure only): the reference's evaluation metrics restated without the NaN
tricks — explicit validity masks and sums.  Anchors: utils/binary_metrics_utils.py:59-192,
utils/metrics_utils.py:52-120."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch


def plane_iou(query_bdn, gt_b1n, pred_bdn, thresholds: Sequence[float], bins: Optional[torch.Tensor] = None,
              bin_thresholds: Optional[torch.Tensor] = None, clamp_index: bool = False) -> torch.Tensor:
    """(B,D,T,3) [iou, iou_pos, iou_neg] in float32 arithmetic like the reference.  ``clamp_index``: queries
    beyond the last bin edge take the last threshold (the reference's ``thresholds[idxs]`` raises there)."""
    q, p = query_bdn.flatten(2), pred_bdn.flatten(2)
    g = gt_b1n.flatten(2).expand_as(q)
    valid = (g > 0) & (q > 0)
    tgt = (q < g) & valid
    if bins is not None:
        idx = torch.bucketize(q, bins)
        thr_list = [bin_thresholds[idx.clamp_max(bins.numel() - 1) if clamp_index else idx]]
    else:
        thr_list = list(thresholds)
    outs = []
    nv = valid.sum(2).float()
    nt = tgt.sum(2).float()
    for thr in thr_list:
        pr = (p > thr) & valid
        np_, ni = pr.sum(2).float(), (pr & tgt).sum(2).float()
        pos = ni / (nt + np_ - ni)
        nn_t, nn_p, nn_i = nv - nt, nv - np_, nv - nt - np_ + ni
        neg = nn_i / (nn_t + nn_p - nn_i)
        outs.append(torch.stack([2 * (pos * neg) / (pos + neg), pos, neg], -1))
    return torch.stack(outs, 2)


def depth_metrics(gt_bn, pred_bn, valid_bn, mult_a=False) -> Dict[str, torch.Tensor]:
    out = {k: [] for k in ("abs_diff", "abs_rel", "sq_rel", "rmse", "rmse_log", "a5", "a10", "a25", "a0", "a1", "a2", "a3")}
    for b in range(gt_bn.shape[0]):
        m = valid_bn[b].bool()
        g, p = gt_bn[b][m].double(), pred_bn[b][m].double()
        th = torch.maximum(g / p, p / g)
        d = g - p
        vals = {"abs_diff": d.abs().mean(), "abs_rel": (d.abs() / g).mean(), "sq_rel": (d * d / g).mean(), "rmse": (d * d).mean().sqrt(),
                "rmse_log": ((g.log() - p.log()) ** 2).mean().sqrt()}
        for k, lim in (("a5", 1.05), ("a10", 1.10), ("a25", 1.25), ("a0", 1.10), ("a1", 1.25), ("a2", 1.25**2), ("a3", 1.25**3)):
            vals[k] = (th < lim).double().mean() * (100.0 if mult_a else 1.0)
        for k in out:
            out[k].append(vals[k])
    return {k: torch.stack(v) for k, v in out.items()}
