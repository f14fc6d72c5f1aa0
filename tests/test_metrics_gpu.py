"""GPU evaluation metrics (csrc/metrics.hip) vs the reference goldens (G10) and the oracle."""
import numpy as np
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import load_golden
from oracle import metrics as om
from test_oracle_golden import _metric_inputs

pytestmark = pytest.mark.gpu


def test_plane_evaluator_matches_reference():
    from implicit_depth_amd.metrics import PlaneEvaluator, Thresholder, metric_rows

    g = load_golden("g10_metrics")
    q, gt, pred = [t.cuda() for t in _metric_inputs()]
    ev = PlaneEvaluator()
    sc = ev.compute_batch_scores(q, gt, pred, tag="surface")
    rows, keys = metric_rows(sc)
    assert keys == list(g["iou_keys"])
    np.testing.assert_allclose(rows.cpu().numpy(), g["iou"], rtol=1e-6, equal_nan=True)
    planes = torch.tensor([1.5 + 0.5 * x for x in range(8)])
    th = Thresholder(planes, torch.as_tensor(g["thr_values"]))
    sc2 = ev.compute_batch_scores_test(q, gt, pred, th)
    rows2, keys2 = metric_rows(sc2)
    assert keys2 == list(g["iou_thr_keys"])
    np.testing.assert_allclose(rows2.cpu().numpy(), g["iou_thr"], rtol=1e-6, equal_nan=True)
    # thresholder=None falls back to the constant thresholds, like the reference (:136-139)
    assert sorted(ev.compute_batch_scores_test(q, gt, pred, None, tag="surface")) == keys


def test_depth_metrics_match_reference_and_oracle():
    from implicit_depth_amd.metrics import compute_depth_metrics_batched

    g = load_golden("g10_metrics")
    _, gt, _ = _metric_inputs()
    pred = (gt * (1 + 0.2 * syn.randn(gt.shape, 62, "noise"))).clamp_min(0.1)
    valid = gt.flatten(1) > 0.5
    dm = compute_depth_metrics_batched(gt.flatten(1).cuda(), pred.flatten(1).cuda(), valid.cuda())
    np.testing.assert_allclose(torch.stack([dm[k] for k in g["dm_keys"]], 1).cpu().numpy(), g["dm"], rtol=2e-5)
    ref = om.depth_metrics(gt.flatten(1), pred.flatten(1), valid, mult_a=True)
    dm100 = compute_depth_metrics_batched(gt.flatten(1).cuda(), pred.flatten(1).cuda(), valid.cuda(), mult_a=True)
    for k in ref:
        np.testing.assert_allclose(dm100[k].cpu().double().numpy(), ref[k].numpy(), rtol=2e-5)


def test_full_resolution_iou_counts_are_exact():
    """480x640 ground-truth resolution (test_bd.py upsamples to it): integer counting is exact."""
    B, D, H, W = 2, 8, 480, 640
    q = syn.rendered_depth_planes(B, H, W, D)
    gt = 1.0 + 3.5 * torch.sigmoid(syn.randn((B, 1, H, W), 70, "gt"))
    pred = torch.sigmoid(syn.randn((B, D, H, W), 71, "pred"))
    from implicit_depth_amd.metrics import plane_iou

    got = plane_iou(q.cuda(), gt.cuda(), pred.cuda(), [0.3, 0.5, 0.7]).cpu()
    ref = om.plane_iou(q, gt, pred, [0.3, 0.5, 0.7])
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6, equal_nan=True)


def test_thresholder_query_beyond_last_bin_uses_last_threshold():
    """Query depths above bins[-1] (= 100.0) / +inf: bucketize returns nb, where the reference's
    thresholds[idxs] would raise; the kernel takes the last threshold (as the fused infer_depth search
    does, csrc/mlp.hip) instead of reading past the buffer."""
    from implicit_depth_amd.metrics import Thresholder, plane_iou

    B, D, H, W = 1, 3, 16, 24
    q = syn.rendered_depth_planes(B, H, W, D).clone()
    q[:, 1] = 250.0
    q[:, 2] = float("inf")
    gt = torch.full((B, 1, H, W), 300.0)
    gt[..., : W // 2] = 2.0
    pred = torch.sigmoid(syn.randn((B, D, H, W), 72, "pred"))
    planes = torch.tensor([1.5 + 0.5 * x for x in range(8)])
    thr = torch.linspace(0.2, 0.9, 8)
    th = Thresholder(planes, thr)
    got = plane_iou(q.cuda(), gt.cuda(), pred.cuda(), [], bins=th.bins.cuda(), bin_thresholds=thr.cuda()).cpu()
    idx = torch.bucketize(q.flatten(2), th.bins).clamp_max(th.bins.numel() - 1)
    ref = om.plane_iou(q.flatten(2), gt.flatten(2), pred.flatten(2), [], bins=th.bins, bin_thresholds=thr, clamp_index=True)
    assert int(idx.max()) == th.bins.numel() - 1
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6, equal_nan=True)
