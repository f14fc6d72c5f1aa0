#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference; never on the GPU box).  The
reference is imported with empty stub modules standing in for third-party packages that
are absent here (timm, kornia, torchvision, antialiased_cnns — SURVEY.md §8c); none of the
stubbed symbols is on the measured path.  Inputs and weights come from
``implicit_depth_amd.synthetic`` (seeded, name-keyed), so the fixtures only need to hold the
reference's OUTPUTS plus an input checksum.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

os.environ["PYTORCH_JIT"] = "0"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("antialiased_cnns")
    _stub("timm")
    k = _stub("kornia")
    k.filters = _stub("kornia.filters")
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.ops = _stub("torchvision.ops", FeaturePyramidNetwork=object)
    tv.transforms = _stub("torchvision.transforms")
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    sys.path.insert(0, REF)


def chk(t: torch.Tensor) -> np.ndarray:
    t = t.double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    import_reference()
    import implicit_depth_amd.synthetic as syn
    from modules.cost_volume import CostVolumeManager, EfficientCostVolumeManager, FeatureVolumeManager
    from modules.layers import BasicBlock
    from modules.networks import BDDecoderPP, BinaryMLPNetwork, CVEncoder, DepthDecoderPP
    from utils.geometry_utils import BackprojectDepth, Project3D, pose_distance

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)

    # ---- G1: dot-product cost volume -------------------------------------------------
    print("G1 cost volume (dot)")
    cases = {
        # name: (B,K,C,H,W,D, seed, behind_view, big_rotation_view)
        "g1_small": (1, 2, 16, 24, 32, 16, 0, -1, -1),
        "g1_b2k7": (2, 7, 16, 24, 32, 8, 1, 3, 5),
        "g1_ragged": (1, 3, 16, 20, 36, 5, 2, 1, -1),
    }
    for name, (B, K, C, H, W, D, seed, bv, rv) in cases.items():
        inp = syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv)
        m = CostVolumeManager(H, W, D)
        cv, low, planes, mask = m(**inp)
        fast = EfficientCostVolumeManager(H, W, D)
        cvf = fast(**inp)[0]
        assert mask is None
        save(
            name,
            dims=np.array([B, K, C, H, W, D, seed, bv, rv]),
            cost_volume=cv,
            lowest_cost=low,
            planes=planes[0, :, 0, 0],
            fast_maxabs=np.array((cv - cvf).abs().max().item()),
            in_chk=np.stack([chk(inp["cur_feats"]), chk(inp["src_feats"]), chk(inp["src_extrinsics"])]),
        )
    # full-size case (BASELINE config 2: K=8, D=64, 96x128): checksums + strided slices
    B, K, C, H, W, D = 1, 8, 16, 96, 128, 64
    inp = syn.cost_volume_inputs(B, K, C, H, W, 0)
    cv, low, planes, _ = CostVolumeManager(H, W, D)(**inp)
    save(
        "g1_full_k8d64",
        dims=np.array([B, K, C, H, W, D, 0, -1, -1]),
        cost_chk=chk(cv),
        cost_slice=cv[:, ::4, ::6, ::8],
        lowest_chk=chk(low),
        lowest_slice=low[:, ::3, ::4],
        planes=planes[0, :, 0, 0],
    )

    # ---- G2: MLP feature volume ------------------------------------------------------
    print("G2 feature volume (MLP)")
    import contextlib, io

    for name, (B, K, C, H, W, D, seed, bv, rv) in {
        "g2_small": (1, 7, 16, 24, 32, 8, 3, -1, -1),
        "g2_b2": (2, 7, 16, 20, 36, 4, 4, 2, 6),
        "g2_k2": (1, 2, 16, 24, 32, 4, 5, -1, -1),
    }.items():
        inp = syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv)
        with contextlib.redirect_stdout(io.StringIO()):
            m = FeatureVolumeManager(H, W, D, mlp_channels=[202, 128, 128, 1], num_source_views=K)
        syn.fill_state_dict(m.mlp, seed=100 + seed, gain=1.4)
        fv, low, planes, mask = m(**inp, return_mask=True)
        with contextlib.redirect_stdout(io.StringIO()):
            fast = m.to_fast()
        fvf, _, _, maskf = fast(**inp, return_mask=True)
        save(
            name,
            dims=np.array([B, K, C, H, W, D, seed, bv, rv]),
            mlp_seed=np.array(100 + seed),
            feature_volume=fv,
            lowest_cost=low,
            overall_mask=mask,
            fast_maxabs=np.array((fv - fvf).abs().max().item()),
            fast_mask_equal=np.array(bool((mask == maskf).all())),
            mlp_chk=np.stack([chk(v) for v in m.mlp.state_dict().values()]),
        )

    # ---- G3: conv stacks -------------------------------------------------------------
    print("G3 BasicBlock / CVEncoder / decoders")
    x = syn.randn((2, 24, 12, 20), 7, "bb_x")
    for tag, (cin, cout, stride) in {"id": (24, 24, 1), "proj": (24, 40, 1), "down": (24, 40, 2)}.items():
        bb = BasicBlock(cin, cout, stride=stride)
        syn.fill_state_dict(bb, seed=10, gain=1.0)
        save(f"g3_basicblock_{tag}", dims=np.array([cin, cout, stride]), y=bb(x), keys=np.array(sorted(bb.state_dict())))

    Hm, Wm, Dcv = 24, 32, 16
    enc_ch = [24, 48, 64, 160, 256]
    pyr = syn.encoder_pyramid(1, Hm * 4, Wm * 4, seed=11)
    cvol = syn.randn((1, Dcv, Hm, Wm), 11, "cv_in")
    cve = CVEncoder(num_ch_cv=Dcv, num_ch_enc=enc_ch[1:], num_ch_outs=[64, 128, 256, 384])
    syn.fill_state_dict(cve, seed=12, gain=1.0)
    cv_out = cve(cvol, list(pyr[1:]))
    save("g3_cvencoder", **{f"o{i}": o for i, o in enumerate(cv_out)}, keys=np.array(sorted(cve.state_dict())))

    dec_in = [pyr[0]] + cv_out
    dec_ch = [24, 64, 128, 256, 384]
    for cls, nm, key in ((BDDecoderPP, "g3_bddecoder", "feature_s{}_b1hw"), (DepthDecoderPP, "g3_depthdecoder", "log_depth_pred_s{}_b1hw")):
        dec = cls(dec_ch)
        syn.fill_state_dict(dec, seed=13, gain=1.0)
        out = dec(dec_in)
        save(nm, **{f"s{i}": out[key.format(i)] for i in range(4)}, keys=np.array(sorted(dec.state_dict())))
        if cls is BDDecoderPP:
            feat_s0 = out["feature_s0_b1hw"]

    # ---- G4: BinaryMLP + sample_prior -------------------------------------------------
    print("G4 BinaryMLP / sample_prior")
    Bq, Hq, Wq, P = 1, 48, 64, 3
    rd = syn.rendered_depth_planes(Bq, Hq, Wq, P)
    rd[:, 1, :5, :7] = 0.0  # invalid rendered depth -> prior = -1
    prior = torch.sigmoid(syn.randn((Bq, 1, Hq, Wq), 14, "prior"))
    for use_prior in (False, True):
        net = BinaryMLPNetwork([64, 64, 128, 256], mlp_size=128, use_prior=use_prior)
        syn.fill_state_dict(net, seed=15, gain=1.2)
        outs = []
        for p in range(P):
            parts = [rd[:, p : p + 1], feat_s0]
            if use_prior:
                parts.append(prior * 2 - 1 if p == 0 else -torch.ones_like(prior))
            outs.append(net([torch.cat(parts, 1).permute(0, 2, 3, 1)], max_scale_only=True)["pred_0"].permute(0, 3, 1, 2))
        save(f"g4_binarymlp_prior{int(use_prior)}", logits=torch.cat(outs, 1), keys=np.array(sorted(net.state_dict())))

    # sample_prior is a method of BDModel; call the unbound function on a tiny shim that
    # carries the two geometry modules it uses (bd_model.py:395-410).
    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        _stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    sys.modules["kornia"].filters.sobel = None
    try:
        from experiment_modules.bd_model import BDModel

        shim = types.SimpleNamespace(backprojector=BackprojectDepth(Hq, Wq), projector=Project3D())
        Ks0 = syn.intrinsics(Wq, Hq).float()[None]
        cur_pose = syn.source_pose(0).float()[None]
        prev_pose = syn.source_pose(1).float()[None]
        sp = BDModel.sample_prior(
            shim, rd[:, 1:2], prior, cur_pose, torch.linalg.inv(prev_pose), Ks0, torch.linalg.inv(Ks0)
        )
        save("g4_sample_prior", sampled=sp, dims=np.array([Hq, Wq]))
    except Exception as e:  # pragma: no cover - recorded, not fatal
        print("  sample_prior golden skipped:", repr(e))

    # ---- G6: geometry unit vectors ---------------------------------------------------
    print("G6 geometry")
    m = CostVolumeManager(6, 8, 64)
    planes64 = m.generate_depth_planes(1, torch.tensor(0.25).view(1, 1, 1, 1), torch.tensor(5.0).view(1, 1, 1, 1))[0, :, 0, 0]
    m96 = CostVolumeManager(6, 8, 96)
    planes96 = m96.generate_depth_planes(1, torch.tensor(0.25).view(1, 1, 1, 1), torch.tensor(5.0).view(1, 1, 1, 1))[0, :, 0, 0]
    poses = torch.stack([syn.source_pose(k).float() for k in range(8)])
    pd = torch.stack(pose_distance(poses))
    bp = BackprojectDepth(6, 8)
    invK = torch.linalg.inv(syn.intrinsics(8, 6)).float()[None]
    pts = bp(torch.full((1, 1, 6, 8), 1.7), invK)
    pr = Project3D()(pts, syn.intrinsics(8, 6).float()[None], torch.linalg.inv(poses[2])[None])
    save("g6_geometry", planes64=planes64, planes96=planes96, pose_dist=pd, backproject=pts, project=pr)
    gen_bdmodel(syn)
    gen_matching_head(syn)
    gen_skip_decoder(syn)
    gen_depthmodel(syn)
    gen_metrics(syn)


def gen_bdmodel(syn):
    """G5: the reference's BDModel.forward end to end (BASELINE config 1 shape: 128x96 image,
    K=2, D=16, simple_cost_volume) with stand-in backbones; stores the hot path's inputs
    (matching features, encoder pyramid) and the model's outputs."""
    print("G5 BDModel.forward (stub backbones)")
    import timm, antialiased_cnns
    from options import Options
    timm.create_model = lambda *a, **k: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **k: syn.StubResnetStem())
    import contextlib, io
    from experiment_modules.bd_model import BDModel
    for name, fv, K, use_prior in (("g5_bdmodel_dot", "simple_cost_volume", 2, False), ("g5_bdmodel_mlp", "mlp_feature_volume", 7, False)):
        o = Options()
        o.image_width, o.image_height = 128, 96
        o.matching_num_depth_bins = 16
        o.feature_volume_type = fv
        o.model_num_views = K + 1
        o.binary_loss_positive_weight = 1.0
        o.bd_edge_regularision = False
        o.use_prior = use_prior
        torch.nn.Module.save_hyperparameters = lambda self, *a, **k: None
        with contextlib.redirect_stdout(io.StringIO()):
            model = BDModel(o)
        model.eval()
        syn.fill_state_dict(model, seed=30, gain=1.0)
        cur, src = syn.frame_tuple(1, K, 96, 128, seed=31, P=3)
        captured = {}
        orig = model.compute_matching_feats
        def spy(*a, **k):
            r = orig(*a, **k)
            captured["mc"], captured["ms"] = r
            return r
        model.compute_matching_feats = spy
        enc_orig = model.encoder.forward
        def enc_spy(x):
            r = enc_orig(x)
            captured["enc"] = r
            return r
        model.encoder.forward = enc_spy
        l1 = []
        hook = model.matching_model.net[4].register_forward_hook(lambda m, i, o: l1.append(o.detach().clone()))
        out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
        hook.remove()
        extra = {"overall_mask": out["overall_mask_bhw"]} if out["overall_mask_bhw"] is not None else {}
        extra["layer1"] = torch.cat(l1, 0)[None]  # (1, K+1, 64, H/4, W/4): cur image then the K source images (bd_model.py:149-152)
        if name == "g5_bdmodel_mlp":
            extra.update(_infer_depth_goldens(model, cur, src))
        save(name, K=np.array(K), pred_0=out["pred_0"], lowest_cost=out["lowest_cost_bhw"], matching_cur=captured["mc"],
             matching_src=captured["ms"], **{f"enc{i}": e for i, e in enumerate(captured["enc"])}, **extra,
             keys=np.array(sorted(k for k in model.state_dict() if k.split(".")[0] in ("cost_volume", "cost_volume_net", "depth_decoder", "binary_mlp"))))


def gen_g1_window():
    """G1 case sized for the LDS-window kernel (map >= 64 x 12): B=2, K=7, one view behind the camera, one strongly
    rotated, 28x72 map (ragged 32x8 tiles), D=12 (a partial 16-plane super-group).
        python tests/golden/gen_golden.py g1_win
    """
    import_reference()
    import implicit_depth_amd.synthetic as syn
    from modules.cost_volume import CostVolumeManager

    torch.set_grad_enabled(False)
    print("G1 cost volume (dot), window-kernel shape")
    B, K, C, H, W, D, seed, bv, rv = 2, 7, 16, 28, 72, 12, 6, 3, 5
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed, bv, rv)
    cv, low, planes, mask = CostVolumeManager(H, W, D)(**inp)
    save("g1_win_b2k7", dims=np.array([B, K, C, H, W, D, seed, bv, rv]), cost_volume=cv, lowest_cost=low, planes=planes[0, :, 0, 0],
         in_chk=np.stack([chk(inp["cur_feats"]), chk(inp["src_feats"]), chk(inp["src_extrinsics"])]))


def gen_custom_planes():
    """G13: both managers called with a caller-supplied per-pixel ``depth_planes_bdhw`` (cost_volume.py:324-347) and with
    per-sample (B,1,1,1) min/max depth tensors (generate_depth_planes broadcasts them, :98-132).
        python tests/golden/gen_golden.py g13
    """
    import contextlib, io

    import_reference()
    import implicit_depth_amd.synthetic as syn
    from modules.cost_volume import CostVolumeManager, FeatureVolumeManager

    torch.set_grad_enabled(False)
    print("G13 caller-supplied depth planes / per-sample depth ranges")
    B, K, C, H, W, D, seed = 2, 3, 16, 20, 36, 6, 7
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed, 1, -1)
    planes = syn.custom_depth_planes(B, D, H, W, seed=8)
    cv, low, pl, _ = CostVolumeManager(H, W, D)(**inp, depth_planes_bdhw=planes)
    assert pl is planes
    with contextlib.redirect_stdout(io.StringIO()):
        m = FeatureVolumeManager(H, W, D, mlp_channels=[202, 128, 128, 1], num_source_views=K)
    syn.fill_state_dict(m.mlp, seed=107, gain=1.4)
    fv, flow, _, fmask = m(**inp, depth_planes_bdhw=planes, return_mask=True)
    rng = dict(inp, min_depth=torch.tensor([0.25, 0.4]).view(B, 1, 1, 1), max_depth=torch.tensor([5.0, 3.0]).view(B, 1, 1, 1))
    cv2, low2, pl2, _ = CostVolumeManager(H, W, D)(**rng)
    fv2, flow2, _, _ = m(**rng)
    save("g13_custom_planes", dims=np.array([B, K, C, H, W, D, seed, 1, -1]), cost_volume=cv, lowest_cost=low, feature_volume=fv,
         fv_lowest=flow, fv_mask=fmask, range_cost_volume=cv2, range_lowest=low2, range_planes=pl2[:, :, 0, 0], range_feature_volume=fv2,
         range_fv_lowest=flow2)


def _stub_pytorch3d():
    """utils/binary_metrics_utils.py imports the mesh renderer at module top; none of it is on the path."""
    for name in ("pytorch3d", "pytorch3d.io", "pytorch3d.renderer", "pytorch3d.structures", "pytorch3d.utils"):
        _stub(name)
    sys.modules["pytorch3d.io"].load_ply = None
    for n in ("FoVPerspectiveCameras", "HardFlatShader", "MeshRasterizer", "MeshRenderer", "RasterizationSettings", "TexturesAtlas", "TexturesVertex"):
        setattr(sys.modules["pytorch3d.renderer"], n, None)
    sys.modules["pytorch3d.structures"].Meshes = None
    sys.modules["pytorch3d.utils"].cameras_from_opencv_projection = None


def _thresholder():
    """The reference's Thresholder (utils/binary_metrics_utils.py:42-52) over the 8 test planes with a non-trivial
    per-depth threshold table; its ctor calls .cuda() on the table."""
    _stub_pytorch3d()
    from utils.binary_metrics_utils import Thresholder

    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        return Thresholder(torch.tensor([1.5 + 0.5 * i for i in range(8)]), torch.tensor([0.3, 0.35, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7]))
    finally:
        torch.Tensor.cuda = cuda


def _infer_depth_goldens(model, cur, src, full=False):
    """BDModel.forward(..., infer_depth=True) (bd_model.py:273-292) without and with a Thresholder: the final
    search depths, the logits of the 12th evaluation, and — from the reference's own per-step outputs — each
    pixel's smallest |sigmoid(logit) - threshold| over the 12 decisions (pixels below ~1e-4 may legitimately
    branch the other way in another fp32 implementation)."""
    res = {}
    for tag, th in (("", None), ("_thr", _thresholder())):
        model.thresholder = th
        steps = []
        orig = model.run_mlp_val
        def spy(inputs, fmaps, rd, _o=orig):
            r = _o(inputs, fmaps, rd)
            steps.append((rd.detach().clone(), r["pred_0"].detach().clone()))
            return r
        model.run_mlp_val = spy
        out = model("test", dict(cur), src, unbatched_matching_encoder_forward=True, return_mask=True, infer_depth=True)
        model.run_mlp_val = orig
        assert len(steps) == 12
        margin = None
        for rd, pr in steps:
            t = 0.5 if th is None else th.get_thresholds(rd)
            m = (torch.sigmoid(pr) - t).abs()
            margin = m if margin is None else torch.minimum(margin, m)
        sl = (lambda t: t[:, :, ::6, ::8]) if full else (lambda t: t)
        res[f"search_depths{tag}"] = sl(out["search_depths"])
        res[f"search_pred{tag}"] = sl(out["pred_0"])
        res[f"search_margin{tag}"] = sl(margin)
        if full:
            res[f"search_depths{tag}_chk"] = chk(out["search_depths"])
    model.thresholder = None
    return res


def gen_depthmodel(syn):
    """G9: the reference's DepthModel.forward (SimpleRecon regression baseline, depth_model.py:280-440)
    with stand-in backbones: mlp_feature_volume K=2 (DepthModel passes num_source_views) + DepthDecoderPP."""
    print("G9 DepthModel.forward (stub backbones)")
    import contextlib, io
    k = sys.modules["kornia"]
    for name in ("losses", "geometry", "utils"):
        setattr(k, name, _stub("kornia." + name))
    try:
        from options import Options
        from experiment_modules.depth_model import DepthModel
    except Exception as e:
        print("  DepthModel golden skipped (import):", repr(e))
        return
    K = 2
    o = Options()
    o.image_width, o.image_height = 128, 96
    o.matching_num_depth_bins = 16
    o.feature_volume_type = "mlp_feature_volume"
    o.model_num_views = K + 1
    torch.nn.Module.save_hyperparameters = lambda self, *a, **k: None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            model = DepthModel(o)
    except Exception as e:
        print("  DepthModel golden skipped (ctor):", repr(e))
        return
    model.eval()
    syn.fill_state_dict(model, seed=33, gain=1.0)
    cur, src = syn.frame_tuple(1, K, 96, 128, seed=34, P=1)
    captured = {}
    orig = model.compute_matching_feats
    def spy(*a, **kw):
        r = orig(*a, **kw)
        captured["mc"], captured["ms"] = r
        return r
    model.compute_matching_feats = spy
    enc_orig = model.encoder.forward
    def enc_spy(x):
        r = enc_orig(x)
        captured["enc"] = r
        return r
    model.encoder.forward = enc_spy
    out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
    keep = {k: v for k, v in out.items() if torch.is_tensor(v) and (k.startswith("log_depth_pred") or k.startswith("depth_pred") or k in ("lowest_cost_bhw", "overall_mask_bhw"))}
    save("g9_depthmodel", K=np.array(K), matching_cur=captured["mc"], matching_src=captured["ms"], **{f"enc{i}": e for i, e in enumerate(captured["enc"])}, **keep,
         keys=np.array(sorted(k for k in model.state_dict() if k.split(".")[0] in ("cost_volume", "cost_volume_net", "depth_decoder"))))


def metric_inputs(syn, B=2, D=8, H=24, W=32):
    q = syn.rendered_depth_planes(B, H, W, D).clone()
    q[:, 2, :3, :5] = -1.0                      # masked-out query pixels (surface/boundary style)
    gt = 1.0 + 3.5 * torch.sigmoid(syn.randn((B, 1, H, W), 60, "gt"))
    gt[:, :, -4:, :6] = 0.0                     # missing ground truth
    pred = torch.sigmoid(1.5 * syn.randn((B, D, H, W), 61, "pred"))
    return q, gt, pred


def gen_metrics(syn):
    """G10: PlaneEvaluator IoU scores (constant thresholds + per-depth Thresholder) and
    compute_depth_metrics_batched from the reference's utils/."""
    print("G10 evaluation metrics")
    _stub_pytorch3d()
    from utils.binary_metrics_utils import PlaneEvaluator, Thresholder
    from utils.metrics_utils import compute_depth_metrics_batched
    q, gt, pred = metric_inputs(syn)
    ev = PlaneEvaluator()
    sc = ev.compute_batch_scores(q, gt, pred, tag="surface")
    keys = sorted(sc)
    planes = torch.tensor([1.5 + 0.5 * x for x in range(8)])
    thr = torch.linspace(0.35, 0.65, 8)
    th = Thresholder.__new__(Thresholder)
    th.bins = torch.zeros_like(planes); th.bins[:-1] = (planes[1:] + planes[:-1]) / 2; th.bins[-1] = 100.0
    th.thresholds = thr
    sc2 = ev.compute_batch_scores_test(q, gt, pred, th)
    keys2 = sorted(sc2)
    valid = gt.flatten(1) > 0.5
    dm = compute_depth_metrics_batched(gt.flatten(1), (gt * (1 + 0.2 * syn.randn(gt.shape, 62, "noise"))).clamp_min(0.1).flatten(1), valid)
    keys3 = sorted(dm)
    save("g10_metrics", iou_keys=np.array(keys), iou=torch.stack([sc[k] for k in keys], 1), iou_thr_keys=np.array(keys2),
         iou_thr=torch.stack([sc2[k] for k in keys2], 1), thr_values=thr, dm_keys=np.array(keys3), dm=torch.stack([dm[k] for k in keys3], 1))


def gen_matching_head(syn):
    """G7: head of the reference ResnetMatchingEncoder (networks.py:279-283) on a given 64-channel map."""
    print("G7 matching-encoder head")
    from modules.networks import ResnetMatchingEncoder
    enc = ResnetMatchingEncoder(18, 16)
    syn.fill_state_dict(enc, seed=40, gain=1.0)
    x = syn.randn((3, 64, 24, 32), 41, "mh_x")
    y = enc.net[5:](x)
    save("g7_matching_head", y=y, keys=np.array(sorted(k for k in enc.state_dict() if k.split(".")[1] in ("5", "8"))))


def gen_skip_decoder(syn):
    """G8: SkipDecoder / SkipDecoderRegression (networks_fast.py) on the G3 decoder inputs."""
    print("G8 skip decoders")
    from modules.networks_fast import SkipDecoder, SkipDecoderRegression
    pyr = syn.encoder_pyramid(1, 96, 128, seed=11)
    enc = [torch.as_tensor(np.load(os.path.join(HERE, "g3_cvencoder.npz"))[f"o{i}"]) for i in range(4)]
    feats = [pyr[0]] + enc
    for cls, nm in ((SkipDecoder, "g8_skipdecoder"), (SkipDecoderRegression, "g8_skipdecoder_reg")):
        dec = cls([24, 64, 128, 256, 384])
        syn.fill_state_dict(dec, seed=45, gain=1.0)
        out = dec(feats)
        save(nm, **{k: v for k, v in out.items()}, keys=np.array(sorted(dec.state_dict())))


def gen_full_feature_volume():
    """G2 at BASELINE.json's full size (96x128 matching map, K=7, D=64, the reference-native BDModel volume):
    checksums + strided slices of the reference FeatureVolumeManager's outputs.
        python tests/golden/gen_golden.py g2_full
    """
    import contextlib, io

    import_reference()
    import implicit_depth_amd.synthetic as syn
    from modules.cost_volume import FeatureVolumeManager

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    print("G2 feature volume (MLP), full size")
    B, K, C, H, W, D, seed = 1, 7, 16, 96, 128, 64, 0
    inp = syn.cost_volume_inputs(B, K, C, H, W, seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = FeatureVolumeManager(H, W, D, mlp_channels=[202, 128, 128, 1], num_source_views=K)
    syn.fill_state_dict(m.mlp, seed=100 + seed, gain=1.4)
    fv, low, planes, mask = m(**inp, return_mask=True)
    save(
        "g2_full_k7d64",
        dims=np.array([B, K, C, H, W, D, seed, -1, -1]),
        mlp_seed=np.array(100 + seed),
        fv_chk=chk(fv),
        fv_slice=fv[:, ::4, ::6, ::8],
        lowest_chk=chk(low),
        lowest_slice=low[:, ::3, ::4],
        mask_count=np.array(int(mask.sum().item())),
        mask_slice=mask[:, ::3, ::4],
        mlp_chk=np.stack([chk(v) for v in m.mlp.state_dict().values()]),
    )


def gen_full_depthmodel():
    """G9 at full size: the reference's DepthModel.forward (depth_model.py:280-440) on a 512x384 8-frame tuple
    (mlp_feature_volume K=7, D=64, DepthDecoderPP heads); synthetic backbone features as in g5_full.
        python tests/golden/gen_golden.py g9_full
    """
    import contextlib, io

    import_reference()
    import implicit_depth_amd.synthetic as syn
    import timm, antialiased_cnns

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    print("G9 DepthModel.forward, full size (synthetic backbone features)")
    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        _stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    k = sys.modules["kornia"]
    k.filters.sobel = None
    for name in ("losses", "geometry", "utils"):
        setattr(k, name, _stub("kornia." + name))
    timm.create_model = lambda *a, **kw: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **kw: syn.StubResnetStem())
    from options import Options
    from experiment_modules.depth_model import DepthModel

    K, Hi, Wi, D = 7, 384, 512, 64
    o = Options()
    o.image_width, o.image_height = Wi, Hi
    o.matching_num_depth_bins = D
    o.feature_volume_type = "mlp_feature_volume"
    o.model_num_views = K + 1
    torch.nn.Module.save_hyperparameters = lambda self, *a, **kw: None
    with contextlib.redirect_stdout(io.StringIO()):
        model = DepthModel(o)
    model.eval()
    syn.fill_state_dict(model, seed=33, gain=1.0)
    cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=34, P=1)
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 75, "mc")
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 76, "ms")
    pyr = list(syn.encoder_pyramid(1, Hi, Wi, seed=77))
    model.compute_matching_feats = lambda *a, **kw: (mc, ms)
    model.encoder.forward = lambda x: pyr
    out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
    arrs = {}
    for i in range(4):
        for nm in (f"log_depth_pred_s{i}_b1hw", f"depth_pred_s{i}_b1hw"):
            arrs[nm + "_chk"] = chk(out[nm])
            arrs[nm + "_slice"] = out[nm][:, :, ::3, ::4] if i >= 2 else out[nm][:, :, ::6, ::8]
    save("g9_full_depthmodel", dims=np.array([K, Hi, Wi, D]), lowest_slice=out["lowest_cost_bhw"][:, ::3, ::4],
         mask_slice=out["overall_mask_bhw"][:, ::3, ::4], **arrs,
         keys=np.array(sorted(kk for kk in model.state_dict() if kk.split(".")[0] in ("cost_volume", "cost_volume_net", "depth_decoder"))))


def gen_full_temporal():
    """BASELINE.json config 5 at full size: the reference's BDModel.forward with the temporal prior —
    512x384 8-frame tuple, mlp_feature_volume K=7, **96** depth planes, prior-enabled occlusion MLP, the previous
    frame's prediction warped by sample_prior (bd_model.py:395-449).  Synthetic backbone features as in g5_full.
        python tests/golden/gen_golden.py g5_temporal
    """
    import contextlib, io

    import_reference()
    import implicit_depth_amd.synthetic as syn
    import timm, antialiased_cnns
    from options import Options

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    print("G5 BDModel.forward with temporal prior, full size, D=96")
    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        _stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    sys.modules["kornia"].filters.sobel = None
    timm.create_model = lambda *a, **k: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **k: syn.StubResnetStem())
    from experiment_modules.bd_model import BDModel

    torch.nn.Module.cuda = lambda self, *a, **k: self  # the ctor / run_mlp_val call .cuda() on the geometry helpers
    K, Hi, Wi, D, P = 7, 384, 512, 96, 1
    o = Options()
    o.image_width, o.image_height = Wi, Hi
    o.matching_num_depth_bins = D
    o.feature_volume_type = "mlp_feature_volume"
    o.model_num_views = K + 1
    o.binary_loss_positive_weight = 1.0
    o.bd_edge_regularision = False
    o.use_prior = True
    torch.nn.Module.save_hyperparameters = lambda self, *a, **k: None
    with contextlib.redirect_stdout(io.StringIO()):
        model = BDModel(o)
    model.eval()
    syn.fill_state_dict(model, seed=30, gain=1.0)
    cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=31, P=P)
    cur["prior_prediction"] = torch.sigmoid(syn.randn((1, 1, Hi // 2, Wi // 2), 74, "prior"))
    cur["prior_cam_T_world"] = torch.linalg.inv(syn.source_pose(1).float())[None]
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 71, "mc")
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 72, "ms")
    pyr = list(syn.encoder_pyramid(1, Hi, Wi, seed=73))
    model.compute_matching_feats = lambda *a, **k: (mc, ms)
    model.encoder.forward = lambda x: pyr
    out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
    save(
        "g5_full_temporal_d96",
        dims=np.array([K, Hi, Wi, D, P]),
        pred_chk=chk(out["pred_0"]),
        pred_slice=out["pred_0"][:, :, ::6, ::8],
        prior_mask_chk=chk(cur["prior_mask"]),
        prior_mask_slice=cur["prior_mask"][:, :, ::6, ::8],
        lowest_slice=out["lowest_cost_bhw"][:, ::3, ::4],
        keys=np.array(sorted(k for k in model.state_dict() if k.split(".")[0] in ("cost_volume", "cost_volume_net", "depth_decoder", "binary_mlp"))),
        **_infer_depth_goldens(model, cur, src, full=True),  # the binary depth search with the prior channel
    )
    gen_temporal_sequence(model, syn, K, Hi, Wi)


def gen_temporal_sequence(model, syn, K, Hi, Wi, T=16):
    """BASELINE.json config 5 as it actually runs: the reference's temporal inference loop (inference/inference.py:
    139-157) over T frames, each forward receiving the previous frame's sigmoid(pred_0) and cam_T_world as the prior —
    D=96, prior-enabled occlusion MLP, one query plane at 2 m, starting at the matching backbone's layer1 map
    (reference encoder head applied image by image)."""
    print(f"G5 temporal sequence, {T} frames")
    prev_pred = prev_cam_T_world = None
    preds, priors = [], []
    for t in range(T):
        cur, src, l1, pyr = syn.temporal_frame(t, K, Hi, Wi, seed=31)
        def head_feats(*a, _l1=l1, **k):
            f = torch.cat([model.matching_model.net[5:](x) for x in _l1[0].split(1, dim=0)], 0)[None]
            return f[:, 0], f[:, 1:].contiguous()
        model.compute_matching_feats = head_feats
        model.encoder.forward = lambda x, _p=list(pyr): _p
        cur["prior_prediction"], cur["prior_cam_T_world"] = prev_pred, prev_cam_T_world
        out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True, infer_depth=False)
        prev_pred = torch.sigmoid(out["pred_0"])  # sigmoid_custom(x, multiplier=1.0), inference.py:154
        prev_cam_T_world = cur["cam_T_world_b44"]
        preds.append(out["pred_0"])
        priors.append(cur["prior_mask"] if "prior_mask" in cur else -torch.ones_like(out["pred_0"]))
    pred = torch.cat(preds, 0)
    save("g5_temporal_seq16", dims=np.array([K, Hi, Wi, model.cost_volume.num_depth_bins, T]),
         pred_slice=pred[:, :, ::6, ::8], pred_chk=np.stack([chk(p) for p in preds]),
         prior_slice=torch.cat(priors, 0)[:, :, ::6, ::8])


def gen_full_bdmodel():
    """G5 at BASELINE.json's full size: the reference's BDModel.forward on a 512x384 8-frame tuple
    (mlp_feature_volume, K=7, D=64, 8 query planes).  The third-party backbones are replaced by seeded synthetic
    feature maps (regenerated from the same seeds on the test side), so the fixture holds only checksums and
    strided slices of the model's outputs.
        python tests/golden/gen_golden.py g5_full
    """
    import contextlib, io

    import_reference()
    import implicit_depth_amd.synthetic as syn
    import timm, antialiased_cnns
    from options import Options

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    print("G5 BDModel.forward, full size (synthetic backbone features)")
    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        _stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    sys.modules["kornia"].filters.sobel = None
    timm.create_model = lambda *a, **k: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **k: syn.StubResnetStem())
    from experiment_modules.bd_model import BDModel

    for name, fvt, K in (("g5_full_bdmodel_mlp", "mlp_feature_volume", 7), ("g5_full_bdmodel_dot", "simple_cost_volume", 8)):
        Hi, Wi, D, P = 384, 512, 64, 8
        o = Options()
        o.image_width, o.image_height = Wi, Hi
        o.matching_num_depth_bins = D
        o.feature_volume_type = fvt
        o.model_num_views = K + 1
        o.binary_loss_positive_weight = 1.0
        o.bd_edge_regularision = False
        o.use_prior = False
        torch.nn.Module.save_hyperparameters = lambda self, *a, **k: None
        with contextlib.redirect_stdout(io.StringIO()):
            model = BDModel(o)
        model.eval()
        syn.fill_state_dict(model, seed=30, gain=1.0)
        cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=31, P=P)
        mc = syn.randn((1, 16, Hi // 4, Wi // 4), 71, "mc")
        ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 72, "ms")
        pyr = list(syn.encoder_pyramid(1, Hi, Wi, seed=73))
        model.compute_matching_feats = lambda *a, **k: (mc, ms)
        model.encoder.forward = lambda x: pyr
        out = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
        extra = {}
        if out["overall_mask_bhw"] is not None:
            extra = {"mask_count": np.array(int(out["overall_mask_bhw"].sum().item())), "mask_slice": out["overall_mask_bhw"][:, ::3, ::4]}
        # --- the same forward starting one step earlier: at the matching backbone's layer1 map, through the
        # reference's own encoder head net[5:] applied image by image (bd_model.py:149-160, networks.py:279-283)
        layer1 = syn.layer1_maps(1, K, Hi // 4, Wi // 4, seed=78)
        def head_feats(*a, _m=model, _l1=layer1, **k):
            f = torch.cat([_m.matching_model.net[5:](x) for x in _l1[0].split(1, dim=0)], 0)[None]
            return f[:, 0], f[:, 1:].contiguous()
        model.compute_matching_feats = head_feats
        outh = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)
        fc, fs = head_feats()
        extra.update(head_pred_chk=chk(outh["pred_0"]), head_pred_slice=outh["pred_0"][:, :, ::6, ::8],
                     head_lowest_slice=outh["lowest_cost_bhw"][:, ::3, ::4], head_feats_chk=chk(torch.cat([fc[:, None], fs], 1)),
                     head_feats_slice=torch.cat([fc[:, None], fs], 1)[:, :, :, ::6, ::8])
        if outh["overall_mask_bhw"] is not None:
            extra["head_mask_slice"] = outh["overall_mask_bhw"][:, ::3, ::4]
        if fvt == "mlp_feature_volume":
            extra.update({"head_" + k: v for k, v in _infer_depth_goldens(model, cur, src, full=True).items()})
        save(
            name,
            dims=np.array([K, Hi, Wi, D, P]),
            pred_chk=chk(out["pred_0"]),
            pred_slice=out["pred_0"][:, :, ::6, ::8],
            lowest_chk=chk(out["lowest_cost_bhw"]),
            lowest_slice=out["lowest_cost_bhw"][:, ::3, ::4],
            keys=np.array(sorted(k for k in model.state_dict() if k.split(".")[0] in ("cost_volume", "cost_volume_net", "depth_decoder", "binary_mlp"))),
            head_keys=np.array(sorted(k for k in model.state_dict() if k.startswith("matching_model.net.5") or k.startswith("matching_model.net.8"))),
            **extra,
        )


def blocks(t: torch.Tensor, bd: int, bh: int, bw: int) -> np.ndarray:
    """float64 sums over (bd, bh, bw) blocks of a (1, D, H, W) tensor: a localised error anywhere in the tensor moves one of them"""
    t = t[0].double()
    D, H, W = t.shape
    assert D % bd == 0 and H % bh == 0 and W % bw == 0
    return t.reshape(D // bd, bd, H // bh, bh, W // bw, bw).sum((1, 3, 5)).numpy()


def gen_block_checksums():
    """Per-block checksums of the FULL-SIZE reference outputs (g_full_blocks.npz): the strided slices + three global checksums of
    g1_full / g2_full / g5_full leave room for an error confined to a few voxels between slice points.
        python tests/golden/gen_golden.py blocks
    """
    import contextlib, io

    import_reference()
    import implicit_depth_amd.synthetic as syn
    import timm, antialiased_cnns
    from modules.cost_volume import CostVolumeManager, FeatureVolumeManager
    from options import Options

    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    print("block checksums of the full-size goldens")
    out = {}
    B, K, C, H, W, D = 1, 8, 16, 96, 128, 64
    cv = CostVolumeManager(H, W, D)(**syn.cost_volume_inputs(B, K, C, H, W, 0))[0]
    out["g1_full_k8d64_cost_8x8x8"] = blocks(cv, 8, 8, 8)
    K = 7
    with contextlib.redirect_stdout(io.StringIO()):
        m = FeatureVolumeManager(H, W, D, mlp_channels=[202, 128, 128, 1], num_source_views=K)
    syn.fill_state_dict(m.mlp, seed=100, gain=1.4)
    fv = m(**syn.cost_volume_inputs(B, K, C, H, W, 0), return_mask=True)[0]
    out["g2_full_k7d64_fv_8x8x8"] = blocks(fv, 8, 8, 8)
    # g5_full_bdmodel_mlp: pred_0 from finished matching features and from the layer1 map (exactly gen_full_bdmodel's two forwards)
    for name in ("pytorch_lightning", "moviepy", "moviepy.editor"):
        _stub(name)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["moviepy"].editor = sys.modules["moviepy.editor"]
    sys.modules["kornia"].filters.sobel = None
    timm.create_model = lambda *a, **k: syn.StubImageEncoder()
    for nm in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(antialiased_cnns, nm, lambda *a, **k: syn.StubResnetStem())
    from experiment_modules.bd_model import BDModel

    Hi, Wi, P = 384, 512, 8
    o = Options()
    o.image_width, o.image_height = Wi, Hi
    o.matching_num_depth_bins = D
    o.feature_volume_type = "mlp_feature_volume"
    o.model_num_views = K + 1
    o.binary_loss_positive_weight = 1.0
    o.bd_edge_regularision = False
    o.use_prior = False
    torch.nn.Module.save_hyperparameters = lambda self, *a, **k: None
    with contextlib.redirect_stdout(io.StringIO()):
        model = BDModel(o)
    model.eval()
    syn.fill_state_dict(model, seed=30, gain=1.0)
    cur, src = syn.frame_tuple(1, K, Hi, Wi, seed=31, P=P)
    mc = syn.randn((1, 16, Hi // 4, Wi // 4), 71, "mc")
    ms = syn.randn((1, K, 16, Hi // 4, Wi // 4), 72, "ms")
    pyr = list(syn.encoder_pyramid(1, Hi, Wi, seed=73))
    model.compute_matching_feats = lambda *a, **k: (mc, ms)
    model.encoder.forward = lambda x: pyr
    pred = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)["pred_0"]
    out["g5_full_bdmodel_mlp_pred_1x8x8"] = blocks(pred, 1, 8, 8)
    layer1 = syn.layer1_maps(1, K, Hi // 4, Wi // 4, seed=78)

    def head_feats(*a, _m=model, _l1=layer1, **k):
        f = torch.cat([_m.matching_model.net[5:](x) for x in _l1[0].split(1, dim=0)], 0)[None]
        return f[:, 0], f[:, 1:].contiguous()

    model.compute_matching_feats = head_feats
    predh = model("test", cur, src, unbatched_matching_encoder_forward=True, return_mask=True)["pred_0"]
    out["g5_full_bdmodel_mlp_head_pred_1x8x8"] = blocks(predh, 1, 8, 8)
    out["scales"] = np.array([cv.abs().max().item(), fv.abs().max().item(), pred.abs().max().item(), predh.abs().max().item()])
    save("g_full_blocks", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "blocks":
        gen_block_checksums()
    elif len(sys.argv) > 1 and sys.argv[1] == "g2_full":
        gen_full_feature_volume()
    elif len(sys.argv) > 1 and sys.argv[1] == "g5_full":
        gen_full_bdmodel()
    elif len(sys.argv) > 1 and sys.argv[1] == "g5_temporal":
        gen_full_temporal()
    elif len(sys.argv) > 1 and sys.argv[1] == "g1_win":
        gen_g1_window()
    elif len(sys.argv) > 1 and sys.argv[1] == "g13":
        gen_custom_planes()
    elif len(sys.argv) > 1 and sys.argv[1] == "g9_full":
        gen_full_depthmodel()
    else:
        main()
        gen_full_feature_volume()
        gen_full_bdmodel()
        gen_full_temporal()
        gen_full_depthmodel()
        gen_custom_planes()
        gen_g1_window()
        gen_block_checksums()
