"""End-to-end hot path (pipeline.HotPath) vs the oracle chain and vs the module-level
drop-ins.  This is BASELINE config 3's tolerance check at a CPU-oracle-friendly size."""
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from conftest import TOL, rel_err
from oracle import cost_volume as ocv
from oracle import networks as onet

pytestmark = pytest.mark.gpu


def _build(B, K, H, W, D, P, depth_model=False, use_prior=False, seed=0):
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.cost_volume import CostVolumeManager
    from implicit_depth_amd.pipeline import HotPath

    enc_ch = [24, 48, 64, 160, 256]
    cv = CostVolumeManager(H, W, D)
    cve = net.CVEncoder(D, enc_ch[1:], [64, 128, 256, 384])
    dec = (net.DepthDecoderPP if depth_model else net.BDDecoderPP)(enc_ch[:1] + cve.num_ch_enc)
    mlp = None if depth_model else net.BinaryMLPNetwork(dec.num_ch_dec, use_prior=use_prior)
    for i, m in enumerate([cve, dec] + ([mlp] if mlp is not None else [])):
        syn.fill_state_dict(m, seed=seed + 50 + i, gain=1.1 if i == 2 else 1.0)
    inp = syn.cost_volume_inputs(B, K, 16, H, W, seed=seed, behind_view=K - 1)
    pyr = syn.encoder_pyramid(B, H * 4, W * 4, seed=seed)
    rd = syn.rendered_depth_planes(B, H * 2, W * 2, P)
    return HotPath(cv, cve, dec, mlp), inp, pyr, rd


def _oracle(model, inp, pyr, rd, D, depth_model, prior=None):
    sd = lambda m: {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    d = {k: v.double() for k, v in inp.items()}
    cvol, low, _ = ocv.cost_volume_dot(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_Ks"], d["cur_invK"], 0.25, 5.0, D)
    p64 = [t.double() for t in pyr]
    enc = onet.cv_encoder(cvol, p64[1:], sd(model.cost_volume_net))
    dec = onet.unetpp_decoder([p64[0]] + enc, sd(model.depth_decoder), depth_head=depth_model)
    if depth_model:
        return dec, low, None
    logits = onet.occlusion_logits(dec["feature_s0_b1hw"], rd.double(), sd(model.binary_mlp), prior.double() if prior is not None else None)
    return dec, low, logits


@pytest.mark.parametrize("cfg", [(1, 2, 24, 32, 16, 3), (2, 7, 16, 24, 64, 2)])
def test_bd_hot_path_matches_oracle(cfg):
    B, K, H, W, D, P = cfg
    model, inp, pyr, rd = _build(B, K, H, W, D, P)
    dec, low, logits = _oracle(model, inp, pyr, rd, D, False)
    model.cuda()
    d = {k: v.cuda() for k, v in inp.items()}
    out = model(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                rendered_depth=rd.cuda(), return_features=True)
    for i in range(4):
        assert rel_err(out[f"feature_s{i}_b1hw"].cpu(), dec[f"feature_s{i}_b1hw"]) < TOL
    assert out["pred_0"].shape == (B, P, H * 2, W * 2)
    assert rel_err(out["pred_0"].cpu(), logits) < TOL
    assert ((out["lowest_cost_bhw"].cpu().double() - low).abs() > 1e-5).float().mean().item() < 5e-3
    assert out["overall_mask_bhw"] is None
    # replay on new inputs through the cached plan; and agree with module-by-module drop-ins
    inp2 = syn.cost_volume_inputs(B, K, 16, H, W, seed=9)
    d2 = {k: v.cuda() for k, v in inp2.items()}
    out2 = model(d2["cur_feats"], d2["src_feats"], [t.cuda() for t in pyr], d2["src_extrinsics"], d2["src_poses"], d2["src_Ks"], d2["cur_invK"],
                 rendered_depth=rd.cuda(), return_features=True)
    cvol, _, _, _ = model.cost_volume(**dict(d2, min_depth=0.25, max_depth=5.0))  # numbers: planes expanded in the kernel, as in HotPath
    enc = model.cost_volume_net(cvol, [t.cuda() for t in pyr[1:]])
    feats = model.depth_decoder([pyr[0].cuda()] + enc)
    assert rel_err(out2["feature_s0_b1hw"], feats["feature_s0_b1hw"]) < 1e-6


def test_depth_model_hot_path_matches_oracle():
    B, K, H, W, D = 1, 3, 16, 24, 16
    model, inp, pyr, rd = _build(B, K, H, W, D, 1, depth_model=True)
    dec, low, _ = _oracle(model, inp, pyr, rd, D, True)
    model.cuda()
    d = {k: v.cuda() for k, v in inp.items()}
    out = model(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"])
    for i in range(4):
        assert rel_err(out[f"log_depth_pred_s{i}_b1hw"].cpu(), dec[f"log_depth_pred_s{i}_b1hw"]) < TOL
        assert rel_err(out[f"depth_pred_s{i}_b1hw"].cpu(), torch.exp(dec[f"log_depth_pred_s{i}_b1hw"])) < 5e-4


def test_prior_channel_path():
    B, K, H, W, D, P = 1, 2, 16, 24, 8, 1
    model, inp, pyr, rd = _build(B, K, H, W, D, P, use_prior=True)
    prior = torch.sigmoid(syn.randn((B, 1, H * 2, W * 2), 5, "pp")) * 2 - 1
    dec, low, logits = _oracle(model, inp, pyr, rd, D, False, prior=prior)
    model.cuda()
    d = {k: v.cuda() for k, v in inp.items()}
    out = model(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                rendered_depth=rd.cuda(), prior=prior.cuda())
    assert rel_err(out["pred_0"].cpu(), logits) < TOL


def test_temporal_sequence_with_prior_d96():
    """BASELINE config 5 in miniature: 8-frame tuple (K=7), MLP feature volume with 96 planes,
    prior-enabled occlusion MLP, prediction of frame t warped into frame t+1 (inference.py:139-157)."""
    from implicit_depth_amd import networks as net
    from implicit_depth_amd.cost_volume import FeatureVolumeManager
    from implicit_depth_amd.pipeline import HotPath

    B, K, H, W, D = 1, 7, 16, 24, 96
    cv = FeatureVolumeManager(H, W, D, num_source_views=K)
    cve = net.CVEncoder(D, [48, 64, 160, 256], [64, 128, 256, 384])
    dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    mlp = net.BinaryMLPNetwork(dec.num_ch_dec, use_prior=True)
    syn.fill_state_dict(cv.mlp, 70, gain=1.4)
    for i, m in enumerate((cve, dec, mlp)):
        syn.fill_state_dict(m, 71 + i)
    model = HotPath(cv, cve, dec, mlp)
    sd = lambda m: {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    w_fv, w_cve, w_dec, w_mlp = sd(cv.mlp), sd(cve), sd(dec), sd(mlp)
    model.cuda()
    Ks0 = syn.intrinsics(W * 2, H * 2).float()[None]
    prev_pred_ref, prev_pred = None, None
    prev_pose = None
    for t in range(2):
        inp = syn.cost_volume_inputs(B, K, 16, H, W, seed=80 + t)
        pyr = syn.encoder_pyramid(B, H * 4, W * 4, seed=80 + t)
        rd = syn.rendered_depth_planes(B, H * 2, W * 2, 1) * (1.0 + 0.1 * t)
        cur_world_T_cam = syn.source_pose(t).float()[None]
        d = {k: v.double() for k, v in inp.items()}
        vol = ocv.feature_volume(d["cur_feats"], d["src_feats"], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"], 0.25, 5.0, D, w_fv)[0]
        enc = onet.cv_encoder(vol, [p.double() for p in pyr[1:]], w_cve)
        feats = onet.unetpp_decoder([pyr[0].double()] + enc, w_dec, depth_head=False)
        if prev_pred_ref is None:
            prior_ref = -torch.ones(B, 1, H * 2, W * 2, dtype=torch.float64)
        else:
            prior_ref = onet.sample_prior(rd.double(), prev_pred_ref, cur_world_T_cam.double(), torch.linalg.inv(prev_pose).double(), Ks0.double(), torch.linalg.inv(Ks0).double())
        logits_ref = onet.occlusion_logits(feats["feature_s0_b1hw"], rd.double(), w_mlp, prior_ref)
        g = {k: v.cuda() for k, v in inp.items()}
        pin = None
        if prev_pred is not None:
            pin = {"prior_prediction": prev_pred, "prior_cam_T_world": torch.linalg.inv(prev_pose).cuda(), "world_T_cam_b44": cur_world_T_cam.cuda(),
                   "K_s0_b44": Ks0.cuda(), "invK_s0_b44": torch.linalg.inv(Ks0).cuda()}
        out = model(g["cur_feats"], g["src_feats"], [p.cuda() for p in pyr], g["src_extrinsics"], g["src_poses"], g["src_Ks"], g["cur_invK"],
                    rendered_depth=rd.cuda(), prior_inputs=pin)
        assert rel_err(out["pred_0"].cpu(), logits_ref) < 2 * TOL
        prev_pred_ref, prev_pred, prev_pose = torch.sigmoid(logits_ref), torch.sigmoid(out["pred_0"]), cur_world_T_cam


def test_model_built_under_inference_mode():
    """Parameters created inside torch.inference_mode() have no version counter: the packed-weight / plan caches must
    still work (regression: RuntimeError 'Inference tensors do not track version counter')."""
    B, K, H, W, D, P = 1, 2, 16, 24, 8, 2
    with torch.inference_mode():
        model, inp, pyr, rd = _build(B, K, H, W, D, P)
        model.cuda()
        d = {k: v.cuda() for k, v in inp.items()}
        out = model(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                    rendered_depth=rd.cuda())
        out2 = model(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                     rendered_depth=rd.cuda())
    assert torch.equal(out["pred_0"], out2["pred_0"]) and bool(torch.isfinite(out["pred_0"]).all())


def test_plan_cache_replays_alternating_batch_sizes():
    """An eval loop whose last batch is ragged (every scan of test_bd.py:146-152), or a caller alternating two batch sizes,
    must replay cached plans: the plan objects and their buffers stay the same, and the results do not change."""
    (B1, B2), K, H, W, D, P = (3, 2), 2, 16, 32, 16, 2
    model, inp, pyr, rd = _build(B1, K, H, W, D, P)
    model.cuda()

    def run(nb):
        d = {k: v[:nb].cuda() for k, v in inp.items()}
        return model(d["cur_feats"], d["src_feats"], [t[:nb].cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                     rendered_depth=rd[:nb].cuda())["pred_0"].clone()

    y1 = run(B1)
    ents = [id(e["plan"]) for e in model._plans.values()]
    y2 = run(B2)
    assert len(model._plans) == 2
    for _ in range(2):
        assert torch.equal(run(B1), y1) and torch.equal(run(B2), y2)
    assert len(model._plans) == 2 and ents[0] in [id(e["plan"]) for e in model._plans.values()], "the first plan was rebuilt"
    # a build-time switch is part of the key: toggling it builds a new plan instead of replaying the stale one — and the stale twin
    # (same shapes, other switches / parameter versions: it can never be hit again) leaves the cache with its buffers
    from implicit_depth_amd import nhwc

    old = nhwc.MERGE_LEVELS
    nhwc.MERGE_LEVELS = not old
    try:
        y1b = run(B1)
        now = [id(e["plan"]) for e in model._plans.values()]
        assert len(model._plans) == 2 and ents[0] not in now and rel_err(y1b.cpu(), y1.cpu()) < 1e-6
    finally:
        nhwc.MERGE_LEVELS = old


@pytest.mark.parametrize("depth_model", [False, True])
def test_buffer_reuse_aliasing_is_bit_identical(depth_model):
    """Plan.release (nhwc.BUFFER_REUSE) lets later allocations ALIAS the big activation temporaries; the level scheduler must order every new
    writer after the old readers.  The default threshold (64 MiB) pools nothing at test sizes, so this runs the whole HotPath (matching-free:
    CVEncoder + UNet++ decoder + occlusion MLP / depth heads, through schedule_segments) with the threshold at 0 - every released buffer is
    recycled - and demands bit-identical outputs to the plan with private buffers; also with two frames, and replayed (aliased buffers hold
    garbage from the previous run)."""
    from implicit_depth_amd import nhwc

    B, K, H, W, D, P = 2, 2, 24, 32, 16, 2
    old = (nhwc.BUFFER_REUSE, nhwc.REUSE_MIN_BYTES)
    results, recycled = {}, {}
    try:
        for reuse in (False, True):
            nhwc.BUFFER_REUSE, nhwc.REUSE_MIN_BYTES = reuse, 0
            model, inp, pyr, rd = _build(B, K, H, W, D, P, depth_model=depth_model)
            model.cuda()
            d = {k: v.cuda() for k, v in inp.items()}
            outs = []
            for _ in range(2):
                out = model(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"],
                            rendered_depth=None if depth_model else rd.cuda())
                outs.append({k: v.clone() for k, v in out.items() if torch.is_tensor(v)})
            plan = next(iter(model._plans.values()))["plan"]
            recycled[reuse] = (plan.recycled, plan.recycled_candidates)
            results[reuse] = outs
    finally:
        nhwc.BUFFER_REUSE, nhwc.REUSE_MIN_BYTES = old
    print("recycled buffers (taken over, released):", recycled)
    assert recycled[False] == (0, 0) and recycled[True][0] >= 8, recycled
    keys = sorted(results[False][0])
    assert keys and keys == sorted(results[True][0])
    for k in keys:
        for run in range(2):
            assert torch.equal(results[True][run][k], results[False][0][k]), (k, run)


def test_bench_n2_branch_on_one_gpu_over_gloo(tmp_path):
    """bench.py's N > 1 branch (shard, barrier, MAX all-reduce, ragged metric all-gather, rank-0 JSON) under a real launcher with
    real kernels: two ranks share the single GPU of the box over gloo (RCCL refuses duplicate devices; the RCCL leg itself is the
    N = 1 run of profiles/*/torchrun_n1_rccl_debug.log).  Reference analogue: sync_dist logging bd_model.py:672,685, row shape
    test_bd.py:288-339."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rep = tmp_path / "ranks"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29613",
           os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "7", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--ranks-on-device", "0",
           "--no-extras", "--no-split-line", "--no-cpu-baseline", "--rank-report", str(rep)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 7 and out["gathered_metric_rows"] == 7
    assert out["dist_backend"] == "gloo" and "note_ranks_on_device" in out and out["value"] > 0
    ranks = [json.load(open(rep / f"rank{i}.json")) for i in range(2)]
    assert [r_["frames"] for r_ in ranks] == [[0, 4], [4, 7]], "ragged shard: 7 frames over 2 ranks"
    assert ranks[0]["input_checksum"] != ranks[1]["input_checksum"], "per-rank inputs differ (seed = rank)"
    assert all(r_["gathered_rows_match_local"] and r_["gathered_metric_rows"] == 7 for r_ in ranks)
    log = os.path.join(root, "gpurun_out", "bench_n2_gloo_one_gpu.json")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    json.dump({"bench_line": out, "ranks": ranks}, open(log, "w"), indent=1)


def _bench_line(argv, timeout=900, env=None):
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, cwd=root,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"stdout must be exactly one JSON line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0]), r.stderr


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT an outer launcher (the driver's command shape): bench.py starts one process per rank itself
    (torch.distributed.run, 127.0.0.1 rendezvous) and forwards rank 0's JSON line as the only stdout line.  On the 1-GPU box both
    ranks share cuda:0 over gloo; the line must carry the process group's own evidence: backend, world size, one device record per
    rank, the all-gather time.  Reference analogue: Lightning's process-per-GPU launch train.py:124,135."""
    import json

    rep = tmp_path / "ranks"
    out, err = _bench_line(["--gpus", "2", "--batch", "5", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--ranks-on-device", "0",
                            "--no-extras", "--no-cpu-baseline", "--no-parity", "--rank-report", str(rep)])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["dist_backend"] == "gloo" and out["config"]["global_batch"] == 5 and out["scaling"] == "strong"
    assert out["gathered_metric_rows"] == 5 and out["allgather_us"] > 0 and out["launcher"].startswith("self")
    assert [d["rank"] for d in out["devices"]] == [0, 1] and [d["frames"] for d in out["devices"]] == [3, 2]
    assert len({d["pid"] for d in out["devices"]}) == 2, "one PROCESS per rank"
    ranks = [json.load(open(rep / f"rank{i}.json")) for i in range(2)]
    assert [r_["frames"] for r_ in ranks] == [[0, 3], [3, 5]] and all(r_["gathered_rows_match_local"] for r_ in ranks)
    assert "self-launch" in err
    # the default: weak scaling, --batch frames on EVERY rank (DDP semantics: batch_size per process)
    w, _ = _bench_line(["--gpus", "2", "--batch", "3", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--ranks-on-device", "0",
                        "--no-extras", "--no-cpu-baseline", "--no-parity"])
    assert w["scaling"] == "weak" and w["config"]["global_batch"] == 6 and w["config"]["per_gpu_batch"] == 3 and w["gathered_metric_rows"] == 6
    assert [d["frames"] for d in w["devices"]] == [3, 3]


def test_bench_refuses_more_gpus_than_the_box_has():
    """--gpus N above torch.cuda.device_count() (and no --ranks-on-device): rc != 0 with a clear message, nothing launched"""
    import os
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=root)
    assert r.returncode != 0 and "one process per GPU" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")


def test_bench_n1_runs_under_a_one_rank_rccl_group():
    """The driver's N = 1 command (`python bench.py --gpus 1 ...`, no launcher) creates a 1-rank RCCL group: barriers, the MAX all-reduce
    and the metric all-gather go through ncclAllGather / ncclAllReduce, and the line says so.  The rate must not depend on the group:
    within 3 % of the same run with --no-process-group (measured: < 1 %)."""
    import os

    common = ["--gpus", "1", "--batch", "8", "--steps", "12", "--warmup", "3", "--no-extras", "--no-cpu-baseline"]
    a, _ = _bench_line(common)
    assert a["dist_backend"] == "nccl" and a["ranks"] == 1 and len(a["devices"]) == 1 and a["devices"][0]["frames"] == 8, a.get("dist_init_error")
    assert a["allgather_us"] > 0 and a["gathered_metric_rows"] == 8
    assert a["parity"]["ok"] and a["parity"]["worst_frame_vs_b1_rel"] < 1e-4, a["parity"]
    b, _ = _bench_line(common + ["--no-process-group", "--no-parity"])
    assert b["dist_backend"] is None and b["ranks"] is None
    ratio = a["value"] / b["value"]
    print(f"N=1 under a 1-rank RCCL group {a['value']:.1f} frames/s, bare process {b['value']:.1f}: ratio {ratio:.4f}; all-gather {a['allgather_us']:.0f} us")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    import json

    json.dump({"with_group": a, "bare": b, "ratio": ratio}, open(os.path.join(root, "gpurun_out", "bench_n1_rccl_vs_bare.json"), "w"), indent=1)
    assert 0.97 < ratio < 1.03, ratio


def test_zero_cost_volume_through_the_pipeline():
    """`feature_volume_type: zero_cost_volume` (reference modules/cost_volume.py:1307-1384: a volume of zeros, lowest cost = the first plane)
    through HotPath - the one branch of pipeline.py that prepares the volume with torch ops instead of a volume kernel - against the oracle chain
    fed with zeros, and against the module-level ZeroCostVolumeManager."""
    from implicit_depth_amd.cost_volume import ZeroCostVolumeManager
    from implicit_depth_amd.pipeline import HotPath

    B, K, H, W, D, P = 2, 3, 24, 32, 16, 2
    model, inp, pyr, rd = _build(B, K, H, W, D, P)
    zero = HotPath(ZeroCostVolumeManager(H, W, D), model.cost_volume_net, model.depth_decoder, model.binary_mlp).cuda()
    d = {k: v.cuda() for k, v in inp.items()}
    out = zero(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"], rendered_depth=rd.cuda())
    out2 = zero(d["cur_feats"], d["src_feats"], [t.cuda() for t in pyr], d["src_extrinsics"], d["src_poses"], d["src_Ks"], d["cur_invK"], rendered_depth=rd.cuda())
    assert torch.equal(out["pred_0"], out2["pred_0"])  # (the replay re-zeroes the volume buffer)
    sd = lambda m: {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    p64 = [t.double() for t in pyr]
    enc = onet.cv_encoder(torch.zeros(B, D, H, W, dtype=torch.float64), p64[1:], sd(model.cost_volume_net))
    dec = onet.unetpp_decoder([p64[0]] + enc, sd(model.depth_decoder), depth_head=False)
    ref = onet.occlusion_logits(dec["feature_s0_b1hw"], rd.double(), sd(model.binary_mlp), None)
    assert rel_err(out["pred_0"].cpu(), ref) < TOL
    cost, lowest, planes, mask = zero.cost_volume(**dict(d, min_depth=torch.tensor(0.25).view(1, 1, 1, 1).cuda(), max_depth=torch.tensor(5.0).view(1, 1, 1, 1).cuda()))
    assert float(cost.abs().max()) == 0.0 and mask is None and out["overall_mask_bhw"] is None
    assert torch.allclose(out["lowest_cost_bhw"], lowest) and abs(float(lowest.mean()) - 0.25) < 1e-6  # the nearest plane everywhere
