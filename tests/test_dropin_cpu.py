"""Structure-driven conversion (implicit_depth_amd.dropin): channel configs are inferred from
the module being replaced and the state_dict transfers 1:1.  CPU only (no kernels run)."""
import pytest
import torch

import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import cost_volume as cv
from implicit_depth_amd import dropin
from implicit_depth_amd import networks as net


def _same_state(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_conversions_infer_structure_and_copy_weights():
    enc = net.CVEncoder(96, [48, 64, 160, 256], [64, 128, 256, 384])
    syn.fill_state_dict(enc, 1)
    _same_state(enc, dropin.convert_cv_encoder(enc))
    for cls in (net.BDDecoderPP, net.DepthDecoderPP):
        dec = cls([24, 64, 128, 256, 384])
        syn.fill_state_dict(dec, 2)
        new = dropin.convert_decoder(dec)
        assert type(new) is cls
        _same_state(dec, new)
    for cls in (net.SkipDecoder, net.SkipDecoderRegression):
        dec = cls([24, 64, 128, 256, 384])
        syn.fill_state_dict(dec, 5)
        new = dropin.convert_decoder(dec)
        assert type(new) is cls
        _same_state(dec, new)
    for prior in (False, True):
        mlp = net.BinaryMLPNetwork([64, 64, 128, 256], use_prior=prior)
        syn.fill_state_dict(mlp, 3)
        new = dropin.convert_binary_mlp(mlp)
        assert new.use_prior == prior
        _same_state(mlp, new)
    for K in (2, 7):
        fv = cv.FeatureVolumeManager(24, 32, 16, num_source_views=K)
        syn.fill_state_dict(fv.mlp, 4)
        new = dropin.convert_cost_volume(fv)
        assert new.num_source_views == K and new.mlp.net[0].in_features == 26 * K + 20
        _same_state(fv, new)


def test_feature_mlp_column_maps_are_a_permutation_of_the_reference_layout():
    for K in (1, 2, 7):
        vox, pix, pose = cv.feature_mlp_column_maps(K)
        used = [c for c in vox + pix + pose if c >= 0]
        n_in = 16 * (K + 1) + 10 * K + 4
        assert sorted(used) == list(range(n_in)), "every reference MLP input column appears exactly once"
        assert len(vox) == 16 * (K + 4) and len(pix) == 32 and len(pose) == 3 * K
    # fv_mlp_gen_k's layout: eight metadata slots per view group, two blocks per group, every column exactly once
    for K, C in ((9, 16), (12, 16), (16, 16), (7, 32), (3, 32)):
        vox, pix, pose = cv.feature_mlp_column_maps(K, C, layout="gen8")
        used = [c for c in vox + pix + pose if c >= 0]
        assert sorted(used) == list(range(C * (K + 1) + 10 * K + 4)), (K, C)
        J = 2 if K <= 8 else -(-K // 4)
        assert len(vox) == C * K + 16 * 2 * J and len(pix) == C + 16
        plane = C * (K + 1) + 2 * K
        assert vox[C * K:].index(plane) == 16 + 3, "plane depth: group 0, second block, quarter 0, slot 3"
    # fv_mlp_k's layout: the per-view "valid" columns (identically-1 inputs) go to the bias, everything else appears exactly once
    for K in (1, 2, 7, 8):
        vox, pix, pose = cv.feature_mlp_column_maps(K, fold_mask=True)
        used = [c for c in vox + pix + pose + cv.feature_mlp_mask_columns(K) if c >= 0]
        assert sorted(used) == list(range(16 * (K + 1) + 10 * K + 4))
        assert len(vox) == 16 * (K + 4) and len(pix) == 32 and len(pose) == 3 * K
        plane = 16 * (K + 1) + 2 * K
        blk, rem = divmod(vox[16 * K:].index(plane), 16)
        assert (blk, rem // 4, rem % 4) == ((1, 3, 2) if K < 8 else (3, 0, 0)), "plane depth: quarter 3's first view-7 slot, or alone in block 3"
        if K < 8:
            assert all(c < 0 for c in vox[16 * (K + 3):]), "block 3 is empty below 8 views"


def test_inference_mode_parameters_invalidate_weight_caches_on_load_state_dict():
    """Parameters created under torch.inference_mode() have no version counter, yet load_state_dict() / copy_() inside
    inference mode do modify them (test_bd.py builds and loads the model that way): the cache key component must change
    when a drop-in module's state dict is (re)loaded, and on explicit invalidation."""
    import torch

    from implicit_depth_amd import _lib
    from implicit_depth_amd.layers import BasicBlock

    with torch.inference_mode():
        bb = BasicBlock(16, 16)
        w = bb.conv1.weight
        with pytest.raises(RuntimeError):
            w._version
        k0 = _lib.param_version(w)
        assert k0 == _lib.param_version(w)
        bb.load_state_dict({k: v * 2 for k, v in bb.state_dict().items()})
        k1 = _lib.param_version(w)
        assert k1 != k0, "load_state_dict on the module must change the key"
        parent = torch.nn.Sequential(bb)
        parent.load_state_dict(parent.state_dict())
        assert _lib.param_version(w) != k1, "load_state_dict on a parent module must change it too"
        k2 = _lib.param_version(w)
        w.mul_(0.5)  # not observable: documented to need the explicit call
        _lib.invalidate_weight_caches()
        assert _lib.param_version(w) != k2
    # ordinary parameters keep using torch's counter
    bb2 = BasicBlock(16, 16)
    v0 = _lib.param_version(bb2.conv1.weight)
    with torch.no_grad():
        bb2.conv1.weight.mul_(0.5)
    assert _lib.param_version(bb2.conv1.weight) == v0 + 1


def test_plan_cache_is_a_small_lru():
    from implicit_depth_amd import nhwc

    c = nhwc.PlanCache(entries=3)
    for k in "abc":
        c.put(k, k.upper())
    assert c.get("a") == "A"          # refreshes "a"
    c.put("d", "D")                    # evicts the least recently used: "b"
    assert c.get("b") is None and c.get("a") == "A" and c.get("c") == "C" and c.get("d") == "D" and len(c) == 3
    assert isinstance(nhwc.build_flags(), tuple) and nhwc.WINOGRAD in nhwc.build_flags()


def test_plan_cache_drops_entries_with_stale_parameters():
    """the LRU is for shapes only: a key that differs from a cached one just in its ParamKey parts replaces it at once (a checkpoint
    load after a warm-up forward must not keep the old plan's activation buffers alive)"""
    from implicit_depth_amd import nhwc

    c = nhwc.PlanCache(entries=4)
    pk = lambda v: nhwc.ParamKey(("fp32", ("flags",), (1234, v)))
    c.put(("bb", (1, 16, 8, 8), "cuda:0", pk(0)), "plan v0")
    c.put(("bb", (2, 16, 8, 8), "cuda:0", pk(0)), "other shape")
    c.put(("bb", (1, 16, 8, 8), "cuda:0", pk(1)), "plan v1")  # same shape, new parameter version
    assert len(c) == 2 and c.get(("bb", (1, 16, 8, 8), "cuda:0", pk(0))) is None
    assert c.get(("bb", (1, 16, 8, 8), "cuda:0", pk(1))) == "plan v1" and c.get(("bb", (2, 16, 8, 8), "cuda:0", pk(0))) == "other shape"


def test_state_dict_load_into_an_unwatched_child_invalidates_and_modules_pickle():
    """load_state_dict straight into a submodule (cost_volume.mlp, a Conv2d inside a decoder's ModuleDict) under inference_mode bumps the
    weights epoch, and the hook is a module-level function: the module still pickles"""
    import io
    import pickle

    from implicit_depth_amd import _lib
    from implicit_depth_amd import networks as net

    with torch.inference_mode():
        dec = net.BDDecoderPP([24, 64, 128, 256, 384])
    _lib.watch_state_dict_loads(dec)
    child = next(m for m in dec.modules() if isinstance(m, torch.nn.Conv2d))
    e0 = _lib._weights_epoch
    with torch.inference_mode():
        child.load_state_dict({k: v * 0.5 for k, v in child.state_dict().items()})
    assert _lib._weights_epoch == e0 + 1
    pickle.load(io.BytesIO(pickle.dumps(dec)))


def test_winograd_kernel_selection_rules():
    """nhwc.wino4_eligible / wino_eligible (host logic, no kernel runs): F(4x4) takes the plain 3x3 stride-1 zero-padded layers with
    Cout % 64 == 0, > 16 input channels, LeakyReLU / ELU / no activation and >= WINO4_MIN_TILES tiles of 32 x 8 pixels x 64 channels that fill
    the map, with or without a fused 1x1 projection; narrow outputs and small grids are left to F(2x2) / the direct kernels
    (the thresholds are measured: profiles/r04/experiments.md 1b)."""
    from torch import nn

    from implicit_depth_amd import nhwc

    class V:  # what the rules read of a view
        pass

    c64, c192, c16 = nn.Conv2d(64, 64, 3, 1, 1), nn.Conv2d(192, 64, 3, 1, 1), nn.Conv2d(16, 64, 3, 1, 1)
    proj, s2, c32 = nn.Conv2d(32, 64, 1), nn.Conv2d(64, 64, 3, 2, 1), nn.Conv2d(64, 32, 3, 1, 1)
    Z, L, E = nhwc.PAD_ZEROS, nhwc.ACT_LRELU, 2
    assert nhwc.WINO4_MIN_TILES == 768 and nhwc.WINOGRAD4
    # bench batch: every level down to 48x64 (32 * 6 * 2 * 2 = 768 tiles at 128 channels); the 24x32 level (384 tile groups) stays with the grouped F(2x2) launch
    assert nhwc.wino4_eligible([(V(), c64)], 64, 32, 192, 256, Z, L)
    assert nhwc.wino4_eligible([(V(), c192)], 64, 32, 96, 128, Z, nhwc.ACT_NONE)
    assert nhwc.wino4_eligible([(V(), nn.Conv2d(128, 128, 3, 1, 1))], 128, 32, 48, 64, Z, L)
    assert not nhwc.wino4_eligible([(V(), nn.Conv2d(256, 256, 3, 1, 1))], 256, 32, 24, 32, Z, L)
    # tile count: B = 16 @96x128 has 16 * 12 * 4 = 768, B = 8 has 384; one frame never qualifies
    assert nhwc.wino4_eligible([(V(), c64)], 64, 16, 96, 128, Z, L) and not nhwc.wino4_eligible([(V(), c64)], 64, 8, 96, 128, Z, L)
    assert not nhwc.wino4_eligible([(V(), c64)], 64, 1, 192, 256, Z, L)
    # shape family
    assert nhwc.wino4_eligible([(V(), c64), (V(), proj)], 64, 32, 192, 256, Z, L), "fused 1x1 projection: conv3x3_wino4_k<true>"
    assert not nhwc.wino4_eligible([(V(), c64), (V(), s2)], 64, 32, 192, 256, Z, L), "a 3x3 second source"
    assert nhwc.wino4_eligible([(V(), c64)], 64, 32, 192, 256, Z, E), "ELU (ConvBlock) is in the epilogue"
    assert not nhwc.wino4_eligible([(V(), c64)], 64, 32, 192, 256, Z, 7), "an unknown activation code"
    assert not nhwc.wino4_eligible([(V(), c16)], 64, 32, 192, 256, Z, L), "<= 16 input channels"
    assert not nhwc.wino4_eligible([(V(), c32)], 32, 32, 192, 256, Z, L), "Cout % 64"
    assert not nhwc.wino4_eligible([(V(), s2)], 64, 32, 96, 128, Z, L), "stride 2"
    assert not nhwc.wino4_eligible([(V(), c64)], 64, 32, 192, 256, nhwc.PAD_ZEROS + 1, L), "replicate padding"
    # tile fill: a 12x16 map covers 12 * 16 of 16 * 32 tile pixels
    assert not nhwc.wino4_eligible([(V(), nn.Conv2d(384, 384, 3, 1, 1))], 384, 64, 12, 16, Z, L)
    # a 1080p x 192-channel image exceeds the kernel's 1 GiB-per-image bound (32-bit halo offsets): the plan must not pick it
    big = V()
    big.H, big.W, big.cs = 1088, 1920, 192
    assert not nhwc.wino4_eligible([(big, c192)], 64, 4, 1088, 1920, Z, L)
    ok = V()
    ok.H, ok.W, ok.cs = 192, 256, 192
    assert nhwc.wino4_eligible([(ok, c192)], 64, 32, 192, 256, Z, L)
    # what F(4x4) leaves goes to F(2x2) where that one's own rules hold
    assert nhwc.wino_eligible([(V(), c64), (V(), proj)], 64, 32, 192, 256, Z)
    assert nhwc.wino_eligible([(V(), c64)], 64, 4, 96, 128, Z) and not nhwc.wino_eligible([(V(), c64)], 64, 1, 24, 32, Z)


def test_plan_buffer_liveness_reuse_rules():
    """Plan.release / Plan.buffer (host logic): a released whole buffer of >= REUSE_MIN_BYTES is handed out again for the same shape, once;
    small buffers (whose ops need their dependency level for grouped launches), channel-padded buffers (zero padding written by nobody),
    slices and foreign tensors are not pooled.  The scheduler orders the new writer after the old readers through the shared tensor id."""
    import torch

    from implicit_depth_amd import nhwc

    p = nhwc.Plan(torch.device("cpu"))
    old = nhwc.REUSE_MIN_BYTES
    nhwc.REUSE_MIN_BYTES = 1 << 16
    try:
        big = p.buffer(2, 32, 32, 16)  # 128 KiB
        p.release(big)
        again = p.buffer(2, 32, 32, 16)
        assert again.buf is big.buf, "same shape: recycled"
        assert p.buffer(2, 32, 32, 16).buf is not big.buf, "... once"
        small = p.buffer(1, 8, 8, 16)  # 4 KiB
        p.release(small)
        assert p.buffer(1, 8, 8, 16).buf is not small.buf
        padded = p.buffer(2, 32, 32, 24)  # 24 -> 32 channels with zero padding
        p.release(padded)
        assert p.buffer(2, 32, 32, 24).buf is not padded.buf
        whole = p.buffer(2, 32, 32, 32)
        p.release(whole.slice(0, 16))
        assert p.buffer(2, 32, 32, 32).buf is not whole.buf, "a slice does not release its buffer"
        foreign = nhwc.View(torch.empty(2, 32, 32, 16), 0, 16)
        p.release(foreign)
        assert p.buffer(2, 32, 32, 16).buf is not foreign.buf
        # regions are keyed by the tensor: a recycled buffer carries its readers' dependencies to the next writer
        assert nhwc._overlap([nhwc._region(big)], [nhwc._region(again)])
        # a buffer still waiting in the pool cannot be released again (it would be handed to two later allocations); once it has been taken
        # over, its new owner may release it
        import pytest

        from implicit_depth_amd import _lib

        twice = p.buffer(2, 16, 16, 64)
        p.release(twice)
        with pytest.raises(_lib.IdhError):
            p.release(twice)
        taken = p.buffer(2, 16, 16, 64)
        assert taken.buf is twice.buf
        p.release(taken)
        assert (p.recycled, p.recycled_candidates) == (2, 3)
    finally:
        nhwc.REUSE_MIN_BYTES = old
