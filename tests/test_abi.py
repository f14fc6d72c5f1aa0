"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/idh.h
declares (no compute calls — there is no GPU here), and argument validation paths that do
not touch the device behave."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    txt = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include"))) if f.endswith(".h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(idh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from implicit_depth_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/idh.h but not exported by libidh.so"
    # the ctypes table binds exactly the header's symbols
    assert sorted(_lib.declared_symbols()) == syms


def test_error_strings_and_host_only_entry_points():
    from implicit_depth_amd import _lib

    L = _lib.lib()
    assert L.idh_version() >= 104
    assert L.idh_error_string(0) == b"ok"
    assert b"workspace" in L.idh_error_string(-4)
    # argument validation happens before any launch: safe without a GPU
    assert L.idh_cost_volume_dot_fwd(None, None, None, None, None, 0.25, 5.0, 1, 2, 16, 8, 8, 4, None, 0, None, None, None) == -1
    assert L.idh_cost_volume_dot_fwd(None, None, None, None, None, 0.25, 5.0, 1, 2, 8, 8, 8, 4, None, 0, None, None, None) == -2


def test_product_path_refuses_cpu_tensors():
    import torch

    import implicit_depth_amd.synthetic as syn
    from implicit_depth_amd import _lib
    from implicit_depth_amd.cost_volume import CostVolumeManager

    inp = syn.cost_volume_inputs(1, 2, 16, 8, 8, 0)
    with pytest.raises(_lib.IdhError):
        CostVolumeManager(8, 8, 4)(**inp)


def test_op_descriptor_layout_matches_library():
    import ctypes

    from implicit_depth_amd import _lib, nhwc

    assert ctypes.sizeof(nhwc.Op) == _lib.lib().idh_sizeof_op()
    assert ctypes.sizeof(_lib.VolumeOpts) == _lib.lib().idh_sizeof_volume_opts() == _lib.VolumeOpts().struct_size
    assert _lib.lib().idh_packed_weight_floats(40, 24, 3) == 9 * 32 * 48
    assert _lib.lib().idh_run_ops(None, 0, None) == 0


def _conv_op(nhwc, N, H, W, cin, cout, tile_m, tile_n, group, split_k=1, stride=1):
    op = nhwc.Op()
    op.kind, op.N = nhwc.OP_CONV, N
    s = op.src[0]
    s.in_, s.w, s.cs, s.H, s.W, s.Cin, s.ks, s.stride = 0x1000, 0x2000, cin, H, W, cin, 3, stride
    op.out, op.out_cs, op.Ho, op.Wo, op.Cout = 0x3000, cout, H // stride, W // stride, cout
    op.ws = 0x4000 if split_k > 1 else None
    op.split_k, op.tile_m, op.tile_n, op.group = split_k, tile_m, tile_n, group
    return op


def test_launch_grouping_decisions_without_a_gpu():
    """idh_count_launches replays idh_run_ops' decisions (include/idh_ops.h, idh_op.group): ops of one dependency level
    share a launch when they are equal-tile 4-row LDS convs (any size) or a mixed run of small members."""
    import ctypes as C

    from implicit_depth_amd import _lib, nhwc

    L = _lib.lib()

    def count(ops):
        arr = (nhwc.Op * len(ops))(*ops)
        return L.idh_count_launches(C.cast(arr, C.c_void_p), len(ops))

    up = nhwc.Op()
    up.kind, up.N, up.group = nhwc.OP_UPSAMPLE2, 1, 5
    up.src[0].in_, up.src[0].cs, up.src[0].H, up.src[0].W, up.src[0].Cin = 0x1000, 64, 48, 64, 64
    up.out, up.out_cs = 0x3000, 64
    lds64 = lambda N, g: _conv_op(nhwc, N, 96, 128, 64, 64, 9, 0, g)
    lds32 = lambda N, g: _conv_op(nhwc, N, 96, 128, 64, 64, 9, 2, g)
    down = lambda N, g, sk=1: _conv_op(nhwc, N, 96, 128, 64, 128, 1, 4, g, split_k=sk, stride=2)
    # one frame: four kinds of work of one level -> one level_k launch; without the group id one launch each
    assert count([lds64(1, 5), lds32(1, 5), down(1, 5), up]) == 1
    assert count([lds64(1, 0), lds32(1, 0), down(1, 0)]) == 3
    # a split-K member adds the (grouped) reduce launch
    assert count([lds64(1, 5), down(1, 5, sk=3), up]) == 2
    # different levels never merge
    assert count([lds64(1, 5), lds32(1, 6)]) == 2
    # 32 frames: members fill the chip on their own -> equal-tile LDS convs still share a grid, the rest run alone
    assert count([lds64(32, 5), lds64(32, 5), lds32(32, 5), down(32, 5)]) == 3
    # Winograd convs of one level share a persistent grid (conv3x3_wino_group_k): plain ones together, those with a fused 1x1
    # source together, at most six per grid; without the level's group id one launch each
    wino = lambda N, g: _conv_op(nhwc, N, 96, 128, 64, 64, nhwc.TILE_WINO, 0, g)

    def wino2(N, g):
        op = wino(N, g)
        s1 = op.src[1]
        s1.in_, s1.w, s1.cs, s1.H, s1.W, s1.Cin, s1.ks, s1.stride = 0x5000, 0x6000, 128, 96, 128, 128, 1, 1
        return op

    assert count([wino(4, 7), wino(4, 7), wino(4, 7)]) == 1
    assert count([wino(4, 0), wino(4, 0)]) == 2
    assert count([wino(4, 7), wino(4, 7), wino2(4, 7), wino2(4, 7)]) == 2
    assert count([wino(4, 7)] * 7) == 2
    assert count([wino(4, 7), lds64(4, 7), lds64(4, 7)]) == 2
    # an image too large for the LDS kernels' 32-bit byte offsets (H*W*cs*4 >= 2 GiB) falls back to the direct kernel
    assert count([_conv_op(nhwc, 1, 16384, 16384, 64, 64, 8, 0, 0)]) == 1
    # validation still applies in the dry run
    bad = lds64(1, 0)
    bad.out = None
    assert count([bad]) == -1


def test_hot_kernels_compile_without_scratch():
    """The stage loops of the F(4x4) conv kernel and the plane loop of the tuned feature-volume kernel must stay spill-free: a scratch reload in front
    of a batch of loads waits for every load in flight (profiles/r04-r05/experiments.md).  Compiles the two sources for gfx950 with
    -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py; no GPU needed) and reads the ScratchSize of each instance."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import kernel_resources as kr

    csrc = os.path.join(root, "implicit-depth_amd", "csrc")
    rows = {r["Name"]: int(r.get("ScratchSize [bytes/lane]", 0)) for src in ("conv_wino4.hip", "feature_volume.hip") for r in kr.resources(os.path.join(csrc, src))}
    for name in ("conv3x3_wino4_k<true, false>", "conv3x3_wino4_k<false, true>", "conv3x3_wino4_k<false, false>", "fv_mlp_k<7>", "fv_mlp_k<0>"):
        assert name in rows, (name, sorted(rows))
        assert rows[name] == 0, (name, rows[name])


def test_network_entry_point_size_queries_run_without_a_gpu():
    """include/idh_net.h: the *_sizes queries are host-only (plan building + idh_count_launches).  At the bench shapes the UNet++ plan of
    a 32-frame batch must hold the F(4x4) layers and recycle activation buffers; bad arguments come back as error codes."""
    import ctypes as C

    from implicit_depth_amd import _lib
    from implicit_depth_amd import net_abi as na
    from implicit_depth_amd import networks as net

    L = _lib.lib()
    keep = []
    cve = net.CVEncoder(64, [48, 64, 160, 256], [64, 128, 256, 384])
    dec = net.DepthDecoderPP([24] + cve.num_ch_enc)
    blocks, heads = na.unetpp_blocks(dec, keep)
    assert heads is not None and len(blocks) == na.UNETPP_BLOCKS
    feats = na.tensors([na.nchw(None, 24, 192, 256)] + [na.nhwc(None, c, 96 >> i, 128 >> i) for i, c in enumerate([64, 128, 256, 384])])
    fouts = na.tensors([na.nhwc(None, c, 192 >> i, 256 >> i) for i, c in enumerate([64, 64, 128, 256])])
    sz = {}
    for N in (1, 32):
        s = na.NetSizes()
        assert L.idh_unetpp_sizes(blocks, na.UNETPP_BLOCKS, heads, N, feats, fouts, C.byref(s)) == 0
        sz[N] = s.as_dict()
    assert sz[1]["wino4"] == 0 and sz[32]["wino4"] >= 60 and sz[32]["recycled"] >= 8 and sz[32]["launches"] > 0 and sz[1]["ops"] == sz[32]["ops"]
    assert sz[32]["workspace_floats"] > sz[1]["workspace_floats"] > 0 and sz[32]["weight_floats"] > 0
    s = na.NetSizes()
    assert L.idh_unetpp_sizes(blocks, 48, heads, 1, feats, fouts, C.byref(s)) == -1  # wrong block count
    bad = na.tensors([na.nchw(None, 24, 192, 256)] + [na.nhwc(None, c, 96 >> i, 100) for i, c in enumerate([64, 128, 256, 384])])
    assert L.idh_unetpp_sizes(blocks, na.UNETPP_BLOCKS, heads, 1, bad, fouts, C.byref(s)) == -1  # pyramid levels must halve
    cb = na.cvencoder_blocks(cve, keep)
    cost = na.nhwc(None, 64, 96, 128)
    img = na.tensors([na.nchw(None, c, 96 >> i, 128 >> i) for i, c in enumerate([48, 64, 160, 256])])
    outs = na.tensors([na.nhwc(None, c, 96 >> i, 128 >> i) for i, c in enumerate([64, 128, 256, 384])])
    assert L.idh_cvencoder_sizes(cb, 4, 32, C.byref(cost), img, outs, C.byref(s)) == 0 and s.ops == 28 and s.wino4 > 0


def test_cpp_plan_builder_mirrors_the_python_thresholds():
    """csrc/networks.hip (the C++ twin of nhwc.Plan behind idh_basic_block_fwd / idh_cvencoder_fwd / idh_unetpp_fwd) carries the kernel-selection
    thresholds as constants: they must equal nhwc.py's defaults, or the two builders stop producing the same op lists."""
    from implicit_depth_amd import nhwc

    src = open(os.path.join(ROOT, "implicit-depth_amd", "csrc", "networks.hip")).read()

    def const(name):
        m = re.search(rf"\b{name}\s*=\s*([0-9.]+(?:ll)?(?:\s*<<\s*[0-9]+)?)", src)
        assert m, name
        return eval(m.group(1).replace("ll", ""))

    assert const("kWinoMinTiles") == nhwc.WINO_MIN_TILES and const("kWinoMinFill") == nhwc.WINO_MIN_FILL
    assert const("kWino4MinTiles") == nhwc.WINO4_MIN_TILES and const("kWino4MinFill") == nhwc.WINO4_MIN_FILL
    assert const("kReuseMinBytes") == nhwc.REUSE_MIN_BYTES and const("kNarrowTileBelow") == nhwc.NARROW_TILE_BELOW
    assert const("kSplitMinChunks") == nhwc.SPLIT_MIN_CHUNKS and const("kSplitMax") == nhwc.SPLIT_MAX
    assert const("kProjChunkWeight") == nhwc.PROJ_CHUNK_WEIGHT and const("kS2FirstMinBlocks") == nhwc.S2_FIRST_MIN_BLOCKS
    assert const("kTargetWaves") == nhwc.TARGET_WAVES and const("kMinWaves") == nhwc.MIN_WAVES
    # switches the C++ builder assumes at their defaults
    assert (nhwc.WINOGRAD, nhwc.WINOGRAD4, nhwc.WINOGRAD4_PROJ, nhwc.BUFFER_REUSE, nhwc.S2_FIRST, nhwc.MERGE_LEVELS, nhwc.WINO_GROUP) == (True,) * 7
    assert (nhwc.FUSE_UPSAMPLE, nhwc.PROJ_LOWRES, nhwc.NARROWEST_TILE_BELOW, nhwc.DEFAULT_MATH) == (False, False, 0, "fp32")
