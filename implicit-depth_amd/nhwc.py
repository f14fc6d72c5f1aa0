"""Host-side execution plans for the conv stage of the hot path.

A *plan* is a flat array of ``idh_op`` descriptors (include/idh_ops.h) plus the NHWC
activation buffers they point into, built once per (module, input shape, device) and replayed
with a single C-ABI call (``idh_run_ops``).  torch only owns the memory and the stream.

What the plans encode (reference modules/networks.py + modules/layers.py):
  BasicBlock   -> conv1(+LeakyReLU)  then ONE launch for conv2 + residual: the 1x1 / strided-3x3
                  projection of the block input is a second K-source of that launch, the identity
                  residual is added in the epilogue
  torch.cat    -> producers write into channel slices of the consumer's input buffer
  upsample     -> bilinear x2 kernel writing straight into that slice
  heads        -> only the ``output_i`` results that survive in the reference's dict are
                  computed (the reference evaluates and discards the others, networks.py:80)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _lib

# --- ctypes mirrors of include/idh_ops.h -------------------------------------------------
OP_CONV, OP_UPSAMPLE2, OP_IMPORT, OP_EXPORT, OP_SPLITK, OP_HEAD, OP_INSTNORM, OP_UPSAMPLE2_NEAREST, OP_COPY, OP_POINTWISE_NCHW, OP_POINTWISE_UP = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
ACT_NONE, ACT_LRELU, ACT_ELU = 0, 1, 2
PAD_ZEROS, PAD_REPLICATE = 0, 1


class ConvSrc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("w", C.c_void_p), ("cs", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cin", C.c_int32), ("ks", C.c_int32), ("stride", C.c_int32), ("pad_mode", C.c_int32), ("_r", C.c_int32),
                ("up_in", C.c_void_p * 2), ("up_cs", C.c_int32 * 2), ("up_c0", C.c_int32), ("up_C", C.c_int32),
                ("norm", C.c_void_p), ("norm_slope", C.c_float), ("norm_act", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("N", C.c_int32), ("src", ConvSrc * 2), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("out", C.c_void_p), ("ws", C.c_void_p), ("res_cs", C.c_int32), ("out_cs", C.c_int32), ("Ho", C.c_int32),
                ("Wo", C.c_int32), ("Cout", C.c_int32), ("act", C.c_int32), ("slope", C.c_float), ("split_k", C.c_int32),
                ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("group", C.c_int32)]


def _bind():
    return _lib.lib()


# --- NHWC views ---------------------------------------------------------------------------
@dataclass
class View:
    """Channel slice [c0, c0+C) of a dense NHWC buffer (N,H,W,CS)."""

    buf: torch.Tensor
    c0: int
    C: int

    @property
    def N(self):
        return self.buf.shape[0]

    @property
    def H(self):
        return self.buf.shape[1]

    @property
    def W(self):
        return self.buf.shape[2]

    @property
    def cs(self):
        return self.buf.shape[3]

    @property
    def ptr(self):
        return self.buf.data_ptr() + 4 * self.c0

    def slice(self, c0, C):
        return View(self.buf, self.c0 + c0, C)

    def dense(self) -> torch.Tensor:
        return self.buf[..., self.c0 : self.c0 + self.C]


@dataclass
class CatView:
    """``torch.cat([direct, upsample(ups[0]), upsample(ups[1])], 1)`` that is never materialised: the consumer conv
    reads ``direct`` as it is and interpolates the x2 bilinear upsampling of the low-resolution ``ups`` maps while it
    stages its halo (idh_conv_src.up_*; reference networks.py:64-76 + generic_utils.py:94-103)."""

    direct: View
    ups: List[View]

    @property
    def N(self):
        return self.direct.N

    @property
    def H(self):
        return self.direct.H

    @property
    def W(self):
        return self.direct.W

    @property
    def C(self):
        return self.direct.C + sum(u.C for u in self.ups)


def ceil16(v: int) -> int:
    return (v + 15) & ~15


def _region(v: "View", pad16: bool = False):
    """(buffer id, first channel, one-past-last channel) touched through view ``v``."""
    return (v.buf.data_ptr(), v.c0, v.c0 + (ceil16(v.C) if pad16 else v.C))


def _overlap(ra, rb) -> bool:
    for (ba, a0, a1) in ra:
        for (bb, b0, b1) in rb:
            if ba == bb and a0 < b1 and b0 < a1:
                return True
    return False


def packed_weight(conv: nn.Conv2d) -> torch.Tensor:
    """[tap][ci/4][co][ci%4] copy of a Conv2d weight, cached on the module and refreshed when
    the parameter is modified in place or moved."""
    w = conv.weight
    key = (w.data_ptr(), _lib.param_version(w), str(w.device))
    cached = getattr(conv, "_idh_packed", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    _lib.require_cuda_f32(w)
    L = _bind()
    co, ci, kh, kw = w.shape
    if kh != kw or kh not in (1, 3):
        raise _lib.IdhError(f"conv kernel {kh}x{kw} not covered by the gfx950 conv kernel (1x1 / 3x3 only)")
    n = L.idh_packed_weight_floats(co, ci, kh)
    dst = torch.empty(n, device=w.device, dtype=torch.float32)
    wc = w.detach().contiguous()
    _lib.check(L.idh_pack_conv_weight(wc.data_ptr(), dst.data_ptr(), co, ci, kh, _lib.stream_ptr()), "idh_pack_conv_weight")
    conv._idh_packed = (key, dst)
    return dst


def packed_wino_weight(conv: nn.Conv2d) -> torch.Tensor:
    """Winograd-domain copy U = G g G^T of a 3x3 Conv2d weight in the MFMA A-fragment order of csrc/conv_wino.hip
    (idh_pack_conv_weight_wino); cached like ``packed_weight``."""
    w = conv.weight
    key = (w.data_ptr(), _lib.param_version(w), str(w.device))
    cached = getattr(conv, "_idh_packed_wino", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    _lib.require_cuda_f32(w)
    L = _bind()
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise _lib.IdhError("the Winograd F(2x2,3x3) kernel covers 3x3 convolutions only")
    dst = torch.empty(L.idh_packed_wino_weight_floats(co, ci), device=w.device, dtype=torch.float32)
    wc = w.detach().contiguous()
    _lib.check(L.idh_pack_conv_weight_wino(wc.data_ptr(), dst.data_ptr(), co, ci, _lib.stream_ptr()), "idh_pack_conv_weight_wino")
    conv._idh_packed_wino = (key, dst)
    return dst


def packed_wino4_weight(conv: nn.Conv2d) -> torch.Tensor:
    """Winograd F(4x4,3x3)-domain copy of a 3x3 Conv2d weight in the A-fragment order of csrc/conv_wino4.hip
    (idh_pack_conv_weight_wino4); cached like ``packed_weight``."""
    w = conv.weight
    key = (w.data_ptr(), _lib.param_version(w), str(w.device))
    cached = getattr(conv, "_idh_packed_wino4", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    _lib.require_cuda_f32(w)
    L = _bind()
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise _lib.IdhError("the Winograd F(4x4,3x3) kernel covers 3x3 convolutions only")
    dst = torch.empty(L.idh_packed_wino4_weight_floats(co, ci), device=w.device, dtype=torch.float32)
    wc = w.detach().contiguous()
    _lib.check(L.idh_pack_conv_weight_wino4(wc.data_ptr(), dst.data_ptr(), co, ci, _lib.stream_ptr()), "idh_pack_conv_weight_wino4")
    conv._idh_packed_wino4 = (key, dst)
    return dst


TILE_WINO = 12  # IDH_TILE_WINO of include/idh_ops.h == the op's tile_m
TILE_WINO4 = 13  # IDH_TILE_WINO4
# Winograd F(2x2,3x3) for the eligible 3x3 stride-1 layers of fp32 plans (csrc/conv_wino.hip): fp32 operands and
# accumulation, 2.25x fewer MFMAs; measured 1.67-1.84x over the direct LDS kernel at B=32 (tools/perf_wino.py).
# Tiles are 32 x 8 pixels x 32 channels; WINO_MIN_TILES: below this many tiles the direct kernels' finer tiles fill the
# chip better (conv stage of the hot path with thresholds off / 512 / 256 / 128 / 64, tools/perf_levels.py: B=1 2.67 / 2.66 /
# 2.46 / 2.45 / 2.66 ms, B=2 4.43 / 3.90 / 3.91 / 3.80 / 3.89, B=4 7.75 / 6.64 / 6.36 / 6.26 / 6.47); WINO_MIN_FILL: smallest useful fraction of the tile grid that lies inside the map.
WINOGRAD = True
WINO_MIN_TILES = 128
WINO_MIN_FILL = 0.74
# the Winograd convs of one dependency level as ONE persistent grid (conv3x3_wino_group_k): bit-identical, fewer launch tails
WINO_GROUP = True


def wino_eligible(srcs, cout: int, N: int, Ho: int, Wo: int, pad_mode: int) -> bool:
    (v0, c0) = srcs[0]
    if c0.kernel_size[0] != 3 or c0.stride[0] != 1 or pad_mode != PAD_ZEROS or cout % 32 or isinstance(v0, CatView):
        return False
    if len(srcs) > 1:  # fused second source: BasicBlock's 1x1 stride-1 projection (accumulated in the output domain)
        (v1, c1) = srcs[1]
        if c1.kernel_size[0] != 1 or c1.stride[0] != 1 or isinstance(v1, CatView):
            return False
    ty, tx = -(-Ho // 8), -(-Wo // 32)
    if Ho * Wo < WINO_MIN_FILL * (ty * 8) * (tx * 32):
        return False
    return N * ty * tx * (cout // 32) >= WINO_MIN_TILES


# Winograd F(4x4,3x3) (csrc/conv_wino4.hip, conv3x3_wino4_k) for the plain 3x3 layers: 1.78x fewer MFMAs than F(2x2); 32 x 8 pixel x 64
# channel tiles on two persistent workgroups per CU (512 slots).  Measured against F(2x2) (tools/perf_wino4.py, profiles/r04/perf_wino4_final.txt):
# 1.11-1.39x at B = 32 on every eligible layer of the network.  Layer by layer it wins or ties down to ~192 tiles and loses below
# (96 tiles: 0.6-0.8x), but inside a whole step the threshold that never loses is 384 (bench.py --batch 1 / 4 / 8 / 16 with 192:
# -5 / -4 / 0 / +1 %; with 384: 0 / +2 / +6 / +7 %; profiles/r04/experiments.md): below it the F(2x2) convs of a UNet++ level share ONE
# grouped launch, which a half-filled grid of F(4x4) tiles does not beat.  A fused 1x1 projection rides in the same kernel (conv3x3_wino4_k<true>:
# 1.07-1.25x over F(2x2)'s); layers with a normalised source stay where they were.
# activation-buffer liveness reuse (Plan.release): BasicBlock intermediates of >= REUSE_MIN_BYTES are recycled by later blocks of the same shape
BUFFER_REUSE = True
REUSE_MIN_BYTES = 64 << 20
WINOGRAD4 = True
WINOGRAD4_PROJ = True  # ... also for the blocks with a fused 1x1 projection (conv3x3_wino4_k<true>)
WINO4_MIN_TILES = 768  # (re-swept late in round 5, tools/perf_levels.py: 384 / 768 -> conv plan 8.73 / 8.64 ms at B = 8, 12.88 / 12.67 at B = 12, equal at 4, 16, 24, 32: a
# level of exactly 384 tile groups - 0.75 of a round of the 512 resident workgroups - runs no faster than the grouped F(2x2) launch of the level)
WINO4_MIN_FILL = 0.85


def wino4_eligible(srcs, cout: int, N: int, Ho: int, Wo: int, pad_mode: int, act: int, out=None, res=None, slope: float = 0.2) -> bool:
    """Mirror of ``idh_conv::wino4_supported`` (csrc/conv_wino4.hip) plus the fill / tile-count rules.  ``out`` / ``res``: the views the op
    writes / adds - their per-image byte sizes (with the channel stride of a wider concat buffer) are 32-bit buffer ranges in the kernel, so a
    layer that exceeds them is planned onto F(2x2) / the direct kernels here instead of failing at run time with IDH_EUNSUPPORTED."""
    (v0, c0) = srcs[0]
    if c0.kernel_size[0] != 3 or c0.stride[0] != 1 or pad_mode != PAD_ZEROS or cout % 64 or isinstance(v0, CatView):
        return False
    if len(srcs) > 1:  # fused second source: BasicBlock's 1x1 stride-1 projection (accumulated in the pixel domain after the output transform)
        (v1, c1) = srcs[1]
        if not WINOGRAD4_PROJ or len(srcs) > 2 or c1.kernel_size[0] != 1 or c1.stride[0] != 1 or isinstance(v1, CatView):
            return False
    if act not in (ACT_NONE, ACT_LRELU, ACT_ELU) or (act == ACT_LRELU and not 0.0 <= slope <= 1.0) or c0.in_channels <= 16:  # (<= 16: the copy pipeline runs a pair of 8-channel stages ahead)
        return False
    if getattr(v0, "H", 0) * getattr(v0, "W", 0) * getattr(v0, "cs", 0) * 4 >= 1 << 30:  # (csrc: the halo's 32-bit offsets run a few rows past an image)
        return False
    for v in (out, res):  # output / residual image: 32-bit byte offsets
        if v is not None and Ho * Wo * getattr(v, "cs", 0) * 4 >= 1 << 31:
            return False
    if len(srcs) > 1:
        (v1, c1) = srcs[1]
        if getattr(v1, "H", 0) * getattr(v1, "W", 0) * getattr(v1, "cs", 0) * 4 >= 1 << 31 or ((c1.in_channels + 15) // 16) * 4 * (((cout + 15) // 16) * 16) * 64 >= 1 << 31:
            return False
    if ((c0.in_channels + 15) // 16) * 4 * (((cout + 15) // 16) * 16) * 36 * 16 * 4 >= 1 << 31:  # packed weights
        return False
    ty, tx = -(-Ho // 8), -(-Wo // 32)
    if Ho * Wo < WINO4_MIN_FILL * (ty * 8) * (tx * 32):
        return False
    return N * ty * tx * (cout // 64) >= WINO4_MIN_TILES


SPLIT_CODE = {"f16x3": 11}  # IDH_SPLIT_F16X3 of include/idh_ops.h == the op's tile_m


def split_packed_weight(conv: nn.Conv2d, math: str, proj: Optional[nn.Conv2d] = None) -> torch.Tensor:
    """16-bit-piece copy of a 3x3 Conv2d weight (and, fused behind it, of the 1x1 projection that shares
    its accumulator) in the LDS image order of csrc/conv_split.hip (idh_pack_conv_weight_split);
    cached like ``packed_weight``."""
    w = conv.weight
    w1 = proj.weight if proj is not None else None
    key = (w.data_ptr(), _lib.param_version(w), str(w.device), math) + ((w1.data_ptr(), _lib.param_version(w1)) if w1 is not None else ())
    cached = getattr(conv, "_idh_packed_split", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    _lib.require_cuda_f32(w)
    L = _bind()
    co, ci, kh, kw = w.shape
    ci1 = 0
    w1c = None
    if w1 is not None:
        _lib.require_cuda_f32(w1)
        if tuple(w1.shape[2:]) != (1, 1) or w1.shape[0] != co:
            raise _lib.IdhError("the fused second source of a split-precision conv must be a 1x1 conv with the same Cout")
        ci1 = w1.shape[1]
        w1c = w1.detach().reshape(co, ci1).contiguous()
    n = L.idh_packed_split_weight_bytes(co, ci, ci1, SPLIT_CODE[math])
    if kh != 3 or kw != 3 or n == 0:
        raise _lib.IdhError("split-precision conv covers 3x3 kernels with Cout % 64 == 0")
    dst = torch.empty(n // 4, device=w.device, dtype=torch.int32)
    wc = w.detach().contiguous()
    _lib.check(L.idh_pack_conv_weight_split(wc.data_ptr(), _lib.ptr(w1c), dst.data_ptr(), co, ci, ci1, SPLIT_CODE[math], _lib.stream_ptr()),
               "idh_pack_conv_weight_split")
    conv._idh_packed_split = (key, dst)
    return dst


# Arithmetic of the 3x3 stride-1 convs: "fp32" = v_mfma_f32_16x16x4_f32 everywhere (default);
# "f16x3" (opt-in) = the layers of the split_eligible() family run on the f16 matrix cores with every fp32 operand
# expanded into 2 scaled f16 pieces and the 3 significant cross products accumulated in fp32
# (csrc/conv_split.hip) — fp32-equivalent results at 3/16 of the fp32-MFMA cost.
MATH_MODES = ("fp32", "f16x3")
DEFAULT_MATH = "fp32"  # process-wide default; per-plan: Plan(math=...) / the module's ``conv_math`` attribute
SPLIT_MIN_BLOCKS = 256  # fewer 8x16x64 tiles than CUs: the fp32 kernels' finer tiles win


def split_eligible(srcs, cout: int, N: int, Ho: int, Wo: int, pad_mode: int) -> bool:
    (v0, c0) = srcs[0]
    if c0.kernel_size[0] != 3 or c0.stride[0] != 1 or pad_mode != PAD_ZEROS or cout % 64:
        return False
    if len(srcs) > 1 and (srcs[1][1].kernel_size[0] != 1 or srcs[1][1].stride[0] != 1):
        return False  # the fused second source is a 1x1 projection (BasicBlock downsample at stride 1)
    if Wo < 16 or Ho < 8:
        return False
    return N * (-(-Ho // 8)) * (-(-Wo // 16)) * (cout // 64) >= SPLIT_MIN_BLOCKS


def choose_split_rows(N: int, Ho: int, Wo: int, cout: int) -> int:
    """16-row tiles (best weight-panel amortisation) when they fill 256 CUs x 3 resident workgroups
    without wasting rows, else 8-row tiles (twice the workgroups, no waste on 24-row maps)."""
    def eff(rows, bonus):
        ty = -(-Ho // rows)
        blocks = N * ty * (-(-Wo // 16)) * (cout // 64)
        return bonus * (Ho / (ty * rows)) * min(1.0, blocks / 768.0)
    return 16 if Ho >= 16 and eff(16, 1.0) >= eff(8, 0.93) else 8


# Decoder concats: optionally fold the x2 upsampling (and the concat itself) into the consumer conv's halo loader
# instead of running upsample2_k into a concat buffer.  Bit-identical, 16 launches and ~1 GB of HBM traffic less per
# 32-frame step — and SLOWER on MI355X (tools/perf_up.py, profiles/r02/fused_upsample_experiment.txt: 82.2 ms
# materialised vs 91.0 ms fused with 8-row tiles / 85.8 ms with 4-row tiles at B=32; 4.10 vs 4.16 ms at B=1): the 4x
# register prefetch of the loader either spills (8-row tile) or costs the tile's operand reuse (4-row), which outweighs
# the 2.6 ms upsample2_k takes at HBM speed.  Off by default; FUSED_UP_ROWS = tile rows of the convs that read such a
# concat.
FUSE_UPSAMPLE = False
FUSED_UP_ROWS = 4

# One launch per dependency level where its members are small (one frame, low-resolution maps): idh_run_ops merges the
# consecutive ops of a level that carry its group id into one ``level_k`` grid when each has <= 512 workgroups.
MERGE_LEVELS = True
# Matching-encoder head: InstanceNorm2d(128) + LeakyReLU applied by the following 3x3 conv while it stages its halo
# (idh_conv_src.norm) instead of a normalised copy of the tensor; False = materialise (bit-identical, tests compare).
FUSE_HEAD_NORM = True
# ... and its first 1x1 conv reading the backbone's NCHW map in place (IDH_OP_POINTWISE_NCHW) instead of after a
# layout-import pass; False = import_nchw + the generic conv (bit-identical).
FUSE_HEAD_IMPORT = True

TARGET_WAVES = 2048  # ~2 waves per SIMD over 256 CUs x 4 SIMDs
MIN_WAVES = 1024


def lds_eligible(srcs, cout: int, Wo: int, pad_mode: int) -> bool:
    """Shape family of the LDS-staged kernel (csrc/conv.hip conv3x3_lds_k)."""
    (v0, c0) = srcs[0]
    if c0.kernel_size[0] != 3 or c0.stride[0] != 1 or cout % 16 or Wo < 16:
        return False
    if pad_mode != PAD_ZEROS and (pad_mode != PAD_REPLICATE or len(srcs) > 1):
        return False
    if len(srcs) > 1:
        (v1, c1) = srcs[1]
        strided3 = c1.kernel_size[0] == 3 and c1.stride[0] == 2 and pad_mode == PAD_ZEROS and cout % 32 == 0 and not isinstance(v1, CatView)
        if not strided3 and (c1.kernel_size[0] != 1 or c1.stride[0] != 1):  # BasicBlock's downsample(x): 1x1, or 3x3 stride 2
            return False
    return True


# A lone 3x3 stride-2 conv (conv1 of a stride-2 BasicBlock, layers.py:62-66) on the LDS-staged kernel's stride-2 loader (the one the strided
# projections use: halo de-interleaved by column parity, every tap's fragment read conflict-free) instead of the direct-fragment kernel, which
# re-reads each input pixel 9 times through L1 (46-55 % MFMA-busy).  Below S2_FIRST_MIN_BLOCKS 4-row workgroups the direct kernel stays: at
# one or two frames it is a member of the level launch (level_k).
S2_FIRST = True
S2_FIRST_MIN_BLOCKS = 512


def s2_first_eligible(srcs, cout: int, N: int, Ho: int, Wo: int, pad_mode: int) -> bool:
    if not S2_FIRST or len(srcs) != 1:
        return False
    (v0, c0) = srcs[0]
    if c0.kernel_size[0] != 3 or c0.stride[0] != 2 or pad_mode != PAD_ZEROS or cout % 32 or Wo < 16 or isinstance(v0, CatView):
        return False
    return N * (-(-Wo // 16)) * (-(-Ho // 4)) * (cout // (16 * lds_subtiles(cout))) >= S2_FIRST_MIN_BLOCKS


def lds_subtiles(cout: int) -> int:
    """16-channel output sub-tiles per workgroup of the LDS-staged kernel (the op's tile_n): 64 channels when the
    layer has them, else 32 / 16 (the matching encoder's 128 -> 16 conv)."""
    return 4 if cout % 64 == 0 else (2 if cout % 32 == 0 else 1)


# Small grids (one frame, low-resolution levels): halve the workgroup's channel tile (64 -> 32) when even 4-row tiles
# leave fewer than this many workgroups, to shorten each workgroup's MFMA phase and spread it over more CUs.
# Measured (tools/perf_levels.py <B> <threshold>, conv stage of the hot path): B=1 3.31 -> 3.04 ms, B=2 5.06 -> 4.88 ms,
# B=4 and B=8 unchanged with 400; 800 / 1600 cost 1-1.5 % at B=4.
NARROW_TILE_BELOW = 400
NARROWEST_TILE_BELOW = 0  # below this many 64-channel workgroups use 16-channel tiles (0 = never)


SPLIT_MIN_CHUNKS = 6
SPLIT_MAX = 16
# decoder in_conv blocks: projection of the upsampled concat slices at low resolution (Plan.basic_block_upcat, pointwise_up_k).  Built, parity-tested,
# and OFF: at B = 32 the conv plan takes 28.80 against 28.53 ms with it.  Per 192x256 block the F(4x4) projection phase it removes costs 338 us
# (783 -> 445 us for conv2: that phase already runs at ~80 % of the matrix rate), pointwise_up_k + the low-resolution 1x1 conv that replace it
# 270 + 68 us, both HBM-bound (profiles/r05/experiments.md 3b).
PROJ_LOWRES = False
PROJ_LOWRES_MIN_TILES = 1536
PROJ_CHUNK_WEIGHT = 0.5  # a 1x1 (projection) chunk in units of a 3x3 chunk when the split-K factor is chosen (1.0 / 0.5 / 0.25 measured: B=1 2.353 / 2.330 / 2.343 ms, B=4 5.135 / 5.099 / 5.110)


def choose_lds_tile(N: int, Ho: int, Wo: int, cout: int, chunks: int):
    """(tile code, split): 8-row tiles (code 8) when they already give one full round of
    256 CUs x 3 resident workgroups, else 4-row tiles (code 9); split K only when the grid still
    cannot fill the chip AND every split keeps >= 6 chunks of 16 channels (measured on MI355X,
    tools/perf_conv_layers.py)."""
    per_row_tiles = N * (-(-Wo // 16)) * (cout // (16 * lds_subtiles(cout)))
    code, rows = 8, 8
    if per_row_tiles * (-(-Ho // 8)) < 768:
        code, rows = 9, 4
    blocks = per_row_tiles * (-(-Ho // rows))
    return code, max(1, min(-(-768 // blocks), chunks // SPLIT_MIN_CHUNKS, SPLIT_MAX))


def choose_tiles(M: int, cout: int, steps: int):
    nsub = ceil16(cout) // 16
    tn = 4 if nsub % 4 == 0 else (2 if nsub % 2 == 0 else 1)
    tm, waves = 1, 0
    for cand in (4, 2, 1):
        waves = -(-M // (16 * cand)) * (nsub // tn)
        tm = cand
        if waves >= TARGET_WAVES:
            break
    split = 1
    if waves < MIN_WAVES:
        split = max(1, min(-(-MIN_WAVES // waves), steps // 4, 32))
    return tm, tn, split


class Plan:
    """Ordered op list + owned buffers.  ``run()`` = one C-ABI call."""

    def __init__(self, device, math: Optional[str] = None):
        self.device = device
        self.math = DEFAULT_MATH if math is None else math
        if self.math not in MATH_MODES:
            raise _lib.IdhError(f"unknown conv math mode {self.math!r} (expected one of {MATH_MODES})")
        self.ops: List[Op] = []
        self.meta: List[dict] = []  # per op: regions read / written, for the level scheduler
        self.keep: List[torch.Tensor] = []  # buffers / packed weights referenced by raw pointer
        self._free: Dict[tuple, List[torch.Tensor]] = {}  # released buffers by shape (``release``)
        self.recycled_candidates = 0  # buffers handed to the pool by ``release``
        self.recycled = 0  # ... and how many of them a later ``buffer()`` took over (tests assert that aliasing really happened)
        self._arr = None
        self._pos: Optional[List[int]] = None  # op index at build time -> index after schedule()
        self.flops = 0  # 2*MAC of the conv ops (algorithmic, no padding)

    # buffers -------------------------------------------------------------------------
    def buffer(self, N, H, W, Cch) -> View:
        """Dense NHWC buffer.  The conv kernel reads whole 16-channel blocks, so channel counts
        that are not a multiple of 16 get zero-filled padding channels (written by nobody)."""
        cs = ceil16(Cch)
        pool = self._free.get((N, H, W, cs)) if cs == Cch else None
        if pool:
            self.recycled += 1
            return View(pool.pop(), 0, Cch)
        alloc = torch.zeros if cs != Cch else torch.empty
        t = alloc(N, H, W, cs, device=self.device, dtype=torch.float32)
        self.keep.append(t)
        return View(t, 0, Cch)

    def release(self, v: View):
        """Liveness reuse of activation memory: the caller will record no further op that touches ``v`` (a whole buffer of this plan), so a
        later ``buffer()`` of the same shape may alias it.  Safe under the level scheduler by construction — the op regions are keyed by
        the tensor, so the new writer gets a write-after-read dependency on the last reader — but that dependency also serialises blocks
        that used to share a dependency level, which small grids need for their grouped launches: only buffers of at least
        ``REUSE_MIN_BYTES`` (ops that fill the chip on their own) are pooled."""
        t = v.buf
        if (BUFFER_REUSE and v.c0 == 0 and v.C == t.shape[-1] and t.dim() == 4 and t.numel() * 4 >= REUSE_MIN_BYTES
                and any(t is k for k in self.keep)):
            pool = self._free.setdefault(tuple(t.shape), [])
            if any(t is f for f in pool):  # a second release of a buffer that is still in the pool would hand it to TWO later allocations
                raise _lib.IdhError("Plan.release: buffer released twice")
            pool.append(t)
            self.recycled_candidates += 1

    # ops -----------------------------------------------------------------------------
    def conv(self, x: View, conv: nn.Conv2d, out: View, act=ACT_NONE, slope=0.2, res: Optional[View] = None,
             x2: Optional[View] = None, conv2: Optional[nn.Conv2d] = None, pad_mode=PAD_ZEROS, norm=None):
        """``norm`` = (stats, act, slope): read ``x`` through act((x - mean) * rstd) with the (N, 2, C) statistics of
        ``instance_norm(x, None)`` (``norm_on_load_eligible`` layers only)."""
        op = Op()
        op.kind = OP_CONV
        op.N = x.N
        srcs = [(x, conv)] + ([(x2, conv2)] if x2 is not None else [])
        steps = 0
        use_split = self.math != "fp32" and split_eligible(srcs, conv.out_channels, out.N, out.H, out.W, pad_mode)
        use_wino = (WINOGRAD and self.math == "fp32" and norm is None and
                    wino_eligible(srcs, conv.out_channels, out.N, out.H, out.W, pad_mode))
        use_wino4 = (WINOGRAD4 and self.math == "fp32" and norm is None and (x2 is None or res is None) and
                     wino4_eligible(srcs, conv.out_channels, out.N, out.H, out.W, pad_mode, act, out, res, slope))
        if use_wino4:
            use_wino = False
        for i, (v, cv) in enumerate(srcs):
            ks, st = cv.kernel_size[0], cv.stride[0]
            if v.C != cv.in_channels:
                raise _lib.IdhError(f"conv expects {cv.in_channels} input channels, view has {v.C}")
            cat = v if isinstance(v, CatView) else None
            if cat is not None:
                if use_split or not lds_eligible(srcs, conv.out_channels, out.W, pad_mode) or st != 1:
                    raise _lib.IdhError("a fused-upsample concat can only feed the LDS-staged fp32 conv kernel")
                if cat.direct.C % 16 or any(u.C != cat.ups[0].C or u.C % 16 or (2 * u.H, 2 * u.W) != (cat.H, cat.W) for u in cat.ups) or not 1 <= len(cat.ups) <= 2:
                    raise _lib.IdhError("fused-upsample concat: channel counts must be multiples of 16 and the maps exactly half size")
                v = cat.direct
            elif v.C % 16 and (v.c0 != 0 or v.cs != ceil16(v.C)):
                raise _lib.IdhError("a conv input whose channel count is not a multiple of 16 must be a whole zero-padded buffer")
            if use_split:  # one blob: [3x3 panels][1x1 panels of the second source][scales]
                w = split_packed_weight(conv, self.math, conv2)
            elif use_wino4 and i == 0:
                w = packed_wino4_weight(cv)
            elif use_wino and i == 0:
                w = packed_wino_weight(cv)
            else:
                w = packed_weight(cv)
            self.keep.append(w)
            s = op.src[i]
            cin = cat.C if cat is not None else v.C
            s.in_, s.w, s.cs, s.H, s.W, s.Cin = v.ptr, w.data_ptr(), v.cs, v.H, v.W, cin
            s.ks, s.stride, s.pad_mode = ks, st, pad_mode
            if cat is not None:
                s.up_c0, s.up_C = cat.direct.C, cat.ups[0].C
                for ui, u in enumerate(cat.ups):
                    s.up_in[ui], s.up_cs[ui] = u.ptr, u.cs
            steps += ks * ks * (ceil16(cin) // 16)
            self.flops += 2 * out.N * out.H * out.W * cv.out_channels * cv.in_channels * ks * ks
        bias = conv.bias
        if conv2 is not None and conv2.bias is not None:
            bias = (conv.bias + conv2.bias).detach() if conv.bias is not None else conv2.bias
        if bias is not None:
            bias = bias.detach().contiguous()
            self.keep.append(bias)
            op.bias = bias.data_ptr()
        if res is not None:
            op.res, op.res_cs = res.ptr, res.cs
        op.out, op.out_cs = out.ptr, out.cs
        op.Ho, op.Wo, op.Cout = out.H, out.W, conv.out_channels
        op.act, op.slope = act, slope
        M = out.N * out.H * out.W
        if use_split:
            tm, tn, split = SPLIT_CODE[self.math], choose_split_rows(out.N, out.H, out.W, conv.out_channels), 1
        elif use_wino4:
            tm, tn, split = TILE_WINO4, 0, 1
        elif use_wino:
            tm, tn, split = TILE_WINO, 0, 1
        elif lds_eligible(srcs, conv.out_channels, out.W, pad_mode):
            # split-K factor from the chunk count in units of a 3x3 chunk: a 1x1 (projection) chunk is PROJ_CHUNK_WEIGHT of one
            chunks = int(sum((ceil16(v.C) // 16) * (1.0 if cv.kernel_size[0] == 3 else PROJ_CHUNK_WEIGHT) for v, cv in srcs))
            tm, split = choose_lds_tile(out.N, out.H, out.W, conv.out_channels, chunks)
            if tm == 8 and FUSED_UP_ROWS == 4 and any(isinstance(v, CatView) for v, _ in srcs):
                tm = 9
            tn = lds_subtiles(conv.out_channels)
            if tn == 4 and tm == 9 and NARROW_TILE_BELOW and not any(isinstance(v, CatView) for v, _ in srcs):
                blocks64 = out.N * (-(-out.H // 4)) * (-(-out.W // 16)) * (conv.out_channels // 64) * split
                if blocks64 < NARROW_TILE_BELOW:
                    tn = 1 if blocks64 < NARROWEST_TILE_BELOW else 2
            tn = 0 if tn == 4 else tn
        elif self.math == "fp32" and norm is None and s2_first_eligible(srcs, conv.out_channels, out.N, out.H, out.W, pad_mode):
            tm, split = choose_lds_tile(out.N, out.H, out.W, conv.out_channels, ceil16(x.C) // 16)
            tn = lds_subtiles(conv.out_channels)
            tn = 0 if tn == 4 else tn
        else:
            tm, tn, split = choose_tiles(M, conv.out_channels, steps)
        op.tile_m, op.tile_n, op.split_k = tm, tn, split
        if norm is not None:
            stats, n_act, n_slope = norm
            if tm not in (8, 9) or tn != 1 or x2 is not None or isinstance(x, CatView) or tuple(stats.shape) != (x.N, 2, x.C):
                raise _lib.IdhError("normalise-on-load needs the LDS-staged fp32 conv with 16-channel tiles and one source")
            op.src[0].norm, op.src[0].norm_act, op.src[0].norm_slope = stats.data_ptr(), n_act, n_slope
        if split > 1:
            ws = torch.empty(split * M * ceil16(conv.out_channels), device=self.device, dtype=torch.float32)
            self.keep.append(ws)
            op.ws = ws.data_ptr()
        self.ops.append(op)
        reads = ([_region(res)] if res is not None else []) + ([(norm[0].data_ptr(), 0, 1)] if norm is not None else [])
        for v, _ in srcs:
            if isinstance(v, CatView):
                reads += [_region(v.direct)] + [_region(u) for u in v.ups]
            else:
                reads.append(_region(v, pad16=True))
        self.meta.append({"reads": reads, "writes": [_region(out)]})
        self._arr = None
        return out

    def upsample2(self, x: View, out: View, nearest: bool = False):
        op = Op()
        op.kind, op.N = (OP_UPSAMPLE2_NEAREST if nearest else OP_UPSAMPLE2), x.N
        s = op.src[0]
        s.in_, s.cs, s.H, s.W, s.Cin = x.ptr, x.cs, x.H, x.W, x.C
        op.out, op.out_cs = out.ptr, out.cs
        self.ops.append(op)
        self.meta.append({"reads": [_region(x)], "writes": [_region(out)]})
        self._arr = None
        return out

    def pointwise_up(self, x: View, conv: nn.Conv2d, low: View, out: View):
        """out = conv1x1(x) + upsample2(low) (IDH_OP_POINTWISE_UP, csrc/conv.hip pointwise_up_k)."""
        if conv.kernel_size[0] != 1 or conv.in_channels != x.C or conv.out_channels != out.C or low.C != out.C or (2 * low.H, 2 * low.W) != (x.H, x.W):
            raise _lib.IdhError("pointwise_up: a 1x1 conv of x plus the x2 upsampling of a half-resolution map with the output's channels")
        op = Op()
        op.kind, op.N = OP_POINTWISE_UP, x.N
        w = packed_weight(conv)
        self.keep.append(w)
        s = op.src[0]
        s.in_, s.w, s.cs, s.H, s.W, s.Cin, s.ks, s.stride = x.ptr, w.data_ptr(), x.cs, x.H, x.W, x.C, 1, 1
        l = op.src[1]
        l.in_, l.cs, l.H, l.W, l.Cin = low.ptr, low.cs, low.H, low.W, low.C
        if conv.bias is not None:
            b = conv.bias.detach().contiguous()
            self.keep.append(b)
            op.bias = b.data_ptr()
        op.out, op.out_cs, op.Ho, op.Wo, op.Cout = out.ptr, out.cs, out.H, out.W, out.C
        self.flops += 2 * x.N * x.H * x.W * conv.out_channels * conv.in_channels
        self.ops.append(op)
        self.meta.append({"reads": [_region(x), _region(low)], "writes": [_region(out)]})
        self._arr = None
        return out

    def instance_norm(self, x: View, out: Optional[View], act=ACT_NONE, slope=0.2):
        """nn.InstanceNorm2d(C) (no affine, eps 1e-5) optionally followed by LeakyReLU.  ``out=None``: statistics only —
        returns the (N, 2, C) mean / rstd tensor for a consumer conv that normalises on load (``conv(..., norm=...)``)."""
        if x.C % 4 or 256 % (x.C // 4):
            raise _lib.IdhError(f"instance-norm kernel needs C/4 to divide 256 (C={x.C})")
        op = Op()
        op.kind, op.N = OP_INSTNORM, x.N
        s = op.src[0]
        s.in_, s.cs, s.H, s.W, s.Cin = x.ptr, x.cs, x.H, x.W, x.C
        op.act, op.slope = act, slope
        if out is not None:
            op.out, op.out_cs = out.ptr, out.cs
        nchunks = -(-(x.H * x.W) // 1024)
        ws = torch.empty(x.N * (nchunks + 1) * 2 * x.C, device=self.device, dtype=torch.float32)
        self.keep.append(ws)
        op.ws = ws.data_ptr()
        self.ops.append(op)
        stats = ws[x.N * nchunks * 2 * x.C:].view(x.N, 2, x.C)
        self.meta.append({"reads": [_region(x)], "writes": [_region(out)] if out is not None else [(stats.data_ptr(), 0, 1)]})
        self._arr = None
        return out if out is not None else stats

    def copy(self, x: View, out: View):
        """NHWC view -> NHWC view (e.g. an existing feature map into a slice of a concat buffer)."""
        if (x.N, x.H, x.W, x.C) != (out.N, out.H, out.W, out.C):
            raise _lib.IdhError("copy: shape mismatch")
        op = Op()
        op.kind, op.N = OP_COPY, x.N
        s = op.src[0]
        s.in_, s.cs, s.H, s.W, s.Cin = x.ptr, x.cs, x.H, x.W, x.C
        op.out, op.out_cs = out.ptr, out.cs
        self.ops.append(op)
        self.meta.append({"reads": [_region(x)], "writes": [_region(out)]})
        self._arr = None
        return out

    def import_nchw(self, shape_nchw, out: View) -> int:
        """(N,C,H,W) dense -> NHWC slice.  The source pointer is patched per call with
        ``set_in``; returns the op index."""
        N, Cc, H, W = [int(v) for v in shape_nchw]
        if (N, H, W, Cc) != (out.N, out.H, out.W, out.C):
            raise _lib.IdhError(f"import of {tuple(shape_nchw)} into a view of {(out.N, out.C, out.H, out.W)}")
        op = Op()
        op.kind, op.N = OP_IMPORT, N
        s = op.src[0]
        s.H, s.W, s.Cin = H, W, Cc
        op.out, op.out_cs = out.ptr, out.cs
        self.ops.append(op)
        self.meta.append({"reads": [], "writes": [_region(out)]})
        self._arr = None
        return len(self.ops) - 1

    @staticmethod
    def pointwise_nchw_eligible(conv: nn.Conv2d) -> bool:
        """Shape family of csrc/conv.hip pointwise_nchw_k: the matching head's nn.Conv2d(64, 128, 1)."""
        return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.in_channels == 64 and conv.out_channels == 128
                and conv.groups == 1)

    def pointwise_nchw(self, shape_nchw, conv: nn.Conv2d, out: View) -> int:
        """1x1 conv read straight from a dense (N,C,H,W) tensor into an NHWC view (IDH_OP_POINTWISE_NCHW): layout
        import and convolution in one pass.  The source pointer is patched per call with ``set_in``; returns the op index."""
        N, Cc, H, W = [int(v) for v in shape_nchw]
        if (N, H, W) != (out.N, out.H, out.W) or Cc != conv.in_channels or out.C != conv.out_channels or not self.pointwise_nchw_eligible(conv):
            raise _lib.IdhError(f"pointwise_nchw: {tuple(shape_nchw)} with {conv} into a view of {(out.N, out.C, out.H, out.W)}")
        op = Op()
        op.kind, op.N = OP_POINTWISE_NCHW, N
        s = op.src[0]
        w = packed_weight(conv)
        self.keep.append(w)
        s.w, s.H, s.W, s.Cin, s.ks, s.stride = w.data_ptr(), H, W, Cc, 1, 1
        if conv.bias is not None:
            bias = conv.bias.detach().contiguous()
            self.keep.append(bias)
            op.bias = bias.data_ptr()
        op.out, op.out_cs, op.Ho, op.Wo, op.Cout = out.ptr, out.cs, H, W, conv.out_channels
        self.flops += 2 * N * H * W * conv.out_channels * conv.in_channels
        self.ops.append(op)
        self.meta.append({"reads": [], "writes": [_region(out)]})
        self._arr = None
        return len(self.ops) - 1

    def export_nchw(self, x: View, out_nchw: Optional[torch.Tensor] = None) -> int:
        op = Op()
        op.kind, op.N = OP_EXPORT, x.N
        s = op.src[0]
        s.in_, s.cs, s.H, s.W, s.Cin = x.ptr, x.cs, x.H, x.W, x.C
        if out_nchw is not None:
            op.out = out_nchw.data_ptr()
        self.ops.append(op)
        self.meta.append({"reads": [_region(x)], "writes": []})
        self._arr = None
        return len(self.ops) - 1

    def head(self, x: View, conv: nn.Conv2d, out: torch.Tensor):
        """1x1 conv to one channel (DepthDecoderPP heads, reference networks.py:158-161)."""
        if conv.out_channels != 1 or conv.kernel_size != (1, 1):
            raise _lib.IdhError("pointwise head kernel covers Conv2d(C,1,1) only")
        op = Op()
        op.kind, op.N = OP_HEAD, x.N
        w = conv.weight.detach().reshape(-1).contiguous()
        b = conv.bias.detach().contiguous()
        self.keep += [w, b]
        s = op.src[0]
        s.in_, s.w, s.cs, s.H, s.W, s.Cin = x.ptr, w.data_ptr(), x.cs, x.H, x.W, x.C
        op.bias, op.out = b.data_ptr(), out.data_ptr()
        self.ops.append(op)
        self.meta.append({"reads": [_region(x)], "writes": []})
        self._arr = None
        return len(self.ops) - 1

    # composite: BasicBlock (reference layers.py:78-95) ---------------------------------
    def basic_block(self, x: View, blk, out: Optional[View] = None) -> View:
        st = blk.conv1.stride[0]
        Ho = (x.H + 2 - 3) // st + 1
        Wo = (x.W + 2 - 3) // st + 1
        planes = blk.conv1.out_channels
        h = self.buffer(x.N, Ho, Wo, planes)
        self.conv(x, blk.conv1, h, act=ACT_LRELU, slope=0.2)
        if out is None:
            out = self.buffer(x.N, Ho, Wo, planes)
        if blk.downsample is None:
            self.conv(h, blk.conv2, out, act=ACT_LRELU, slope=0.2, res=x)
        elif (self.math != "fp32" and blk.downsample[0].kernel_size[0] == 3
              and split_eligible([(h, blk.conv2)], planes, x.N, Ho, Wo, PAD_ZEROS)):
            # stride-2 block: the strided 3x3 projection (1/3 of the block's conv2-stage flops) runs on its own with
            # the fp32 direct kernel and enters the split-precision conv2 as the residual
            proj = self.buffer(x.N, Ho, Wo, planes)
            self.conv(x, blk.downsample[0], proj)
            self.conv(h, blk.conv2, out, act=ACT_LRELU, slope=0.2, res=proj)
        else:
            self.conv(h, blk.conv2, out, act=ACT_LRELU, slope=0.2, x2=x, conv2=blk.downsample[0])
        self.release(h)  # the block's intermediate dies with conv2
        return out

    def basic_block_upcat(self, cat: View, lows: List[View], blk) -> Optional[View]:
        """BasicBlock on cat = [right | up(lows[0]) | up(lows[1])...] (the UNet++ decoders' in_conv, networks.py:52-77) with the projection
        branch split by linearity: a 1x1 conv commutes with bilinear upsampling (its weights sum to one, so the bias may sit on either side),
        ``downsample(cat) = W_a right + up(W_b lo + W_c lo2 + b)``.  Two thirds of the projection run at a quarter of the pixels
        (one 1x1 conv on the low-resolution maps), ``pointwise_up_k`` adds the right slice's share, and conv2 takes the sum as an ordinary
        residual - the F(4x4) kernel's plain + residual instance instead of its projection phase over 3 x planes channels.  Same function as
        ``basic_block(cat, blk)`` up to fp32 rounding.  Returns None (caller falls back) where it does not apply: small grids (the extra
        launches cost more than the projection phase there), widths the pointwise kernel does not have."""
        ds = blk.downsample[0] if blk.downsample is not None else None
        planes = blk.conv1.out_channels
        c = planes
        if (not PROJ_LOWRES or self.math != "fp32" or ds is None or ds.kernel_size[0] != 1 or ds.stride[0] != 1 or planes not in (64, 128) or
                not 1 <= len(lows) <= 2 or cat.C != c * (1 + len(lows)) or any(l.C != c or (2 * l.H, 2 * l.W) != (cat.H, cat.W) for l in lows)):
            return None
        if cat.N * (-(-cat.H // 8)) * (-(-cat.W // 32)) * (planes // 64) < PROJ_LOWRES_MIN_TILES:
            return None
        key = (ds.weight.data_ptr(), _lib.param_version(ds.weight), None if ds.bias is None else _lib.param_version(ds.bias), str(ds.weight.device))
        parts = getattr(ds, "_idh_proj_parts", None)
        if parts is None or parts[0] != key:
            with torch.no_grad():
                mods = []
                for i in range(1 + len(lows)):
                    m = nn.Conv2d(c, planes, 1, bias=(i == 1 and ds.bias is not None)).to(ds.weight.device)
                    m.weight.copy_(ds.weight[:, i * c:(i + 1) * c])
                    if m.bias is not None:
                        m.bias.copy_(ds.bias)
                    m.requires_grad_(False)
                    mods.append(m)
            parts = (key, mods)
            ds._idh_proj_parts = parts
        wa, wb = parts[1][0], parts[1][1]
        wc = parts[1][2] if len(lows) == 2 else None
        plo = self.buffer(cat.N, lows[0].H, lows[0].W, planes)
        if wc is None:
            self.conv(lows[0], wb, plo)
        else:
            self.conv(lows[0], wb, plo, x2=lows[1], conv2=wc)
        r = self.buffer(cat.N, cat.H, cat.W, planes)
        self.pointwise_up(cat.slice(0, c), wa, plo, r)
        h = self.buffer(cat.N, cat.H, cat.W, planes)
        self.conv(cat, blk.conv1, h, act=ACT_LRELU, slope=0.2)
        out = self.buffer(cat.N, cat.H, cat.W, planes)
        self.conv(h, blk.conv2, out, act=ACT_LRELU, slope=0.2, res=r)
        self.release(h)
        self.release(r)
        self.release(plo)
        return out

    # scheduling ----------------------------------------------------------------------
    def schedule(self):
        """Re-order the ops by dependency level (longest path from the inputs) and mark the
        4-row-tile LDS convs of one level as a launch group: they are mutually independent, so
        ``idh_run_ops`` runs them as ONE grid.  This is how the many small low-resolution convs
        of the UNet++ grid (each filling only a fraction of 256 CUs) get to run side by side
        — the reference executes them strictly one after another."""
        self.schedule_segments(0)

    @staticmethod
    def _launch_rank(op) -> tuple:
        """Order of the ops of one dependency level, and which of them carry the level's group id (rank < 3):
        4-row LDS convs first, by channel tile (one ``conv3x3_lds_group_k`` grid per run of equal tiles); then, with
        ``MERGE_LEVELS``, the other members a mixed ``level_k`` launch can host (csrc/conv.hip: the stride-2 /
        1x1 direct conv with 16x64 wave tiles, bilinear x2 upsampling); everything else runs on its own."""
        if WINO_GROUP and op.kind == OP_CONV and op.tile_m == TILE_WINO:
            # Winograd convs of a level share one persistent grid (conv3x3_wino_group_k): plain ones, then those with a fused
            # 1x1 source; the largest first so that the small ones fill its tail
            return (-1, 1 if op.src[1].in_ else 0, -op.N * op.Ho * op.Wo * op.Cout)
        if op.kind == OP_CONV and op.tile_m == 9:
            return (0, op.tile_n)
        if MERGE_LEVELS and op.kind == OP_CONV and op.tile_m == 1 and op.tile_n == 4:
            return (1, 0)
        if MERGE_LEVELS and op.kind == OP_UPSAMPLE2:
            return (2, 0)
        if MERGE_LEVELS and op.kind == OP_IMPORT:
            return (2, 1)  # the layout imports of a level: adjacent and under the level's group id -> one import_nchw_group_k grid when they are small
        return (3, 0)

    def schedule_segments(self, n_first: int) -> int:
        """``schedule()`` for a plan that is replayed in two pieces — ops [0, n_first) then the rest — because a
        kernel outside the plan (the fused cost volume) consumes the first piece's output and produces the second
        piece's input.  Each piece is levelled on its own; returns the length of the first piece."""
        n = len(self.ops)
        n_first = max(0, min(int(n_first), n))
        order: List[int] = []
        level = [0] * n
        for lo, hi in ((0, n_first), (n_first, n)):
            for j in range(lo, hi):
                mj = self.meta[j]
                for i in range(lo, j):
                    mi = self.meta[i]
                    if _overlap(mi["writes"], mj["reads"]) or _overlap(mi["writes"], mj["writes"]) or _overlap(mi["reads"], mj["writes"]):
                        level[j] = max(level[j], level[i] + 1)
            order += sorted(range(lo, hi), key=lambda k: (level[k],) + self._launch_rank(self.ops[k]) + (k,))
        for k in range(n):
            self.ops[k].group = level[k] + 1 if self._launch_rank(self.ops[k])[0] < 3 else 0
        self.ops = [self.ops[k] for k in order]
        self.meta = [self.meta[k] for k in order]
        self.levels = [level[k] for k in order]
        pos = [0] * n
        for new, old in enumerate(order):
            pos[old] = new
        self._pos = pos if self._pos is None else [pos[p] for p in self._pos]
        self._arr = None
        return n_first

    # execution -----------------------------------------------------------------------
    def _array(self):
        if self._arr is None:
            self._arr = (Op * len(self.ops))(*self.ops)
        return self._arr

    def _idx(self, idx: int) -> int:
        return idx if self._pos is None else self._pos[idx]

    def run(self, start: int = 0, end: Optional[int] = None):
        """Replay ops [start, end) (default: all) with one C-ABI call."""
        L = _bind()
        arr = self._array()
        end = len(self.ops) if end is None else end
        if end <= start:
            return
        base = C.addressof(arr) + start * C.sizeof(Op)
        _lib.check(L.idh_run_ops(C.c_void_p(base), end - start, _lib.stream_ptr()), "idh_run_ops")

    def count_launches(self, start: int = 0, end: Optional[int] = None) -> int:
        """Kernel launches ``run(start, end)`` issues (``idh_count_launches``: same decisions, nothing launched)."""
        L = _bind()
        arr = self._array()
        end = len(self.ops) if end is None else end
        if end <= start:
            return 0
        n = L.idh_count_launches(C.c_void_p(C.addressof(arr) + start * C.sizeof(Op)), end - start)
        if n < 0:
            _lib.check(n, "idh_count_launches")
        return n

    def set_in(self, idx: int, t: torch.Tensor):
        """idx = op index returned at build time (stable across schedule())."""
        self._array()[self._idx(idx)].src[0].in_ = t.data_ptr()

    def set_out(self, idx: int, t: torch.Tensor, exp_out: Optional[torch.Tensor] = None):
        """Patch an op's output pointer; ``exp_out``: the head op's optional exp(output) tensor."""
        op = self._array()[self._idx(idx)]
        op.out = t.data_ptr()
        if exp_out is not None:
            op.ws = exp_out.data_ptr()


def math_of(module: nn.Module) -> str:
    """Conv arithmetic of a drop-in module: its ``conv_math`` attribute (set by ``dropin.convert(model,
    math=...)``) or the process-wide DEFAULT_MATH."""
    return getattr(module, "conv_math", None) or DEFAULT_MATH


PLAN_CACHE_ENTRIES = 4


class ParamKey(tuple):
    """The part of a plan-cache key that names parameter versions / build-time switches (as opposed to input shapes)."""

    def __add__(self, other):  # (tuple.__add__ would hand back a plain tuple, which PlanCache.put's stale-twin eviction does not recognise)
        return ParamKey(tuple.__add__(self, other))


class PlanCache:
    """Small LRU of execution plans (one entry = ops + all activation buffers of one input shape): an evaluation loop whose last
    batch is ragged, or a caller alternating two layouts / batch sizes, replays instead of rebuilding; the oldest entry (and its
    buffers) goes when a fifth shape arrives.  The LRU is for SHAPES only: an entry that differs from a new key just in its ParamKey
    parts (weights reloaded or updated in place, a build switch toggled) can never be hit again, so it is dropped at once — a
    checkpoint load after a warm-up forward does not double the resident activation memory."""

    def __init__(self, entries: int = PLAN_CACHE_ENTRIES):
        import collections

        self.entries = entries
        self.d = collections.OrderedDict()

    def get(self, key):
        ent = self.d.get(key)
        if ent is not None:
            self.d.move_to_end(key)
        return ent

    @staticmethod
    def _same_shapes(a, b) -> bool:
        return (isinstance(a, tuple) and isinstance(b, tuple) and len(a) == len(b) and
                all(x == y or (isinstance(x, ParamKey) and isinstance(y, ParamKey)) for x, y in zip(a, b)))

    def put(self, key, ent):
        for k in [k for k in self.d if k != key and self._same_shapes(k, key)]:
            del self.d[k]  # same shapes, stale parameters / switches
        self.d[key] = ent
        self.d.move_to_end(key)
        while len(self.d) > self.entries:
            self.d.popitem(last=False)
        return ent

    def values(self):
        return self.d.values()

    def __len__(self):
        return len(self.d)

    def clear(self):
        self.d.clear()


def build_flags() -> tuple:
    """Module-level switches that shape a plan at build time (part of every plan-cache key: toggling one takes effect on the
    next call instead of silently replaying a plan built under the old setting)."""
    return (WINO_GROUP, FUSE_UPSAMPLE, FUSED_UP_ROWS, MERGE_LEVELS, FUSE_HEAD_NORM, FUSE_HEAD_IMPORT, NARROW_TILE_BELOW, NARROWEST_TILE_BELOW, SPLIT_MIN_CHUNKS,
            SPLIT_MAX, SPLIT_MIN_BLOCKS, WINOGRAD, WINO_MIN_TILES, WINO_MIN_FILL, WINOGRAD4, WINOGRAD4_PROJ, WINO4_MIN_TILES, WINO4_MIN_FILL, BUFFER_REUSE, REUSE_MIN_BYTES, DEFAULT_MATH,
            S2_FIRST, S2_FIRST_MIN_BLOCKS, PROJ_CHUNK_WEIGHT, PROJ_LOWRES, PROJ_LOWRES_MIN_TILES)


def _plan_cache(module: nn.Module) -> PlanCache:
    c = module.__dict__.get("_idh_plans")
    if c is None:
        c = PlanCache()
        module.__dict__["_idh_plans"] = c
        _lib.watch_state_dict_loads(module)
    return c


def _param_key(module: nn.Module):
    return ParamKey((math_of(module), build_flags()) + tuple((p.data_ptr(), _lib.param_version(p)) for p in module.parameters()))


def _check_in(*ts):
    for t in ts:
        _lib.require_cuda_f32(t)
        if t.dim() != 4:
            raise _lib.IdhError(f"expected NCHW tensors, got shape {tuple(t.shape)}")


# --- BasicBlock as a stand-alone drop-in ----------------------------------------------------
def block_forward_nchw(blk, x: torch.Tensor) -> torch.Tensor:
    _check_in(x)
    x = x.contiguous()
    key = ("bb", tuple(x.shape), str(x.device), _param_key(blk))
    cache = _plan_cache(blk)
    ent = cache.get(key)
    if ent is None:
        p = Plan(x.device, math=math_of(blk))
        N, Cc, H, W = x.shape
        xin = p.buffer(N, H, W, Cc)
        i_in = p.import_nchw(x.shape, xin)
        y = p.basic_block(xin, blk)
        i_out = p.export_nchw(y)
        p.schedule()
        ent = (p, i_in, i_out, y)
        cache.put(key, ent)
    p, i_in, i_out, y = ent
    out = torch.empty(y.N, y.C, y.H, y.W, device=x.device, dtype=torch.float32)
    p.set_in(i_in, x)
    p.set_out(i_out, out)
    p.run()
    return out


# --- CVEncoder (reference networks.py:186-215) --------------------------------------------
def build_cv_encoder(p: Plan, enc, cost: View, img_feat_shapes: Sequence[Sequence[int]]):
    """Adds the encoder to plan ``p``.  Returns (outputs: List[View], import op indices for
    img_feats)."""
    outs, imports = [], []
    x = cost
    for i in range(enc.num_blocks):
        ds = enc.convs[f"ds_conv_{i}"]
        st = ds.conv1.stride[0]
        Ho, Wo = (x.H + 2 - 3) // st + 1, (x.W + 2 - 3) // st + 1
        cout = ds.conv1.out_channels
        cimg = img_feat_shapes[i][1]
        if tuple(img_feat_shapes[i][2:]) != (Ho, Wo):
            raise _lib.IdhError(f"img_feats[{i}] spatial size {tuple(img_feat_shapes[i][2:])} != {(Ho, Wo)}")
        cat = p.buffer(x.N, Ho, Wo, cout + cimg)
        p.basic_block(x, ds, out=cat.slice(0, cout))
        imports.append(p.import_nchw(img_feat_shapes[i], cat.slice(cout, cimg)))
        seq = enc.convs[f"conv_{i}"]
        y = p.basic_block(cat, seq[0])
        y = p.basic_block(y, seq[1])
        outs.append(y)
        x = y
    return outs, imports


def cv_encoder_forward_nchw(enc, x: torch.Tensor, img_feats: List[torch.Tensor]) -> List[torch.Tensor]:
    _check_in(x, *img_feats)
    x = x.contiguous()
    img_feats = [f.contiguous() for f in img_feats]
    key = ("cve", tuple(x.shape), tuple(tuple(f.shape) for f in img_feats), str(x.device), _param_key(enc))
    cache = _plan_cache(enc)
    ent = cache.get(key)
    if ent is None:
        p = Plan(x.device, math=math_of(enc))
        N, D, H, W = x.shape
        xin = p.buffer(N, H, W, D)
        i_x = p.import_nchw(x.shape, xin)
        outs, i_img = build_cv_encoder(p, enc, xin, [f.shape for f in img_feats])
        i_out = [p.export_nchw(o) for o in outs]
        p.schedule()
        ent = (p, i_x, i_img, i_out, outs)
        cache.put(key, ent)
    p, i_x, i_img, i_out, outs = ent
    res = [torch.empty(o.N, o.C, o.H, o.W, device=x.device, dtype=torch.float32) for o in outs]
    p.set_in(i_x, x)
    for i, f in zip(i_img, img_feats):
        p.set_in(i, f)
    for i, r in zip(i_out, res):
        p.set_out(i, r)
    p.run()
    return res


# --- UNet++ decoders (reference networks.py:20-84, 118-183) ------------------------------
def build_decoder(p: Plan, dec, feats: List[View]):
    """Adds the UNet++ grid to ``p``; returns {scale i: View of the surviving output_i result
    (before the optional 1x1 depth head)}."""
    prev = list(feats)
    outputs: List[View] = []
    final: Dict[int, View] = {}
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            right = dec.convs[f"right_conv_{i}{j - 1}"]
            diag = dec.convs[f"diag_conv_{i + 1}{j - 1}"]
            cout = right.conv1.out_channels
            has_up = (i + j) != 4
            xi = prev[i]
            seq = dec.convs[f"in_conv_{i}{j}"]
            fuse = (FUSE_UPSAMPLE and p.math == "fp32" and cout % 16 == 0 and xi.W >= 16 and
                    lds_eligible([(xi, seq[0].conv1)], seq[0].conv1.out_channels, xi.W, PAD_ZEROS) and seq[0].downsample is not None and
                    seq[0].downsample[0].kernel_size[0] == 1)
            if fuse:  # torch.cat + F.interpolate live in the consumer's loader
                r = p.basic_block(xi, right)
                lo = p.basic_block(prev[i + 1], diag)
                if (lo.H * 2, lo.W * 2) != (xi.H, xi.W):
                    raise _lib.IdhError("decoder pyramid levels must differ by exactly x2")
                ups = [lo] + ([p.basic_block(outputs[-1], dec.convs[f"up_conv_{i + 1}{j}"])] if has_up else [])
                cat = CatView(r, ups)
            else:
                cat = p.buffer(xi.N, xi.H, xi.W, cout * (3 if has_up else 2))
                p.basic_block(xi, right, out=cat.slice(0, cout))
                lo = p.basic_block(prev[i + 1], diag)
                if (lo.H * 2, lo.W * 2) != (xi.H, xi.W):
                    raise _lib.IdhError("decoder pyramid levels must differ by exactly x2")
                p.upsample2(lo, cat.slice(cout, cout))
                lows = [lo]
                # (liveness reuse) the half-resolution maps have no reader after their upsampling - unless the low-resolution projection
                # (basic_block_upcat, PROJ_LOWRES) is going to read them again in in_conv's first block
                hold_lows = PROJ_LOWRES
                if not hold_lows:
                    p.release(lo)
                if has_up:
                    lo2 = p.basic_block(outputs[-1], dec.convs[f"up_conv_{i + 1}{j}"])
                    p.upsample2(lo2, cat.slice(2 * cout, cout))
                    lows.append(lo2)
                    if not hold_lows:
                        p.release(lo2)
            y0 = p.basic_block_upcat(cat, lows, seq[0]) if isinstance(cat, View) else None
            if y0 is None:
                y0 = p.basic_block(cat, seq[0])
            if isinstance(cat, View):
                if hold_lows:
                    for l in lows:
                        p.release(l)
                p.release(cat)
            y = p.basic_block(y0, seq.conv_0)
            p.release(y0)
            outputs.append(y)
            if j == 4 - i:  # the only (i,j) whose output_i result survives in the dict
                head = dec.convs[f"output_{i}"]
                final[i] = p.basic_block(y, head[0]) if not isinstance(head[0], nn.Identity) else y
        prev = outputs[::-1]
    return final


def decoder_forward_nchw(dec, input_features: List[torch.Tensor]) -> Dict[str, torch.Tensor]:
    _check_in(*input_features)
    feats = [f.contiguous() for f in input_features]
    key = ("dec", tuple(tuple(f.shape) for f in feats), str(feats[0].device), _param_key(dec))
    cache = _plan_cache(dec)
    ent = cache.get(key)
    if ent is None:
        p = Plan(feats[0].device, math=math_of(dec))
        views, i_in = [], []
        for f in feats:
            v = p.buffer(f.shape[0], f.shape[2], f.shape[3], f.shape[1])
            i_in.append(p.import_nchw(f.shape, v))
            views.append(v)
        final = build_any_decoder(p, dec, views)
        i_out = {}
        for i, v in final.items():
            if getattr(dec, "depth_head", False):
                i_out[i] = p.head(v, dec.convs[f"output_{i}"][1], torch.empty(1, device=feats[0].device))
            else:
                i_out[i] = p.export_nchw(v)
        p.schedule()
        ent = (p, i_in, i_out, final)
        cache.put(key, ent)
    p, i_in, i_out, final = ent
    for i, f in zip(i_in, feats):
        p.set_in(i, f)
    res = {}
    for i, v in final.items():
        ch = 1 if getattr(dec, "depth_head", False) else v.C
        t = torch.empty(v.N, ch, v.H, v.W, device=feats[0].device, dtype=torch.float32)
        p.set_out(i_out[i], t)
        res[dec.out_key.format(i)] = t
    p.run()
    return res


def binary_mlp_forward(net, inputs, max_scale_only):
    from .mlp import binary_mlp_forward as f

    return f(net, inputs, max_scale_only)


# --- matching-encoder head (reference networks.py:279-283) ---------------------------------
def norm_on_load_eligible(p: Plan, x: View, conv: nn.Conv2d, pad_mode: int) -> bool:
    """Can ``conv`` read ``x`` through a fused InstanceNorm (idh_conv_src.norm)?  LDS-staged fp32 3x3 kernel, 16-channel
    output tiles, whole 16-channel input blocks."""
    return (p.math == "fp32" and x.C % 16 == 0 and lds_subtiles(conv.out_channels) == 1 and
            lds_eligible([(x, conv)], conv.out_channels, x.W, pad_mode))


def build_matching_head(p: Plan, enc, x: Optional[View], nchw_shape=None):
    """Head of ResnetMatchingEncoder (reference modules/networks.py:279-283) on an NHWC view ``x`` — or, with
    ``nchw_shape`` = (N, C, H, W) and ``x`` = None, directly on the backbone's dense NCHW tensor: the first 1x1 conv then
    reads it in place (``Plan.pointwise_nchw``) and the index of that op is returned next to the output view so the caller
    can patch the source pointer."""
    c1, c2 = enc.net[5], enc.net[8]
    if nchw_shape is not None:
        N, _, H, W = [int(v) for v in nchw_shape]
        h = p.buffer(N, H, W, c1.out_channels)
        i_in = p.pointwise_nchw(nchw_shape, c1, h)
        x = h  # dims only
    else:
        h = p.buffer(x.N, x.H, x.W, c1.out_channels)
        p.conv(x, c1, h)
    y = p.buffer(x.N, x.H, x.W, c2.out_channels)
    if FUSE_HEAD_NORM and norm_on_load_eligible(p, h, c2, PAD_REPLICATE):
        # InstanceNorm2d(128) + LeakyReLU: one statistics pass; the 3x3 conv normalises while staging its halo
        stats = p.instance_norm(h, None)
        p.conv(h, c2, y, pad_mode=PAD_REPLICATE, norm=(stats, ACT_LRELU, 0.2))
    else:
        hn = p.buffer(x.N, x.H, x.W, c1.out_channels)
        p.instance_norm(h, hn, act=ACT_LRELU, slope=0.2)
        p.conv(hn, c2, y, pad_mode=PAD_REPLICATE)
    out = p.buffer(x.N, x.H, x.W, c2.out_channels)
    p.instance_norm(y, out)
    return out if nchw_shape is None else (out, i_in)


def matching_head_forward(enc, feat_nchw: torch.Tensor, channels_last: bool = False) -> torch.Tensor:
    _check_in(feat_nchw)
    x = feat_nchw.contiguous()
    key = ("mh", tuple(x.shape), str(x.device), channels_last, _param_key(enc.net[5]) + _param_key(enc.net[8]))
    cache = _plan_cache(enc)
    ent = cache.get(key)
    if ent is None:
        p = Plan(x.device, math=math_of(enc))
        N, Cc, H, W = x.shape
        if FUSE_HEAD_IMPORT and Plan.pointwise_nchw_eligible(enc.net[5]):
            y, i_in = build_matching_head(p, enc, None, nchw_shape=x.shape)
        else:
            xin = p.buffer(N, H, W, Cc)
            i_in = p.import_nchw(x.shape, xin)
            y = build_matching_head(p, enc, xin)
        i_out = None if channels_last else p.export_nchw(y)
        p.schedule()
        ent = (p, i_in, i_out, y)
        cache.put(key, ent)
    p, i_in, i_out, y = ent
    p.set_in(i_in, x)
    if channels_last:
        p.run()
        return y.dense().clone()
    out = torch.empty(y.N, y.C, y.H, y.W, device=x.device, dtype=torch.float32)
    p.set_out(i_out, out)
    p.run()
    return out


# --- SkipDecoder / SkipDecoderRegression (reference modules/networks_fast.py:10-145) ------------
def _conv_block(p: Plan, x: View, blk, out: Optional[View] = None) -> View:
    """ConvBlock: conv3x3 -> ELU -> conv3x3 -> ELU (networks_fast.py:10-28); ReLU (= LeakyReLU with slope 0) with use_elu=False."""
    act, slope = (ACT_LRELU, 0.0) if isinstance(blk.non_lin, nn.ReLU) else (ACT_ELU, 0.2)
    h = p.buffer(x.N, x.H, x.W, blk.conv1.out_channels)
    p.conv(x, blk.conv1, h, act=act, slope=slope)
    if out is None:
        out = p.buffer(x.N, x.H, x.W, blk.conv2.out_channels)
    p.conv(h, blk.conv2, out, act=act, slope=slope)
    p.release(h)
    return out


def build_skip_decoder(p: Plan, dec, feats: List[View]) -> Dict[int, View]:
    """features[-1] -> block1..block4, each: ConvBlock, nearest x2, concat skip, ConvBlock."""
    x = feats[-1]
    final: Dict[int, View] = {}
    for bi, blk in enumerate((dec.block1, dec.block2, dec.block3, dec.block4)):
        skip = feats[-2 - bi]
        cout = blk.pre_concat_conv.conv2.out_channels
        y = _conv_block(p, x, blk.pre_concat_conv)
        if (y.H * 2, y.W * 2) != (skip.H, skip.W):
            raise _lib.IdhError("decoder pyramid levels must differ by exactly x2")
        cat = p.buffer(skip.N, skip.H, skip.W, cout + skip.C)
        p.upsample2(y, cat.slice(0, cout), nearest=True)
        p.copy(skip, cat.slice(cout, skip.C))  # torch.cat([x, cat_feats], 1), networks_fast.py:44
        x = _conv_block(p, cat, blk.post_concat_conv)
        final[3 - bi] = x
    return final


def build_any_decoder(p: Plan, dec, feats: List[View]) -> Dict[int, View]:
    """UNet++ (BDDecoderPP / DepthDecoderPP) or skip decoder, by the structure of ``dec``."""
    if hasattr(dec, "block1"):
        return build_skip_decoder(p, dec, feats)
    return build_decoder(p, dec, feats)


def build_regression_heads(p: Plan, dec, final: Dict[int, View]) -> Dict[int, "tuple"]:
    """SkipDecoderRegression heads (networks_fast.py:106-145): 1x1 conv -> ELU -> 1x1 conv -> ELU ->
    1x1 conv to one channel, per scale.  Returns {scale: (View of the 128-ch penultimate map, last Conv2d)}."""
    heads = {}
    for i, seq in ((3, dec.out1), (2, dec.out2), (1, dec.out3), (0, dec.out4)):
        v = final[i]
        h1 = p.buffer(v.N, v.H, v.W, seq[0].out_channels)
        p.conv(v, seq[0], h1, act=ACT_ELU)
        h2 = p.buffer(v.N, v.H, v.W, seq[2].out_channels)
        p.conv(h1, seq[2], h2, act=ACT_ELU)
        heads[i] = (h2, seq[4])
    return heads


def skip_regression_forward_nchw(dec, input_features: List[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """SkipDecoderRegression.forward (networks_fast.py:137-145): feature maps + log-depth heads."""
    _check_in(*input_features)
    feats = [f.contiguous() for f in input_features]
    key = ("skipreg", tuple(tuple(f.shape) for f in feats), str(feats[0].device), _param_key(dec))
    cache = _plan_cache(dec)
    ent = cache.get(key)
    if ent is None:
        p = Plan(feats[0].device, math=math_of(dec))
        views, i_in = [], []
        for f in feats:
            v = p.buffer(f.shape[0], f.shape[2], f.shape[3], f.shape[1])
            i_in.append(p.import_nchw(f.shape, v))
            views.append(v)
        final = build_skip_decoder(p, dec, views)
        heads = build_regression_heads(p, dec, final)
        i_feat = {i: p.export_nchw(v) for i, v in final.items()}
        i_head = {i: p.head(hv, last, torch.empty(1, device=feats[0].device)) for i, (hv, last) in heads.items()}
        p.schedule()
        ent = (p, i_in, i_feat, i_head, final)
        cache.put(key, ent)
    p, i_in, i_feat, i_head, final = ent
    for i, f in zip(i_in, feats):
        p.set_in(i, f)
    res = {}
    dev = feats[0].device
    for i, v in final.items():
        t = torch.empty(v.N, v.C, v.H, v.W, device=dev)
        p.set_out(i_feat[i], t)
        res[f"feature_s{i}_b1hw"] = t
        d = torch.empty(v.N, 1, v.H, v.W, device=dev)
        p.set_out(i_head[i], d)
        res[f"log_depth_pred_s{i}_b1hw"] = d
    p.run()
    return res
