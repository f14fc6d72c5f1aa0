"""ctypes mirror of include/idh_net.h — the network-level C entry points (``idh_basic_block_fwd``, ``idh_cvencoder_fwd``,
``idh_unetpp_fwd``) — and helpers that describe a drop-in (or reference) module to them.

The Python drop-ins do NOT go through these: ``nhwc.Plan`` builds the same op lists with its run-time-tunable thresholds and caches them.
This module is (a) what the parity tests call to prove that a C host gets the same numbers, and (b) the worked example of the binding a
non-Python host writes.  Reference: modules/layers.py:78-95 (BasicBlock.forward), modules/networks.py:186-215
(CVEncoder), :20-84 / :118-183 (UNet++ decoders).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import _lib

LAYOUT_NHWC, LAYOUT_NCHW = 0, 1
UNETPP_BLOCKS = 49


class Tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("layout", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("cs", C.c_int32)]


class ConvParams(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("ks", C.c_int32), ("stride", C.c_int32)]


class BlockParams(C.Structure):
    _fields_ = [("conv1", ConvParams), ("conv2", ConvParams), ("downsample", ConvParams)]


class NetSizes(C.Structure):
    _fields_ = [("workspace_floats", C.c_size_t), ("weight_floats", C.c_size_t), ("ops", C.c_int32), ("launches", C.c_int32),
                ("wino4", C.c_int32), ("wino2", C.c_int32), ("recycled", C.c_int32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def nhwc(t: Optional[torch.Tensor], C_: int = 0, H: int = 0, W: int = 0, cs: int = 0) -> Tensor:
    """An (N,H,W,cs) tensor (or just its shape, for the size queries) as an NHWC idh_tensor of ``C_`` channels (default: all)."""
    if t is not None:
        _, H, W, cs = t.shape
        C_ = C_ or cs
    return Tensor(t.data_ptr() if t is not None else None, LAYOUT_NHWC, C_, H, W, cs or C_)


def nchw(t: Optional[torch.Tensor], C_: int = 0, H: int = 0, W: int = 0) -> Tensor:
    if t is not None:
        if not t.is_contiguous():
            raise _lib.IdhError("NCHW idh_tensor must be dense")
        _, C_, H, W = t.shape
    return Tensor(t.data_ptr() if t is not None else None, LAYOUT_NCHW, C_, H, W, 0)


def conv_params(conv: Optional[nn.Conv2d], keep: list) -> ConvParams:
    if conv is None:
        return ConvParams(None, None, 0, 0, 0, 0)
    w = conv.weight.detach().contiguous()
    b = conv.bias.detach().contiguous() if conv.bias is not None else None
    keep += [w, b]
    ks = conv.kernel_size[0]
    return ConvParams(w.data_ptr() if w.is_cuda else None, b.data_ptr() if (b is not None and b.is_cuda) else None, conv.out_channels, conv.in_channels, ks,
                      conv.stride[0])


def block_params(blk, keep: list) -> BlockParams:
    """A BasicBlock (drop-in ``layers.BasicBlock`` or the reference's ``modules.layers.BasicBlock``: same attribute names)."""
    ds = blk.downsample[0] if blk.downsample is not None else None
    return BlockParams(conv_params(blk.conv1, keep), conv_params(blk.conv2, keep), conv_params(ds, keep))


def cvencoder_blocks(enc, keep: list):
    """blocks[3 i + 0 / 1 / 2] = ds_conv_i, conv_i[0], conv_i[1] (include/idh_net.h)"""
    out = []
    for i in range(enc.num_blocks):
        seq = enc.convs[f"conv_{i}"]
        out += [block_params(enc.convs[f"ds_conv_{i}"], keep), block_params(seq[0], keep), block_params(seq[1], keep)]
    return (BlockParams * len(out))(*out)


def unetpp_blocks(dec, keep: list):
    """The 49 BasicBlocks of a UNet++ decoder in the order the reference's forward visits them (networks.py:64-84), then output_1..3[0];
    and the four 1x1 heads of a DepthDecoderPP (or None)."""
    out = []
    for j in range(1, 5):
        for i in range(4 - j, -1, -1):
            out.append(block_params(dec.convs[f"right_conv_{i}{j - 1}"], keep))
            out.append(block_params(dec.convs[f"diag_conv_{i + 1}{j - 1}"], keep))
            if i + j != 4:
                out.append(block_params(dec.convs[f"up_conv_{i + 1}{j}"], keep))
            seq = dec.convs[f"in_conv_{i}{j}"]
            out += [block_params(seq[0], keep), block_params(seq.conv_0, keep)]
    for i in (1, 2, 3):
        out.append(block_params(dec.convs[f"output_{i}"][0], keep))
    assert len(out) == UNETPP_BLOCKS
    heads = None
    if len(dec.convs["output_0"]) == 2:
        heads = (ConvParams * 4)(*[conv_params(dec.convs[f"output_{i}"][1], keep) for i in range(4)])
    return (BlockParams * len(out))(*out), heads


def _sigs():
    P = C.POINTER
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    return {
        "idh_basic_block_sizes": (i32, [P(BlockParams), i32, P(Tensor), P(Tensor), P(NetSizes)]),
        "idh_basic_block_pack": (i32, [P(BlockParams), i32, P(Tensor), P(Tensor), vp, vp]),
        "idh_basic_block_fwd": (i32, [P(BlockParams), vp, i32, P(Tensor), P(Tensor), vp, sz, vp]),
        "idh_cvencoder_sizes": (i32, [P(BlockParams), i32, i32, P(Tensor), P(Tensor), P(Tensor), P(NetSizes)]),
        "idh_cvencoder_pack": (i32, [P(BlockParams), i32, i32, P(Tensor), P(Tensor), P(Tensor), vp, vp]),
        "idh_cvencoder_fwd": (i32, [P(BlockParams), i32, vp, i32, P(Tensor), P(Tensor), P(Tensor), vp, sz, vp]),
        "idh_unetpp_sizes": (i32, [P(BlockParams), i32, P(ConvParams), i32, P(Tensor), P(Tensor), P(NetSizes)]),
        "idh_unetpp_pack": (i32, [P(BlockParams), i32, P(ConvParams), i32, P(Tensor), P(Tensor), vp, vp]),
        "idh_unetpp_fwd": (i32, [P(BlockParams), i32, P(ConvParams), vp, i32, P(Tensor), P(Tensor), P(vp), P(vp), vp, sz, vp]),
    }


SIGS = _sigs()


def tensors(ts: Sequence[Tensor]):
    return (Tensor * len(ts))(*ts)


def ptr_array(ts: Optional[List[torch.Tensor]]):
    if ts is None:
        return None
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
