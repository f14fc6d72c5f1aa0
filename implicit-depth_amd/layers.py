"""Residual block drop-in (reference modules/layers.py:34-95).

Parameter names (conv1/conv2/downsample.0) match the reference so checkpoints load unchanged;
the forward runs the gfx950 implicit-GEMM conv kernels (csrc/conv.hip) on NHWC activations.
"""
from __future__ import annotations

from torch import nn

from . import _lib


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=True)
        if inplanes == planes and stride == 1:
            self.downsample = None
        else:
            k = 1 if stride == 1 else 3
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, k, stride=stride, padding=k // 2, bias=True), nn.Identity())
        self.stride = stride
        self.inplanes, self.planes = inplanes, planes

    def forward(self, x):
        from .nhwc import block_forward_nchw

        return block_forward_nchw(self, x)
