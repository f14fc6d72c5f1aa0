"""Buffer-compatible stand-ins for the reference's geometry modules.

The fused kernels do the back-projection / projection arithmetic themselves; these classes
exist so that a drop-in ``CostVolumeManager`` exposes the same ``state_dict`` buffers as the
reference (``backprojector.pix_coords_13N``, ``projector.eps`` — SURVEY.md §5 checkpoint
row; reference utils/geometry_utils.py:22-89) and loads reference checkpoints unchanged.
"""
import torch
from torch import nn


class BackprojectDepth(nn.Module):
    def __init__(self, height: int, width: int):
        super().__init__()
        self.height, self.width = height, width
        ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
        pix = torch.stack([xs + 0.5, ys + 0.5, torch.ones_like(xs, dtype=torch.float32)], 0)
        self.register_buffer("pix_coords_13N", pix.reshape(1, 3, -1).float())


class Project3D(nn.Module):
    def __init__(self, eps: float = 1e-5):
        super().__init__()
        self.register_buffer("eps", torch.tensor(eps).view(1, 1, 1))
