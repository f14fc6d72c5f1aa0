"""Drop-ins for the conv / MLP blocks around the cost volume (reference modules/networks.py).

Module trees and parameter names are identical to the reference's (``convs.ds_conv_0.conv1``,
``convs.in_conv_01.conv_0.conv2``, ``mlps.s0.0`` …) so reference checkpoints load with
``load_state_dict`` unchanged; every forward runs hand-written gfx950 kernels on NHWC
activations (see nhwc.py for the execution plans).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
from torch import nn

from . import _lib
from .layers import BasicBlock


def double_basic_block(num_ch_in, num_ch_out, num_repeats=2):
    layers = nn.Sequential(BasicBlock(num_ch_in, num_ch_out))
    for i in range(num_repeats - 1):
        layers.add_module(f"conv_{i}", BasicBlock(num_ch_out, num_ch_out))
    return layers


class CVEncoder(nn.Module):
    """reference modules/networks.py:186-215"""

    def __init__(self, num_ch_cv, num_ch_enc, num_ch_outs):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        self.convs = nn.ModuleDict()
        self.num_ch_enc = []
        self.num_blocks = len(num_ch_outs)
        self.num_ch_img = list(num_ch_enc)
        for i in range(self.num_blocks):
            cin = num_ch_cv if i == 0 else num_ch_outs[i - 1]
            cout = num_ch_outs[i]
            self.convs[f"ds_conv_{i}"] = BasicBlock(cin, cout, stride=1 if i == 0 else 2)
            self.convs[f"conv_{i}"] = nn.Sequential(BasicBlock(num_ch_enc[i] + cout, cout), BasicBlock(cout, cout))
            self.num_ch_enc.append(cout)

    def forward(self, x, img_feats):
        from .nhwc import cv_encoder_forward_nchw

        return cv_encoder_forward_nchw(self, x, img_feats)


class _DecoderPP(nn.Module):
    """UNet++ grid shared by BDDecoderPP / DepthDecoderPP (reference :20-84 / :118-183)."""

    depth_head = False
    out_key = "feature_s{}_b1hw"

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([64, 64, 128, 256])
        self.convs = nn.ModuleDict()
        for j in range(1, 5):
            for i in range(4 - j, -1, -1):
                cout = int(self.num_ch_dec[i])
                total = 0
                cin = num_ch_enc[i + 1] if j == 1 else int(self.num_ch_dec[i + 1])
                self.convs[f"diag_conv_{i + 1}{j - 1}"] = BasicBlock(cin, cout)
                total += cout
                cin = num_ch_enc[i] if j == 1 else int(self.num_ch_dec[i])
                self.convs[f"right_conv_{i}{j - 1}"] = BasicBlock(cin, cout)
                total += cout
                if i + j != 4:
                    self.convs[f"up_conv_{i + 1}{j}"] = BasicBlock(int(self.num_ch_dec[i + 1]), cout)
                    total += cout
                self.convs[f"in_conv_{i}{j}"] = double_basic_block(total, cout)
                head = [BasicBlock(cout, cout) if i != 0 else nn.Identity()]
                if self.depth_head:
                    head.append(nn.Conv2d(cout, self.num_output_channels, 1))
                self.convs[f"output_{i}"] = nn.Sequential(*head)

    def forward(self, input_features):
        from .nhwc import decoder_forward_nchw

        return decoder_forward_nchw(self, input_features)


class BDDecoderPP(_DecoderPP):
    depth_head = False
    out_key = "feature_s{}_b1hw"

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=16, use_skips=True):
        super().__init__(num_ch_enc, scales, num_output_channels, use_skips)


class DepthDecoderPP(_DecoderPP):
    depth_head = True
    out_key = "log_depth_pred_s{}_b1hw"


class MLP(nn.Module):
    """reference modules/networks.py:218-233 (LeakyReLU default slope 0.01)."""

    def __init__(self, channel_list, disable_final_activation=False):
        super().__init__()
        layers = []
        for i in range(len(channel_list) - 1):
            layers.append(nn.Linear(channel_list[i], channel_list[i + 1]))
            layers.append(nn.LeakyReLU(inplace=True))
        if disable_final_activation:
            layers = layers[:-1]
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        """reference networks.py:232-233.  Not on the hot path (the fused feature-volume kernel reads the
        weights directly); kept so the attribute behaves like the reference's module."""
        return self.net(x)


class BinaryMLPNetwork(nn.Module):
    """reference modules/networks.py:87-115 (only scale 0 is evaluated at test time)."""

    def __init__(self, input_channels, mlp_size=128, use_prior=False):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        self.scales = list(range(4))
        self.use_prior = use_prior
        extra = 2 if use_prior else 1
        self.mlps = nn.ModuleDict()
        for scale, ch in enumerate(input_channels):
            self.mlps[f"s{scale}"] = nn.Sequential(
                nn.Linear(int(ch) + extra, mlp_size), nn.ELU(inplace=True), nn.Linear(mlp_size, mlp_size), nn.ELU(inplace=True), nn.Linear(mlp_size, 1)
            )

    def forward(self, inputs: List[torch.Tensor], max_scale_only: bool = False) -> Dict[str, torch.Tensor]:
        from .nhwc import binary_mlp_forward

        return binary_mlp_forward(self, inputs, max_scale_only)


class ResnetMatchingEncoder(nn.Module):
    """Matching encoder (reference modules/networks.py:236-287).

    ``net[0:5]`` — conv1/bn1/relu/maxpool/layer1 of antialiased_cnns.resnet18 — is third-party
    code that is not part of the reference tree: the caller passes those five
    modules in (or any stand-in with 64 output channels at 1/4 resolution) and they run as
    ordinary torch modules.  The head ``net[5:10]`` (1x1 conv 64->128, InstanceNorm, LeakyReLU(0.2),
    3x3 replicate-padded conv 128->C, InstanceNorm) runs on the gfx950 kernels and can hand its
    output over channels-last, which is the layout the cost-volume kernels consume.
    State-dict keys are the reference's (``net.5.weight`` … ``net.8.bias``)."""

    def __init__(self, backbone_modules, num_ch_out: int = 16, backbone_channels: int = 64):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        backbone_modules = list(backbone_modules)
        if len(backbone_modules) != 5:
            raise ValueError("expected the 5 backbone modules conv1, bn1, relu, maxpool, layer1")
        self.num_ch_out = num_ch_out
        self.net = nn.Sequential(
            *backbone_modules,
            nn.Conv2d(backbone_channels, 128, (1, 1)),
            nn.InstanceNorm2d(128),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(128, num_ch_out, (3, 3), padding=1, padding_mode="replicate"),
            nn.InstanceNorm2d(num_ch_out),
        )

    def backbone(self, x):
        for i in range(5):
            x = self.net[i](x)
        return x

    def forward(self, input_image, channels_last: bool = False):
        from .nhwc import matching_head_forward

        return matching_head_forward(self, self.backbone(input_image), channels_last)


# --- skip decoder family (reference modules/networks_fast.py) ---------------------------------
class ConvBlock(nn.Module):
    """conv3x3 -> ELU -> conv3x3 -> ELU (networks_fast.py:10-28); ``use_elu=False``: ReLU.  ``use_bn`` is accepted and — exactly
    as in the reference, whose ConvBlock never creates a normalisation layer (networks_fast.py:11-19) — has no effect."""

    def __init__(self, in_ch, out_ch, use_elu=True, use_bn=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.non_lin = nn.ELU(inplace=True) if use_elu else nn.ReLU(inplace=True)


class ConvUpsampleAndConcatBlock(nn.Module):
    """ConvBlock -> nearest x2 -> cat(skip) -> ConvBlock (networks_fast.py:31-47)."""

    def __init__(self, in_ch, out_ch, skip_chns, use_elu=True, use_bn=False):
        super().__init__()
        self.pre_concat_conv = ConvBlock(in_ch, out_ch, use_elu, use_bn)
        self.post_concat_conv = ConvBlock(out_ch + skip_chns, out_ch, use_elu, use_bn)


class SkipDecoder(nn.Module):
    """reference modules/networks_fast.py:49-99 (``depth_decoder_name: skip``)."""

    depth_head = False
    out_key = "feature_s{}_b1hw"

    def __init__(self, input_channels, use_bn=False):
        super().__init__()
        _lib.watch_state_dict_loads(self)  # load_state_dict invalidates packed-weight caches of inference-mode parameters
        input_channels = list(input_channels)[::-1]
        self.input_channels = input_channels
        self.output_channels = [256, 128, 64, 64]
        self.num_ch_dec = self.output_channels[::-1]
        for i in range(4):
            # in_ch = input_channels[i] exactly as the reference (:59-82): it relies on the encoder widths
            # [.., 64, 128, 256, 384] coinciding with the previous block's output width
            setattr(self, f"block{i + 1}", ConvUpsampleAndConcatBlock(input_channels[i], self.output_channels[i], input_channels[i + 1], use_bn=use_bn))

    def forward(self, features):
        from .nhwc import decoder_forward_nchw

        return decoder_forward_nchw(self, features)


class SkipDecoderRegression(SkipDecoder):
    """reference modules/networks_fast.py:102-145: SkipDecoder + per-scale 1x1 MLP heads."""

    def __init__(self, input_channels, use_bn=False):
        super().__init__(input_channels, use_bn=use_bn)
        for i in range(4):
            setattr(self, f"out{i + 1}", nn.Sequential(nn.Conv2d(self.output_channels[i], 128, 1), nn.ELU(inplace=True), nn.Conv2d(128, 128, 1),
                                                       nn.ELU(inplace=True), nn.Conv2d(128, 1, 1)))

    def forward(self, features):
        from .nhwc import skip_regression_forward_nchw

        return skip_regression_forward_nchw(self, features)
