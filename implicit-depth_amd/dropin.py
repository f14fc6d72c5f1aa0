"""Swap the hot-path modules of a reference model for the MI355X drop-ins.

``convert(model)`` does for all four modules what the reference's own
``model.cost_volume = model.cost_volume.to_fast()`` (test_bd.py:80-81) does for one: it builds
the drop-in twin *from the structure of the module it replaces* (channel counts are read off
the layers, so no Options object is needed), copies the state_dict (names are identical), and
assigns it over the attribute.  ``BDModel.forward`` / ``DepthModel.forward`` then run unchanged.
"""
from __future__ import annotations

from torch import nn

from . import cost_volume as cv
from . import networks as net
from .pipeline import HotPath


def _device(m: nn.Module):
    for t in list(m.parameters()) + list(m.buffers()):
        return t.device
    return None


def convert_cost_volume(ref: nn.Module) -> nn.Module:
    name = type(ref).__name__
    H, W, D = ref.matching_height, ref.matching_width, ref.num_depth_bins
    if name in ("CostVolumeManager", "EfficientCostVolumeManager"):
        new = cv.CostVolumeManager(H, W, D)
    elif name == "ZeroCostVolumeManager":
        new = cv.ZeroCostVolumeManager(H, W, D)
    elif name in ("FeatureVolumeManager", "FastFeatureVolumeManager"):
        n_in = ref.mlp.net[0].in_features  # 16(K+1) + 10K + 4 (reference cost_volume.py:405-423)
        if (n_in - 20) % 26:
            raise ValueError(f"cannot infer the number of source views from an MLP input width of {n_in}")
        new = cv.FeatureVolumeManager(H, W, D, num_source_views=(n_in - 20) // 26)
    else:
        raise ValueError(f"unrecognised cost volume class {name}")
    new.load_state_dict(ref.state_dict())
    return new.to(_device(ref))


def convert_cv_encoder(ref: nn.Module) -> nn.Module:
    n = ref.num_blocks
    outs = [ref.convs[f"ds_conv_{i}"].conv1.out_channels for i in range(n)]
    enc = [ref.convs[f"conv_{i}"][0].conv1.in_channels - outs[i] for i in range(n)]
    new = net.CVEncoder(ref.convs["ds_conv_0"].conv1.in_channels, enc, outs)
    new.load_state_dict(ref.state_dict())
    return new.to(_device(ref))


def convert_decoder(ref: nn.Module) -> nn.Module:
    if hasattr(ref, "block1"):  # SkipDecoder / SkipDecoderRegression (networks_fast.py)
        blocks = [ref.block1, ref.block2, ref.block3, ref.block4]
        outs = [b.pre_concat_conv.conv2.out_channels for b in blocks]
        enc_rev = [ref.block1.pre_concat_conv.conv1.in_channels] + [b.post_concat_conv.conv1.in_channels - o for b, o in zip(blocks, outs)]
        new = (net.SkipDecoderRegression if hasattr(ref, "out1") else net.SkipDecoder)(enc_rev[::-1])
        new.load_state_dict(ref.state_dict())
        return new.to(_device(ref))
    enc = [ref.convs[f"right_conv_{i}0"].conv1.in_channels for i in range(4)] + [ref.convs["diag_conv_40"].conv1.in_channels]
    head = len(ref.convs["output_0"]) == 2
    new = (net.DepthDecoderPP if head else net.BDDecoderPP)(enc)
    new.load_state_dict(ref.state_dict())
    return new.to(_device(ref))


def convert_binary_mlp(ref: nn.Module) -> nn.Module:
    scales = sorted(ref.mlps.keys())
    first = ref.mlps["s0"][0]
    use_prior = getattr(ref, "use_prior", None)
    widths = [ref.mlps[s][0].in_features for s in scales]
    if use_prior is None:
        use_prior = (widths[0] - 64) == 2  # scale-0 features have num_ch_dec[0] = 64 channels (networks.py:30)
    extra = 2 if use_prior else 1
    new = net.BinaryMLPNetwork([w - extra for w in widths], mlp_size=first.out_features, use_prior=use_prior)
    new.load_state_dict(ref.state_dict())
    return new.to(_device(ref))


def convert(model: nn.Module, math: str = None) -> nn.Module:
    """In-place: replace ``cost_volume``, ``cost_volume_net``, ``depth_decoder`` and (BDModel)
    ``binary_mlp`` of a reference model.  Idempotent.  ``math``: None keeps the default (fp32 MFMA); "f16x3" selects the split-precision
    kernels for this model's convs (and, for "f16x3", its MLP feature volume and BinaryMLP)."""
    if not isinstance(model.cost_volume, cv.CostVolumeManager):
        model.cost_volume = convert_cost_volume(model.cost_volume)
    if not isinstance(model.cost_volume_net, net.CVEncoder):
        model.cost_volume_net = convert_cv_encoder(model.cost_volume_net)
    if not isinstance(model.depth_decoder, (net._DecoderPP, net.SkipDecoder)):
        model.depth_decoder = convert_decoder(model.depth_decoder)
    if hasattr(model, "binary_mlp") and not isinstance(model.binary_mlp, net.BinaryMLPNetwork):
        model.binary_mlp = convert_binary_mlp(model.binary_mlp)
    if math is not None:
        from . import nhwc

        if math not in nhwc.MATH_MODES:
            raise ValueError(f"math must be one of {nhwc.MATH_MODES}")
        mlp_math = "f16x3" if math == "f16x3" else "fp32"
        model.cost_volume_net.conv_math = model.depth_decoder.conv_math = math
        if isinstance(model.cost_volume, cv.FeatureVolumeManager):
            model.cost_volume.mlp_math = mlp_math
        if hasattr(model, "binary_mlp"):
            model.binary_mlp.mlp_math = mlp_math
    return model


def hot_path_of(model: nn.Module, min_depth: float = 0.25, max_depth: float = 5.0, math: str = None) -> HotPath:
    """Fused pipeline sharing the (converted) modules of ``model``."""
    convert(model, math)
    o = getattr(model, "run_opts", None)
    if o is not None:
        min_depth, max_depth = o.min_matching_depth, o.max_matching_depth
    mm = getattr(model, "matching_model", None)
    if mm is not None and not (hasattr(mm, "net") and len(mm.net) == 10 and isinstance(mm.net[5], nn.Conv2d) and isinstance(mm.net[8], nn.Conv2d)):
        mm = None  # FPNMatchingEncoder etc.: the caller keeps running it and passes finished matching features
    hot = HotPath(model.cost_volume, model.cost_volume_net, model.depth_decoder, getattr(model, "binary_mlp", None), min_depth, max_depth,
                  conv_math=getattr(model.cost_volume_net, "conv_math", None), matching_model=mm)
    hot.thresholder = getattr(model, "thresholder", None)  # test_bd.py:103 sets it on the model for the infer_depth search
    return hot
