"""Swap the hot-path modules of a reference model for the MI355X drop-ins.

``convert(model)`` does for all four modules what the reference's own
``model.cost_volume = model.cost_volume.to_fast()`` (test_bd.py:80-81) does for one: it builds
the drop-in twin *from the structure of the module it replaces* (channel counts are read off
the layers, so no Options object is needed), copies the state_dict (names are identical), and
assigns it over the attribute.  ``BDModel.forward`` / ``DepthModel.forward`` then run unchanged.
"""
from __future__ import annotations

import torch
from torch import nn

from . import cost_volume as cv
from . import networks as net
from .pipeline import HotPath


def _device(m: nn.Module):
    for t in list(m.parameters()) + list(m.buffers()):
        return t.device
    return None


def infer_volume_dims(n_in: int):
    """(num_source_views, matching_dim_size) from the input width of the feature-volume MLP, C (K+1) + 10 K + 4
    (reference cost_volume.py:405-423): the reference keeps neither number on the module.  Unique for C in {16, 32} and
    K <= 16 (the first coincidence, 540 = 16*21 + 204 = 32*13 + 124, needs 20 views)."""
    for C in (16, 32):
        K, r = divmod(n_in - C - 4, C + 10)
        if r == 0 and 1 <= K <= 16:
            return K, C
    raise ValueError(f"cannot infer the number of source views / matching channels from an MLP input width of {n_in}")


def convert_cost_volume(ref: nn.Module) -> nn.Module:
    name = type(ref).__name__
    H, W, D = ref.matching_height, ref.matching_width, ref.num_depth_bins
    if name in ("CostVolumeManager", "EfficientCostVolumeManager"):
        new = cv.CostVolumeManager(H, W, D)
    elif name == "ZeroCostVolumeManager":
        new = cv.ZeroCostVolumeManager(H, W, D)
    elif name in ("FeatureVolumeManager", "FastFeatureVolumeManager"):
        K, C = infer_volume_dims(ref.mlp.net[0].in_features)
        new = cv.FeatureVolumeManager(H, W, D, num_source_views=K, matching_dim_size=C)
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


def fused_forward(model: nn.Module, math: str = None):
    """A replacement for ``BDModel.forward`` / ``DepthModel.forward`` at inference time (bd_model.py:175-311,
    depth_model.py:280-440): same arguments, same output dictionary, but everything between the third-party backbones
    and the outputs runs as ONE fused pass of ``HotPath`` (matching-encoder head, volume, CVEncoder, decoder, occlusion
    MLP / depth heads) instead of module by module.  The third-party image encoder and ResNet18 stem still run as the
    model's own torch modules.  ``model.forward = fused_forward(model)`` installs it; the model's config / checkpoint
    surface is untouched (the hot-path modules are converted in place and share their parameters with the pipeline).
    Side effects on the input dicts are the reference's: ``cur_data["prior_mask"]`` and, with ``bd_edge_regularision``,
    ``cur_data["edge_mask"]`` (computed by the reference's own ``get_edge_mask``, which must then be importable)."""
    from . import _lib

    hot = hot_path_of(model, math=math)
    is_bd = hasattr(model, "binary_mlp")
    opts = model.run_opts
    if getattr(opts, "matching_scale", 1) != 1:
        raise _lib.IdhError("the fused forward covers matching_scale = 1: the only value the reference's BDModel / DepthModel constructors accept "
                            "(0 and 2 raise IndexError in their channel lists, bd_model.py:75-83) and the one every shipped configuration uses")

    def forward(phase, cur_data, src_data, unbatched_matching_encoder_forward=False, return_mask=False, infer_depth=False, infer_res=None):
        if phase == "train":
            raise _lib.IdhError("the fused forward is an inference path (no autograd, no flip augmentation)")
        del infer_res  # unused by the reference as well
        ms = opts.matching_scale
        cur_image, src_image = cur_data["image_b3hw"], src_data["image_b3hw"]
        src_K, cur_invK = src_data[f"K_s{ms}_b44"], cur_data[f"invK_s{ms}_b44"]
        # relative poses, bd_model.py:196-204
        src_cam_T_cur_cam = src_data["cam_T_world_b44"] @ cur_data["world_T_cam_b44"].unsqueeze(1)
        cur_cam_T_src_cam = cur_data["cam_T_world_b44"].unsqueeze(1) @ src_data["world_T_cam_b44"]
        hot.thresholder = getattr(model, "thresholder", None)
        with torch.inference_mode():
            cur_feats = list(model.encoder(cur_image))  # third-party image encoder (strong image prior), bd_model.py:218
            kw = {}
            mc = msrc = None
            if hot.matching_model is not None:
                # third-party stem on frame b's current image followed by its K source images (bd_model.py:149-160);
                # the encoder head runs inside the pipeline
                frames = torch.cat([cur_image.unsqueeze(1), src_image], 1)
                stem = model.matching_model.net[:5]
                flat = frames.flatten(0, 1)
                l1 = torch.cat([stem(f) for f in flat.split(1, 0)], 0) if unbatched_matching_encoder_forward else stem(flat)
                kw["matching_layer1"] = l1.unflatten(0, frames.shape[:2])
            else:
                mc, msrc = model.compute_matching_feats(cur_image, src_image, unbatched_matching_encoder_forward)
            if is_bd:
                kw["rendered_depth"] = cur_data["rendered_depth"]
                kw["infer_depth"] = infer_depth
                if getattr(opts, "use_prior", False) and cur_data.get("prior_prediction", None) is not None:
                    kw["prior_inputs"] = {k: cur_data[k] for k in ("prior_prediction", "world_T_cam_b44", "prior_cam_T_world", "K_s0_b44", "invK_s0_b44")}
            out = hot(mc, msrc, cur_feats, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK, return_mask=return_mask, **kw)
        if "prior_mask" in out:
            cur_data["prior_mask"] = out.pop("prior_mask")  # run_mlp_val stores it on the inputs (bd_model.py:431)
        if is_bd and getattr(opts, "bd_edge_regularision", False) and "depth_b1hw" in cur_data:
            # run_mlp_val's other side effect (bd_model.py:444-447): the edge mask compute_binary_losses reads.  It is the
            # reference's own loss-side helper (utils/generic_utils.py:286, needs kornia) — not part of this path
            try:
                from utils.generic_utils import get_edge_mask
            except ImportError as e:
                raise _lib.IdhError("run_opts.bd_edge_regularision needs the reference's utils.generic_utils.get_edge_mask (kornia) on the "
                                    "import path to fill inputs['edge_mask']; unset the option for pure inference") from e
            cur_data["edge_mask"] = get_edge_mask(cur_data["depth_b1hw"])
        return out

    return forward
