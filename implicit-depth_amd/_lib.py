"""ctypes binding of lib/libidh.so (the C ABI in include/idh.h).

There is deliberately NO fallback: if the library is missing or a call fails the product
path raises.  ``import torch`` happens before the CDLL load so that libidh.so's
``libamdhip64.so.7`` dependency resolves to the HIP runtime PyTorch-ROCm already mapped
(one runtime per process: device pointers and streams are then interchangeable).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDH_LIB: developer override used by the ablation builds of tools/abl_split.sh
LIB_PATH = os.environ.get("IDH_LIB") or os.path.join(_HERE, "lib", "libidh.so")

_lib = None
MIN_ABI_VERSION = 105

f32p = C.c_void_p  # device pointers travel as integers


class IdhError(RuntimeError):
    pass


class VolumeOpts(C.Structure):
    """ctypes mirror of ``idh_volume_opts`` (include/idh.h)."""

    _fields_ = [("cur_batch_stride", C.c_int64), ("src_batch_stride", C.c_int64), ("planes", C.c_void_p),
                ("planes_batch_stride", C.c_int64), ("planes_plane_stride", C.c_int64), ("planes_pixel_stride", C.c_int32),
                ("kernel", C.c_int32), ("scratch", C.c_void_p), ("scratch_floats", C.c_int64), ("struct_size", C.c_int64)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(VolumeOpts)


CV_KERNEL_LANE, CV_KERNEL_QUAD, CV_KERNEL_WINDOW = 1, 2, 3  # IDH_CV_KERNEL_* of include/idh.h


_SIGS = {
    "idh_version": (C.c_int, []),
    "idh_sizeof_volume_opts": (C.c_size_t, []),
    "idh_cost_volume_dot_scratch_floats": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "idh_error_string": (C.c_char_p, [C.c_int]),
    "idh_nchw_to_nhwc_f32": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "idh_nhwc_to_nchw_f32": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "idh_packed_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "idh_pack_conv_weight": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "idh_packed_split_weight_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "idh_pack_conv_weight_split": (C.c_int, [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "idh_packed_wino_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "idh_pack_conv_weight_wino": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_void_p]),
    "idh_packed_wino4_weight_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "idh_pack_conv_weight_wino4": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_void_p]),
    "idh_sizeof_op": (C.c_size_t, []),
    "idh_run_ops": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "idh_count_launches": (C.c_int, [C.c_void_p, C.c_int]),
    "idh_packed_mlp_weight_floats": (C.c_size_t, [C.c_int]),
    "idh_pack_mlp_weight": (C.c_int, [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "idh_binary_mlp_fwd": (C.c_int, [f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "idh_binary_mlp_strided_fwd": (C.c_int, [f32p, C.c_longlong, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "idh_binary_mlp_f16x3_fwd": (C.c_int, [f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "idh_feature_volume_workspace_bytes": (C.c_size_t, [C.c_int]),
    "idh_feature_volume_fwd": (
        C.c_int,
        [f32p] * 6 + [C.c_float, C.c_float] + [C.c_int] * 6 + [f32p] * 6 + [f32p, C.c_int, f32p, C.c_void_p, f32p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "idh_feature_volume_f16x3_fwd": (
        C.c_int,
        [f32p] * 6 + [C.c_float, C.c_float] + [C.c_int] * 6 + [f32p] * 6 + [f32p, C.c_int, f32p, C.c_void_p, f32p, C.c_void_p, C.c_size_t, C.c_void_p],
    ),
    "idh_feature_volume_ex_fwd": (
        C.c_int,
        [f32p] * 6 + [C.c_float, C.c_float] + [C.c_int] * 6 + [f32p] * 6 + [f32p, C.c_int, f32p, C.c_void_p, f32p, C.c_void_p, C.c_size_t,
                                                                  C.c_int, C.POINTER(VolumeOpts), C.c_void_p],
    ),
    "idh_cost_volume_dot_ex_fwd": (
        C.c_int,
        [f32p, f32p, f32p, f32p, f32p, C.c_float, C.c_float] + [C.c_int] * 6 + [f32p, C.c_int, f32p, f32p, C.POINTER(VolumeOpts), C.c_void_p],
    ),
    "idh_cost_volume_dot_kernel_name": (C.c_char_p, [C.c_int] * 5),
    "idh_packed_mlp_weight_f16_bytes": (C.c_size_t, [C.c_int]),
    "idh_pack_mlp_weight_f16": (C.c_int, [f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "idh_binary_mlp_search_fwd": (C.c_int, [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.c_float, C.c_float, f32p, f32p, C.c_void_p]),
    "idh_binary_mlp_search_thr_fwd": (C.c_int, [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_float, f32p, f32p, C.c_int, f32p, f32p, C.c_void_p]),
    "idh_binary_mlp_search_f16x3_fwd": (C.c_int, [f32p, C.c_int, C.c_int, f32p, C.c_int, C.c_float, f32p, C.c_void_p, f32p, C.c_int, C.c_int, C.c_int,
                                                  C.c_float, C.c_float, C.c_float, f32p, f32p, C.c_int, f32p, f32p, C.c_void_p]),
    "idh_metrics_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "idh_plane_iou_fwd": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "idh_depth_metrics_fwd": (C.c_int, [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "idh_sample_prior_fwd": (C.c_int, [f32p, f32p, C.c_int, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p]),
    "idh_cost_volume_dot_fwd": (
        C.c_int,
        [f32p, f32p, f32p, f32p, f32p, C.c_float, C.c_float] + [C.c_int] * 6 + [f32p, C.c_int, f32p, f32p, C.c_void_p],
    ),
}


def _all_sigs():
    from . import net_abi  # (structs of include/idh_net.h live there; imported lazily: net_abi imports this module)

    return {**_SIGS, **net_abi.SIGS}


def declared_symbols():
    return sorted(_all_sigs())


def lib():
    """Load (once) and return the ctypes handle; raises IdhError if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IdhError(
                f"{LIB_PATH} not found — the gfx950 HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)."
            )
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _all_sigs().items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        # the ABI this mirror was written against (include/idh.h): struct layouts, tile codes and packed-weight layouts changed at these versions
        # (102: IDH_TILE_WINO4 names the shared-transform F(4x4) kernel and ITS packed layout - blobs packed by an older library are not portable)
        ver = h.idh_version()
        if ver < MIN_ABI_VERSION and not os.environ.get("IDH_LIB_ANY_ABI"):  # (IDH_LIB_ANY_ABI: A/B timing against a library built from an older tree)
            raise IdhError(f"{LIB_PATH} reports ABI version {ver}, this binding needs >= {MIN_ABI_VERSION}: rebuild with `python implicit-depth_amd/build.py --force`")
        if h.idh_sizeof_volume_opts() != C.sizeof(VolumeOpts):
            raise IdhError(f"{LIB_PATH}: sizeof(idh_volume_opts) = {h.idh_sizeof_volume_opts()} in the library, {C.sizeof(VolumeOpts)} in this binding")
        _lib = h
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().idh_error_string(code).decode()
        raise IdhError(f"{what} failed: {msg} ({code})")


def ptr(t: "torch.Tensor | None"):
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_weights_epoch = 0


def invalidate_weight_caches() -> None:
    """Forget every packed-weight / plan cache entry keyed on parameters whose modifications torch does not count
    (tensors created under ``torch.inference_mode()``).  Called automatically after ``load_state_dict`` on the drop-in
    modules and ``HotPath``; call it by hand after any other in-place edit of inference-mode parameters
    (``param.copy_()``, ``param.mul_()`` ... inside ``with torch.inference_mode():``)."""
    global _weights_epoch
    _weights_epoch += 1


def _sd_post_hook(module, incompatible_keys) -> None:
    """load_state_dict post-hook (a module-level function, not a closure: a module carrying it still pickles)."""
    invalidate_weight_caches()


def watch_state_dict_loads(module) -> None:
    """``load_state_dict()`` on the module, on any parent, or on ANY of its submodules (e.g. straight into ``cost_volume.mlp`` or a
    ``Conv2d`` inside a decoder's ``ModuleDict``) invalidates the caches above: load_state_dict runs the post-hooks of every module
    it descends into, so the hook is registered on each of them."""
    for m in module.modules():
        if not m.__dict__.get("_idh_sd_hook"):
            m.register_load_state_dict_post_hook(_sd_post_hook)
            m.__dict__["_idh_sd_hook"] = True


def param_version(t):
    """Cache key component that changes when parameter ``t`` is modified in place: torch's version counter — or, for
    tensors created under ``torch.inference_mode()`` (no counter, yet ``load_state_dict`` / ``copy_`` inside inference mode DO
    modify them), the epoch bumped by ``invalidate_weight_caches()``."""
    try:
        return t._version
    except RuntimeError:
        return ("inference", _weights_epoch)


def require_cuda_f32(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise IdhError("implicit_depth_amd kernels need tensors on the MI355X (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise IdhError(f"implicit_depth_amd kernels are fp32 (got {t.dtype})")
