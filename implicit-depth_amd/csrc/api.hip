#include "idh_common.h"

extern "C" int idh_version(void) { return 105; }  // 101: idh_volume_opts.scratch / scratch_floats / struct_size; 102: Winograd F(4x4) conv (IDH_TILE_WINO4);
                                                   // 103: struct_size accepted when >= the fields it guards, hidden visibility (the C ABI is the only export);
                                                   // 104: IDH_OP_POINTWISE_UP, tile_m 8 / 9 for a lone 3x3 stride-2 source, split-K boundaries of the LDS conv in cost units
                                                   // 105: idh_binary_mlp_fwd takes any feature row stride / 4-byte-aligned base; network-level entry points idh_basic_block_fwd,
                                                   //      idh_cvencoder_fwd, idh_unetpp_fwd (csrc/networks.hip); idh_pack_conv_weight_wino4 row order (w4_v2p); run lists in idh_volume_opts.scratch
extern "C" size_t idh_sizeof_volume_opts(void) { return sizeof(idh_volume_opts); }

extern "C" const char *idh_error_string(int code) {
    switch (code) {
        case IDH_OK: return "ok";
        case IDH_EINVAL: return "invalid argument (shape, null pointer or parameter)";
        case IDH_EUNSUPPORTED: return "configuration not covered by the gfx950 kernels";
        case IDH_ELAUNCH: return "HIP kernel launch failed";
        case IDH_EWORKSPACE: return "workspace missing or too small";
        default: return "unknown idh error";
    }
}
