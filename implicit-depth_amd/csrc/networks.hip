// Network-level entry points of the conv stage (include/idh_net.h): BasicBlock, CVEncoder, the UNet++ decoders as host-side builders over
// idh_run_ops.  Replaces BasicBlock.forward (reference modules/layers.py:78-95), CVEncoder.forward (modules/networks.py:186-215) and
// BDDecoderPP / DepthDecoderPP.forward (modules/networks.py:64-84, 163-183) for hosts that are not Python.
//
// This file contains NO kernels of its own except a bias adder: it is the C++ twin of implicit-depth_amd/nhwc.py's Plan (fp32 arithmetic, default
// thresholds) — buffers carved out of the caller's workspace, packed weights out of the caller's blob, the same tile selection per conv, the same
// concat elimination (producers write channel slices), the same dependency-level schedule and launch groups, the same liveness reuse of the big
// activation temporaries — so a pass is the same op list the Python drop-ins replay and the results are bit-identical to theirs
// (tests/test_net_abi_gpu.py compares the two, and both with the reference's goldens).
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "idh_common.h"
#include "../../include/idh_net.h"
#include "../../include/idh_ops.h"

namespace {

// ---- the constants of nhwc.py this builder mirrors (the defaults; the Python side can be re-tuned at run time, this side is the shipped setting)
constexpr int kWinoMinTiles = 128;        // WINO_MIN_TILES
constexpr double kWinoMinFill = 0.74;     // WINO_MIN_FILL
constexpr int kWino4MinTiles = 768;       // WINO4_MIN_TILES
constexpr double kWino4MinFill = 0.85;    // WINO4_MIN_FILL
constexpr long long kReuseMinBytes = 64ll << 20;  // REUSE_MIN_BYTES
constexpr int kNarrowTileBelow = 400;     // NARROW_TILE_BELOW
constexpr int kSplitMinChunks = 6, kSplitMax = 16;  // SPLIT_MIN_CHUNKS, SPLIT_MAX
constexpr double kProjChunkWeight = 0.5;  // PROJ_CHUNK_WEIGHT
constexpr int kS2FirstMinBlocks = 512;    // S2_FIRST_MIN_BLOCKS
constexpr int kTargetWaves = 2048, kMinWaves = 1024;
constexpr int kTileWino = IDH_TILE_WINO, kTileWino4 = IDH_TILE_WINO4;

inline int ceil16(int v) { return (v + 15) & ~15; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

__global__ __launch_bounds__(256) void bias_sum_k(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f);
}

enum Mode { MODE_SIZES, MODE_PACK, MODE_RUN };

struct Buf {
    float *base;  // device address of channel 0 of pixel 0 (a fake, aligned address in MODE_SIZES / MODE_PACK)
    int N, H, W, cs;
    bool internal;
    bool zero_fill;  // internal buffer with padding channels: cleared on the stream before the pass
    size_t floats;
};

struct View {
    int buf = -1, c0 = 0, C = 0;
};

struct Region {
    int buf, c0, c1;
};

struct Meta {
    std::vector<Region> reads, writes;
};

struct Src {
    View v;
    const idh_conv_params *cv;
};

class Plan {
  public:
    Plan(Mode mode, float *ws, size_t ws_cap, float *blob, hipStream_t st) : mode_(mode), ws_(ws), ws_cap_(ws_cap), blob_(blob), st_(st) {
        if (mode_ != MODE_RUN) ws_ = reinterpret_cast<float *>(uintptr_t(1) << 32);   // fake, aligned: only offsets matter
        if (mode_ == MODE_SIZES) blob_ = reinterpret_cast<float *>(uintptr_t(1) << 40);
    }

    int err = IDH_OK;
    std::vector<idh_op> ops;
    std::vector<Meta> meta;
    std::vector<Buf> bufs;
    size_t ws_off = 0, blob_off = 0;
    int n_wino4 = 0, n_wino2 = 0, recycled = 0;

    int N(const View &v) const { return bufs[v.buf].N; }
    int H(const View &v) const { return bufs[v.buf].H; }
    int W(const View &v) const { return bufs[v.buf].W; }
    int cs(const View &v) const { return bufs[v.buf].cs; }
    float *ptr(const View &v) const { return bufs[v.buf].base + v.c0; }
    static View slice(const View &v, int c0, int C) { return View{v.buf, v.c0 + c0, C}; }

    float *ws_alloc(size_t floats) {
        float *p = ws_ + ws_off;
        ws_off = align64(ws_off + floats);
        if (mode_ == MODE_RUN && ws_off > ws_cap_) err = IDH_EWORKSPACE;
        return p;
    }
    float *blob_alloc(size_t floats) {
        float *p = blob_ + blob_off;
        blob_off = align64(blob_off + floats);
        return p;
    }

    // Dense NHWC buffer of the plan (Plan.buffer): channel counts that are not a multiple of 16 get zero padding channels
    View buffer(int n, int h, int w, int c) {
        const int cs_ = ceil16(c);
        if (cs_ == c) {
            auto it = free_.find(std::make_tuple(n, h, w, cs_));
            if (it != free_.end() && !it->second.empty()) {
                const int b = it->second.back();
                it->second.pop_back();
                ++recycled;
                return View{b, 0, c};
            }
        }
        Buf b{};
        b.N = n; b.H = h; b.W = w; b.cs = cs_; b.internal = true; b.zero_fill = cs_ != c;
        b.floats = (size_t)n * h * w * cs_;
        b.base = ws_alloc(b.floats);
        bufs.push_back(b);
        return View{(int)bufs.size() - 1, 0, c};
    }
    // A caller-owned NHWC tensor as a view (read or written in place)
    View external(const idh_tensor &t, int n) {
        Buf b{};
        b.N = n; b.H = t.H; b.W = t.W; b.cs = t.cs; b.internal = false; b.zero_fill = false;
        b.base = mode_ == MODE_RUN ? t.ptr : reinterpret_cast<float *>((uintptr_t(2) << 32) + ((uintptr_t)bufs.size() << 24));
        bufs.push_back(b);
        return View{(int)bufs.size() - 1, 0, t.C};
    }
    // Plan.release: the caller records no further op on `v` (a whole internal buffer): a later buffer() of the same shape may alias it
    void release(const View &v) {
        const Buf &b = bufs[v.buf];
        if (!b.internal || v.c0 != 0 || v.C != b.cs || (long long)b.floats * 4 < kReuseMinBytes) return;
        auto &pool = free_[std::make_tuple(b.N, b.H, b.W, b.cs)];
        if (std::find(pool.begin(), pool.end(), v.buf) != pool.end()) { err = IDH_EINVAL; return; }
        pool.push_back(v.buf);
    }

    static Region region(const View &v, bool pad16 = false) { return Region{v.buf, v.c0, v.c0 + (pad16 ? ceil16(v.C) : v.C)}; }

    // ---- kernel-family predicates (nhwc.py: wino_eligible, wino4_eligible, lds_eligible, s2_first_eligible) --------------------------------
    bool wino_eligible(const std::vector<Src> &srcs, int cout, int n, int Ho, int Wo) const {
        const idh_conv_params &c0 = *srcs[0].cv;
        if (c0.ks != 3 || c0.stride != 1 || cout % 32) return false;
        if (srcs.size() > 1 && (srcs[1].cv->ks != 1 || srcs[1].cv->stride != 1)) return false;
        const int ty = cdiv(Ho, 8), tx = cdiv(Wo, 32);
        if ((double)Ho * Wo < kWinoMinFill * (ty * 8) * (tx * 32)) return false;
        return (long long)n * ty * tx * (cout / 32) >= kWinoMinTiles;
    }
    bool wino4_eligible(const std::vector<Src> &srcs, int cout, int n, int Ho, int Wo, int act, float slope, const View &out, const View *res) const {
        const idh_conv_params &c0 = *srcs[0].cv;
        const View &v0 = srcs[0].v;
        if (c0.ks != 3 || c0.stride != 1 || cout % 64) return false;
        if (srcs.size() > 1 && (srcs.size() > 2 || srcs[1].cv->ks != 1 || srcs[1].cv->stride != 1)) return false;
        if ((act != IDH_ACT_NONE && act != IDH_ACT_LRELU && act != IDH_ACT_ELU) || (act == IDH_ACT_LRELU && !(slope >= 0.f && slope <= 1.f)) || c0.cin <= 16) return false;
        if ((long long)H(v0) * W(v0) * cs(v0) * 4 >= (1ll << 30)) return false;
        if ((long long)Ho * Wo * cs(out) * 4 >= (1ll << 31)) return false;
        if (res && (long long)Ho * Wo * cs(*res) * 4 >= (1ll << 31)) return false;
        if (srcs.size() > 1) {
            const View &v1 = srcs[1].v;
            if ((long long)H(v1) * W(v1) * cs(v1) * 4 >= (1ll << 31) || (long long)((srcs[1].cv->cin + 15) / 16) * 4 * ceil16(cout) * 64 >= (1ll << 31)) return false;
        }
        if ((long long)((c0.cin + 15) / 16) * 4 * ceil16(cout) * 36 * 16 * 4 >= (1ll << 31)) return false;
        const int ty = cdiv(Ho, 8), tx = cdiv(Wo, 32);
        if ((double)Ho * Wo < kWino4MinFill * (ty * 8) * (tx * 32)) return false;
        return (long long)n * ty * tx * (cout / 64) >= kWino4MinTiles;
    }
    static bool lds_eligible(const std::vector<Src> &srcs, int cout, int Wo) {
        const idh_conv_params &c0 = *srcs[0].cv;
        if (c0.ks != 3 || c0.stride != 1 || cout % 16 || Wo < 16) return false;
        if (srcs.size() > 1) {
            const idh_conv_params &c1 = *srcs[1].cv;
            const bool strided3 = c1.ks == 3 && c1.stride == 2 && cout % 32 == 0;
            if (!strided3 && (c1.ks != 1 || c1.stride != 1)) return false;
        }
        return true;
    }
    static int lds_subtiles(int cout) { return cout % 64 == 0 ? 4 : (cout % 32 == 0 ? 2 : 1); }
    static bool s2_first_eligible(const std::vector<Src> &srcs, int cout, int n, int Ho, int Wo) {
        if (srcs.size() != 1) return false;
        const idh_conv_params &c0 = *srcs[0].cv;
        if (c0.ks != 3 || c0.stride != 2 || cout % 32 || Wo < 16) return false;
        return (long long)n * cdiv(Wo, 16) * cdiv(Ho, 4) * (cout / (16 * lds_subtiles(cout))) >= kS2FirstMinBlocks;
    }
    static void choose_lds_tile(int n, int Ho, int Wo, int cout, int chunks, int &code, int &split) {
        const long long per_row = (long long)n * cdiv(Wo, 16) * (cout / (16 * lds_subtiles(cout)));
        code = 8;
        int rows = 8;
        if (per_row * cdiv(Ho, 8) < 768) { code = 9; rows = 4; }
        const long long blocks = per_row * cdiv(Ho, rows);
        long long s = (768 + blocks - 1) / blocks;
        s = std::min<long long>(s, chunks / kSplitMinChunks);
        s = std::min<long long>(s, kSplitMax);
        split = (int)std::max<long long>(1, s);
    }
    static void choose_tiles(long long M, int cout, int steps, int &tm, int &tn, int &split) {
        const int nsub = ceil16(cout) / 16;
        tn = nsub % 4 == 0 ? 4 : (nsub % 2 == 0 ? 2 : 1);
        long long waves = 0;
        tm = 1;
        for (int cand : {4, 2, 1}) {
            waves = ((M + 16 * cand - 1) / (16 * cand)) * (nsub / tn);
            tm = cand;
            if (waves >= kTargetWaves) break;
        }
        split = 1;
        if (waves < kMinWaves) {
            long long s = (kMinWaves + waves - 1) / waves;
            s = std::min<long long>(s, steps / 4);
            s = std::min<long long>(s, 32);
            split = (int)std::max<long long>(1, s);
        }
    }

    // ---- weights -----------------------------------------------------------------------------------------------------------------------
    enum WLayout { W_DIRECT, W_WINO, W_WINO4 };
    const float *packed(const idh_conv_params &cv, WLayout lay) {
        size_t n = 0;
        if (lay == W_WINO4) n = idh_packed_wino4_weight_floats(cv.cout, cv.cin);
        else if (lay == W_WINO) n = idh_packed_wino_weight_floats(cv.cout, cv.cin);
        else n = idh_packed_weight_floats(cv.cout, cv.cin, cv.ks);
        float *dst = blob_alloc(n);
        if (mode_ == MODE_PACK) {
            if (!cv.weight) { err = IDH_EINVAL; return dst; }
            int rc;
            if (lay == W_WINO4) rc = idh_pack_conv_weight_wino4(cv.weight, dst, cv.cout, cv.cin, st_);
            else if (lay == W_WINO) rc = idh_pack_conv_weight_wino(cv.weight, dst, cv.cout, cv.cin, st_);
            else rc = idh_pack_conv_weight(cv.weight, dst, cv.cout, cv.cin, cv.ks, st_);
            if (rc != IDH_OK) err = rc;
        }
        return dst;
    }
    const float *bias_of(const idh_conv_params &a, const idh_conv_params *b) {
        // (the blob always holds cout floats per conv launch: a layer's kernel then never reads the caller's parameter memory)
        float *dst = blob_alloc(a.cout);
        if (mode_ == MODE_PACK) {
            hipLaunchKernelGGL(bias_sum_k, dim3(cdiv(a.cout, 256)), dim3(256), 0, st_, a.bias, b ? b->bias : nullptr, dst, a.cout);
            if (hipGetLastError() != hipSuccess) err = IDH_ELAUNCH;
        }
        return dst;
    }

    // ---- ops (Plan.conv / upsample2 / import_nchw / export_nchw / head) ---------------------------------------------------------------------
    View conv(const View &x, const idh_conv_params &cv, const View &out, int act, float slope, const View *res, const View *x2, const idh_conv_params *cv2) {
        if (err) return out;
        idh_op op;
        std::memset(&op, 0, sizeof op);
        op.kind = IDH_OP_CONV;
        op.N = N(x);
        std::vector<Src> srcs{{x, &cv}};
        if (x2) srcs.push_back({*x2, cv2});
        const int n = N(out), Ho = H(out), Wo = W(out), cout = cv.cout;
        if (out.C != cout) { err = IDH_EINVAL; return out; }
        bool use_wino = wino_eligible(srcs, cout, n, Ho, Wo);
        const bool use_wino4 = (!x2 || !res) && wino4_eligible(srcs, cout, n, Ho, Wo, act, slope, out, res);
        if (use_wino4) use_wino = false;
        int steps = 0;
        for (size_t i = 0; i < srcs.size(); ++i) {
            const View &v = srcs[i].v;
            const idh_conv_params &c = *srcs[i].cv;
            if (v.C != c.cin || (c.ks != 1 && c.ks != 3) || c.cout != cout) { err = IDH_EINVAL; return out; }
            if (v.C % 16 && (v.c0 != 0 || cs(v) != ceil16(v.C))) { err = IDH_EINVAL; return out; }  // odd channel counts: whole zero-padded buffers only
            const WLayout lay = (use_wino4 && i == 0) ? W_WINO4 : (use_wino && i == 0) ? W_WINO : W_DIRECT;
            idh_conv_src &s = op.src[i];
            s.in = ptr(v); s.w = packed(c, lay); s.cs = cs(v); s.H = H(v); s.W = W(v); s.Cin = v.C;
            s.ks = c.ks; s.stride = c.stride; s.pad_mode = IDH_PAD_ZEROS;
            steps += c.ks * c.ks * (ceil16(v.C) / 16);
        }
        op.bias = bias_of(cv, cv2);  // (always present in the blob - BasicBlock's convs all have one, layers.py:52-55; a NULL bias packs as zeros)
        if (res) { op.res = ptr(*res); op.res_cs = cs(*res); }
        op.out = ptr(out); op.out_cs = cs(out);
        op.Ho = Ho; op.Wo = Wo; op.Cout = cout;
        op.act = act; op.slope = slope;
        const long long M = (long long)n * Ho * Wo;
        int tm, tn, split;
        if (use_wino4) { tm = kTileWino4; tn = 0; split = 1; ++n_wino4; }
        else if (use_wino) { tm = kTileWino; tn = 0; split = 1; ++n_wino2; }
        else if (lds_eligible(srcs, cout, Wo)) {
            double ch = 0;
            for (const Src &s : srcs) ch += (ceil16(s.v.C) / 16) * (s.cv->ks == 3 ? 1.0 : kProjChunkWeight);
            choose_lds_tile(n, Ho, Wo, cout, (int)ch, tm, split);
            tn = lds_subtiles(cout);
            if (tn == 4 && tm == 9 && kNarrowTileBelow) {
                const long long blocks64 = (long long)n * cdiv(Ho, 4) * cdiv(Wo, 16) * (cout / 64) * split;
                if (blocks64 < kNarrowTileBelow) tn = 2;
            }
            tn = tn == 4 ? 0 : tn;
        } else if (s2_first_eligible(srcs, cout, n, Ho, Wo)) {
            choose_lds_tile(n, Ho, Wo, cout, ceil16(x.C) / 16, tm, split);
            tn = lds_subtiles(cout);
            tn = tn == 4 ? 0 : tn;
        } else {
            choose_tiles(M, cout, steps, tm, tn, split);
        }
        op.tile_m = tm; op.tile_n = tn; op.split_k = split;
        if (split > 1) op.ws = ws_alloc((size_t)split * M * ceil16(cout));
        ops.push_back(op);
        Meta m;
        if (res) m.reads.push_back(region(*res));
        for (const Src &s : srcs) m.reads.push_back(region(s.v, true));
        m.writes.push_back(region(out));
        meta.push_back(m);
        return out;
    }
    void upsample2(const View &x, const View &out) {
        if (err) return;
        idh_op op;
        std::memset(&op, 0, sizeof op);
        op.kind = IDH_OP_UPSAMPLE2; op.N = N(x);
        idh_conv_src &s = op.src[0];
        s.in = ptr(x); s.cs = cs(x); s.H = H(x); s.W = W(x); s.Cin = x.C;
        op.out = ptr(out); op.out_cs = cs(out);
        ops.push_back(op);
        meta.push_back(Meta{{region(x)}, {region(out)}});
    }
    void import_nchw(const float *src, int n, int C, int H_, int W_, const View &out) {
        if (err) return;
        if (n != N(out) || H_ != H(out) || W_ != W(out) || C != out.C) { err = IDH_EINVAL; return; }
        idh_op op;
        std::memset(&op, 0, sizeof op);
        op.kind = IDH_OP_NCHW_TO_NHWC; op.N = n;
        op.src[0].in = mode_ == MODE_RUN ? src : reinterpret_cast<const float *>(uintptr_t(3) << 32);
        op.src[0].H = H_; op.src[0].W = W_; op.src[0].Cin = C;
        op.out = ptr(out); op.out_cs = cs(out);
        ops.push_back(op);
        meta.push_back(Meta{{}, {region(out)}});
    }
    void export_nchw(const View &x, float *dst) {
        if (err) return;
        idh_op op;
        std::memset(&op, 0, sizeof op);
        op.kind = IDH_OP_NHWC_TO_NCHW; op.N = N(x);
        idh_conv_src &s = op.src[0];
        s.in = ptr(x); s.cs = cs(x); s.H = H(x); s.W = W(x); s.Cin = x.C;
        op.out = mode_ == MODE_RUN ? dst : reinterpret_cast<float *>(uintptr_t(4) << 32);
        ops.push_back(op);
        meta.push_back(Meta{{region(x)}, {}});
    }
    void head(const View &x, const idh_conv_params &cv, float *out, float *out_exp) {
        if (err) return;
        if (cv.cout != 1 || cv.ks != 1 || cv.cin != x.C) { err = IDH_EINVAL; return; }
        float *w = blob_alloc(cv.cin), *b = blob_alloc(1);
        if (mode_ == MODE_PACK) {
            if (!cv.weight || !cv.bias) { err = IDH_EINVAL; return; }
            if (hipMemcpyAsync(w, cv.weight, sizeof(float) * cv.cin, hipMemcpyDeviceToDevice, st_) != hipSuccess ||
                hipMemcpyAsync(b, cv.bias, sizeof(float), hipMemcpyDeviceToDevice, st_) != hipSuccess) err = IDH_ELAUNCH;
        }
        idh_op op;
        std::memset(&op, 0, sizeof op);
        op.kind = IDH_OP_POINTWISE_HEAD; op.N = N(x);
        idh_conv_src &s = op.src[0];
        s.in = ptr(x); s.w = w; s.cs = cs(x); s.H = H(x); s.W = W(x); s.Cin = x.C;
        op.bias = b;
        op.out = mode_ == MODE_RUN ? out : reinterpret_cast<float *>(uintptr_t(5) << 32);
        op.ws = mode_ == MODE_RUN ? out_exp : nullptr;
        ops.push_back(op);
        meta.push_back(Meta{{region(x)}, {}});
    }

    // ---- BasicBlock (Plan.basic_block; reference layers.py:78-95) ------------------------------------------------------------------------------
    View basic_block(const View &x, const idh_block_params &blk, const View *out_opt = nullptr) {
        const int st = blk.conv1.stride;
        if (st != 1 && st != 2) { err = IDH_EINVAL; return x; }
        const int Ho = (H(x) + 2 - 3) / st + 1, Wo = (W(x) + 2 - 3) / st + 1;
        const int planes = blk.conv1.cout;
        const View h = buffer(N(x), Ho, Wo, planes);
        conv(x, blk.conv1, h, IDH_ACT_LRELU, 0.2f, nullptr, nullptr, nullptr);
        const View out = out_opt ? *out_opt : buffer(N(x), Ho, Wo, planes);
        if (H(out) != Ho || W(out) != Wo) { err = IDH_EINVAL; return out; }
        if (blk.downsample.ks == 0) {
            if (x.C != planes || st != 1) { err = IDH_EINVAL; return out; }
            conv(h, blk.conv2, out, IDH_ACT_LRELU, 0.2f, &x, nullptr, nullptr);
        } else {
            conv(h, blk.conv2, out, IDH_ACT_LRELU, 0.2f, nullptr, &x, &blk.downsample);
        }
        release(h);  // the block's intermediate dies with conv2
        return out;
    }

    // ---- Plan.schedule: dependency levels, launch order inside a level, group ids ----------------------------------------------------------
    static bool overlap(const std::vector<Region> &a, const std::vector<Region> &b) {
        for (const Region &x : a)
            for (const Region &y : b)
                if (x.buf == y.buf && x.c0 < y.c1 && y.c0 < x.c1) return true;
        return false;
    }
    static void launch_rank(const idh_op &op, long long r[3]) {
        r[0] = 3; r[1] = 0; r[2] = 0;
        if (op.kind == IDH_OP_CONV && op.tile_m == kTileWino) { r[0] = -1; r[1] = op.src[1].in ? 1 : 0; r[2] = -(long long)op.N * op.Ho * op.Wo * op.Cout; }
        else if (op.kind == IDH_OP_CONV && op.tile_m == 9) { r[0] = 0; r[1] = op.tile_n; }
        else if (op.kind == IDH_OP_CONV && op.tile_m == 1 && op.tile_n == 4) { r[0] = 1; }
        else if (op.kind == IDH_OP_UPSAMPLE2) { r[0] = 2; }
        else if (op.kind == IDH_OP_NCHW_TO_NHWC) { r[0] = 2; r[1] = 1; }
    }
    void schedule() {
        const int n = (int)ops.size();
        std::vector<int> level(n, 0), order(n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i)
                if (overlap(meta[i].writes, meta[j].reads) || overlap(meta[i].writes, meta[j].writes) || overlap(meta[i].reads, meta[j].writes))
                    level[j] = std::max(level[j], level[i] + 1);
        std::vector<std::array<long long, 5>> key(n);
        for (int k = 0; k < n; ++k) {
            long long r[3];
            launch_rank(ops[k], r);
            key[k] = {level[k], r[0], r[1], r[2], k};
            ops[k].group = r[0] < 3 ? level[k] + 1 : 0;
            order[k] = k;
        }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
        std::vector<idh_op> o2(n);
        for (int k = 0; k < n; ++k) o2[k] = ops[order[k]];
        ops.swap(o2);
    }

    int finish(idh_net_sizes *sizes) {
        if (err) return err;
        schedule();
        if (sizes) {
            sizes->workspace_floats = ws_off;
            sizes->weight_floats = blob_off;
            sizes->ops = (int)ops.size();
            sizes->launches = idh_count_launches(ops.data(), (int)ops.size());
            sizes->wino4 = n_wino4; sizes->wino2 = n_wino2; sizes->recycled = recycled;
            if (sizes->launches < 0) return sizes->launches;
        }
        if (mode_ == MODE_RUN) {
            if (ws_off > ws_cap_) return IDH_EWORKSPACE;
            for (const Buf &b : bufs)
                if (b.internal && b.zero_fill && hipMemsetAsync(b.base, 0, b.floats * sizeof(float), st_) != hipSuccess) return IDH_ELAUNCH;
            return idh_run_ops(ops.data(), (int)ops.size(), st_);
        }
        return IDH_OK;
    }

    Mode mode() const { return mode_; }

  private:
    Mode mode_;
    float *ws_;
    size_t ws_cap_;
    float *blob_;
    hipStream_t st_;
    std::map<std::tuple<int, int, int, int>, std::vector<int>> free_;
};

// `read`: the tensor is read by a conv (whole 16-channel blocks: an odd channel count must be a zero-padded buffer of its own); a tensor that is
// only written may be any 16-byte-aligned channel slice
bool tensor_ok(const idh_tensor *t, bool need_ptr, bool read = true) {
    if (!t || t->C <= 0 || t->H <= 0 || t->W <= 0) return false;
    if (t->layout != IDH_LAYOUT_NHWC && t->layout != IDH_LAYOUT_NCHW) return false;
    if (t->layout == IDH_LAYOUT_NHWC && (t->cs < t->C || (t->cs & 3) || (read && (t->C & 15) && t->cs != ceil16(t->C)))) return false;
    if (need_ptr && (!t->ptr || ((uintptr_t)t->ptr & 15))) return false;
    return true;
}

// an input tensor as a view of the plan: NHWC in place, NCHW through a layout import into a plan buffer
View input_view(Plan &p, const idh_tensor &t, int N) {
    if (t.layout == IDH_LAYOUT_NHWC) return p.external(t, N);
    const View v = p.buffer(N, t.H, t.W, t.C);
    p.import_nchw(t.ptr, N, t.C, t.H, t.W, v);
    return v;
}
// where a block that produces output tensor `t` should write: the caller's NHWC memory, or a plan buffer that is exported afterwards
View output_view(Plan &p, const idh_tensor &t, int N) {
    if (t.layout == IDH_LAYOUT_NHWC) return p.external(t, N);
    return p.buffer(N, t.H, t.W, t.C);
}
void output_done(Plan &p, const idh_tensor &t, const View &v) {
    if (t.layout == IDH_LAYOUT_NCHW) p.export_nchw(v, t.ptr);
}

// ---- the three networks ---------------------------------------------------------------------------------------------------------------------
int build_basic_block(Plan &p, const idh_block_params *blk, int N, const idh_tensor *x, const idh_tensor *out) {
    const bool run = p.mode() == MODE_RUN;
    if (!blk || N <= 0 || !tensor_ok(x, run) || !tensor_ok(out, run, false) || x->C != blk->conv1.cin || out->C != blk->conv1.cout) return IDH_EINVAL;
    const View xin = input_view(p, *x, N);
    const View o = output_view(p, *out, N);
    p.basic_block(xin, *blk, &o);
    output_done(p, *out, o);
    return p.err;
}

// CVEncoder.forward (networks.py:208-215): x = ds_conv_i(x); x = cat([x, img_feats[i]]); x = conv_i(x)
int build_cvencoder(Plan &p, const idh_block_params *blocks, int num_blocks, int N, const idh_tensor *cost, const idh_tensor *img, const idh_tensor *outs) {
    const bool run = p.mode() == MODE_RUN;
    if (!blocks || num_blocks <= 0 || num_blocks > 8 || N <= 0 || !tensor_ok(cost, run) || !img || !outs) return IDH_EINVAL;
    View x = input_view(p, *cost, N);
    for (int i = 0; i < num_blocks; ++i) {
        const idh_block_params &ds = blocks[3 * i], &c0 = blocks[3 * i + 1], &c1 = blocks[3 * i + 2];
        if (!tensor_ok(&img[i], run) || !tensor_ok(&outs[i], run)) return IDH_EINVAL;
        const int st = ds.conv1.stride;
        if (st != 1 && st != 2) return IDH_EINVAL;
        const int Ho = (p.H(x) + 2 - 3) / st + 1, Wo = (p.W(x) + 2 - 3) / st + 1;
        const int cout = ds.conv1.cout, cimg = img[i].C;
        if (img[i].H != Ho || img[i].W != Wo || c0.conv1.cin != cout + cimg || outs[i].C != c1.conv1.cout || outs[i].H != Ho || outs[i].W != Wo) return IDH_EINVAL;
        const View cat = p.buffer(N, Ho, Wo, cout + cimg);
        const View left = Plan::slice(cat, 0, cout);
        p.basic_block(x, ds, &left);
        const View right = Plan::slice(cat, cout, cimg);
        if (img[i].layout == IDH_LAYOUT_NCHW) p.import_nchw(img[i].ptr, N, cimg, Ho, Wo, right);
        else return IDH_EUNSUPPORTED;  // (an NHWC image-feature map would need a copy op into the concat slice: the reference hands NCHW)
        View y = p.basic_block(cat, c0);
        const View o = output_view(p, outs[i], N);
        y = p.basic_block(y, c1, &o);
        output_done(p, outs[i], y);
        x = y;
        if (p.err) return p.err;
    }
    return p.err;
}

// BDDecoderPP / DepthDecoderPP.forward (networks.py:64-84, 163-183)
int build_unetpp(Plan &p, const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, int N, const idh_tensor *feats,
                 const idh_tensor *fouts, float *const *log_depth, float *const *depth) {
    const bool run = p.mode() == MODE_RUN;
    if (!blocks || n_blocks != IDH_UNETPP_BLOCKS || N <= 0 || !feats) return IDH_EINVAL;
    std::vector<View> prev;
    for (int i = 0; i < 5; ++i) {
        if (!tensor_ok(&feats[i], run)) return IDH_EINVAL;
        if (i && (feats[i].H * 2 != feats[i - 1].H || feats[i].W * 2 != feats[i - 1].W)) return IDH_EINVAL;  // pyramid levels differ by exactly x2
        prev.push_back(input_view(p, feats[i], N));
    }
    const idh_block_params *out_blk[4] = {nullptr, &blocks[46], &blocks[47], &blocks[48]};
    auto want = [&](int i) { return fouts && fouts[i].C > 0; };  // (C == 0 skips a level; decided by the shape alone so that sizes / pack / fwd agree)
    std::vector<View> outputs;
    View final_v[4];
    int bi = 0;
    for (int j = 1; j <= 4; ++j) {
        for (int i = 4 - j; i >= 0; --i) {
            const idh_block_params &right = blocks[bi++], &diag = blocks[bi++];
            const bool has_up = (i + j) != 4;
            const idh_block_params *up = has_up ? &blocks[bi++] : nullptr;
            const idh_block_params &in0 = blocks[bi++], &in1 = blocks[bi++];
            const int cout = right.conv1.cout;
            const View xi = prev[i];
            const View cat = p.buffer(N, p.H(xi), p.W(xi), cout * (has_up ? 3 : 2));
            const View s0 = Plan::slice(cat, 0, cout);
            p.basic_block(xi, right, &s0);
            const View lo = p.basic_block(prev[i + 1], diag);
            if (p.H(lo) * 2 != p.H(xi) || p.W(lo) * 2 != p.W(xi)) return IDH_EINVAL;
            p.upsample2(lo, Plan::slice(cat, cout, cout));
            p.release(lo);  // (liveness reuse: the half-resolution map has no reader after its upsampling)
            if (has_up) {
                const View lo2 = p.basic_block(outputs.back(), *up);
                p.upsample2(lo2, Plan::slice(cat, 2 * cout, cout));
                p.release(lo2);
            }
            const View y0 = p.basic_block(cat, in0);
            p.release(cat);
            // the decoder's top-left result IS feature_s0 (output_0[0] = nn.Identity, networks.py:61): written straight into the caller's tensor
            const bool last = j == 4 - i;
            View y;
            if (last && i == 0 && want(0) && fouts[0].layout == IDH_LAYOUT_NHWC) {
                if (!tensor_ok(&fouts[0], run) || fouts[0].C != in1.conv1.cout) return IDH_EINVAL;
                const View o = p.external(fouts[0], N);
                y = p.basic_block(y0, in1, &o);
            } else {
                y = p.basic_block(y0, in1);
                if (last && i == 0 && want(0)) {
                    if (!tensor_ok(&fouts[0], run) || fouts[0].C != in1.conv1.cout) return IDH_EINVAL;
                    p.export_nchw(y, fouts[0].ptr);
                }
            }
            p.release(y0);
            outputs.push_back(y);
            if (last) {  // the only (i, j) whose output_i result survives in the reference's dict
                if (i == 0) final_v[0] = y;
                else if (want(i)) {
                    if (!tensor_ok(&fouts[i], run) || fouts[i].C != out_blk[i]->conv1.cout) return IDH_EINVAL;
                    const View o = output_view(p, fouts[i], N);
                    final_v[i] = p.basic_block(y, *out_blk[i], &o);
                    output_done(p, fouts[i], final_v[i]);
                } else if (heads) {
                    final_v[i] = p.basic_block(y, *out_blk[i]);
                }
            }
            if (p.err) return p.err;
        }
        prev.assign(outputs.rbegin(), outputs.rend());
    }
    if (heads) {
        for (int i = 0; i < 4; ++i) {
            if (run && (!log_depth || !log_depth[i])) return IDH_EINVAL;
            p.head(final_v[i], heads[i], run ? log_depth[i] : nullptr, (run && depth) ? depth[i] : nullptr);
        }
    }
    return p.err;
}

}  // namespace

extern "C" int idh_basic_block_sizes(const idh_block_params *blk, int N, const idh_tensor *x, const idh_tensor *out, idh_net_sizes *sizes) {
    if (!sizes) return IDH_EINVAL;
    Plan p(MODE_SIZES, nullptr, 0, nullptr, nullptr);
    const int rc = build_basic_block(p, blk, N, x, out);
    return rc != IDH_OK ? rc : p.finish(sizes);
}
extern "C" int idh_basic_block_pack(const idh_block_params *blk, int N, const idh_tensor *x, const idh_tensor *out, float *blob, void *stream) {
    if (!blob || ((uintptr_t)blob & 255)) return IDH_EINVAL;
    Plan p(MODE_PACK, nullptr, 0, blob, idh_stream(stream));
    const int rc = build_basic_block(p, blk, N, x, out);
    return rc != IDH_OK ? rc : p.err;
}
extern "C" int idh_basic_block_fwd(const idh_block_params *blk, const float *blob, int N, const idh_tensor *x, const idh_tensor *out, float *ws,
                                   size_t ws_floats, void *stream) {
    if (!blob || ((uintptr_t)blob & 255) || (ws_floats && (!ws || ((uintptr_t)ws & 255)))) return IDH_EINVAL;
    Plan p(MODE_RUN, ws, ws_floats, const_cast<float *>(blob), idh_stream(stream));
    const int rc = build_basic_block(p, blk, N, x, out);
    return rc != IDH_OK ? rc : p.finish(nullptr);
}

extern "C" int idh_cvencoder_sizes(const idh_block_params *blocks, int num_blocks, int N, const idh_tensor *cost, const idh_tensor *img_feats,
                                   const idh_tensor *outs, idh_net_sizes *sizes) {
    if (!sizes) return IDH_EINVAL;
    Plan p(MODE_SIZES, nullptr, 0, nullptr, nullptr);
    const int rc = build_cvencoder(p, blocks, num_blocks, N, cost, img_feats, outs);
    return rc != IDH_OK ? rc : p.finish(sizes);
}
extern "C" int idh_cvencoder_pack(const idh_block_params *blocks, int num_blocks, int N, const idh_tensor *cost, const idh_tensor *img_feats,
                                  const idh_tensor *outs, float *blob, void *stream) {
    if (!blob || ((uintptr_t)blob & 255)) return IDH_EINVAL;
    Plan p(MODE_PACK, nullptr, 0, blob, idh_stream(stream));
    const int rc = build_cvencoder(p, blocks, num_blocks, N, cost, img_feats, outs);
    return rc != IDH_OK ? rc : p.err;
}
extern "C" int idh_cvencoder_fwd(const idh_block_params *blocks, int num_blocks, const float *blob, int N, const idh_tensor *cost,
                                 const idh_tensor *img_feats, const idh_tensor *outs, float *ws, size_t ws_floats, void *stream) {
    if (!blob || ((uintptr_t)blob & 255) || !ws || ((uintptr_t)ws & 255)) return IDH_EINVAL;
    Plan p(MODE_RUN, ws, ws_floats, const_cast<float *>(blob), idh_stream(stream));
    const int rc = build_cvencoder(p, blocks, num_blocks, N, cost, img_feats, outs);
    return rc != IDH_OK ? rc : p.finish(nullptr);
}

extern "C" int idh_unetpp_sizes(const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, int N, const idh_tensor *feats,
                                const idh_tensor *feature_outs, idh_net_sizes *sizes) {
    if (!sizes) return IDH_EINVAL;
    Plan p(MODE_SIZES, nullptr, 0, nullptr, nullptr);
    const int rc = build_unetpp(p, blocks, n_blocks, heads, N, feats, feature_outs, nullptr, nullptr);
    return rc != IDH_OK ? rc : p.finish(sizes);
}
extern "C" int idh_unetpp_pack(const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, int N, const idh_tensor *feats,
                               const idh_tensor *feature_outs, float *blob, void *stream) {
    if (!blob || ((uintptr_t)blob & 255)) return IDH_EINVAL;
    Plan p(MODE_PACK, nullptr, 0, blob, idh_stream(stream));
    const int rc = build_unetpp(p, blocks, n_blocks, heads, N, feats, feature_outs, nullptr, nullptr);
    return rc != IDH_OK ? rc : p.err;
}
extern "C" int idh_unetpp_fwd(const idh_block_params *blocks, int n_blocks, const idh_conv_params *heads, const float *blob, int N,
                              const idh_tensor *feats, const idh_tensor *feature_outs, float *const *log_depth_outs, float *const *depth_outs,
                              float *ws, size_t ws_floats, void *stream) {
    if (!blob || ((uintptr_t)blob & 255) || !ws || ((uintptr_t)ws & 255)) return IDH_EINVAL;
    Plan p(MODE_RUN, ws, ws_floats, const_cast<float *>(blob), idh_stream(stream));
    const int rc = build_unetpp(p, blocks, n_blocks, heads, N, feats, feature_outs, log_depth_outs, depth_outs);
    return rc != IDH_OK ? rc : p.finish(nullptr);
}
