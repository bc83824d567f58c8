// Host runtime of libmewzoom_hip.so: configuration checks, weight registry + packing, workspace
// planning, the layer schedule of the MewZoom forward pass, and the C ABI (include/mewzoom_hip.h).
//
// The schedule follows the reference's MewZoom.forward (src/ultrazoom/model.py:149-164) and its
// sub-modules; each step cites the reference lines it replaces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mewzoom_hip.h"
#include "mz_kernels.h"

using namespace mz;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
// kernel family of this thread's most recent convolution / mix launch (mz_debug_last_kernel(): the tests assert WHICH kernel they compare)
static thread_local const char* g_last_kernel = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MZ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Per-device state: the raised dynamic-LDS limits (hipFuncSetAttribute applies to the CURRENT device) and the CU count.
static constexpr int kMaxDevices = 64;
static int g_dev_ready[kMaxDevices];  // 0 unknown, 1 ok
static int g_dev_cus[kMaxDevices];

static int ensure_device_ready() {
    int n = 0, dev = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MZ_ERR_NO_DEVICE, "no HIP device visible");
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fail(MZ_ERR_NO_DEVICE, "bad current device");
    if (g_dev_ready[dev] == 1) return MZ_OK;
    hipError_t e = init_kernels();
    if (e != hipSuccess) return fail(MZ_ERR_HIP, "kernel init failed: %s", hipGetErrorString(e));
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    g_dev_cus[dev] = cus / 8 * 8;
    g_dev_ready[dev] = 1;
    return MZ_OK;
}

// Environment knobs (A/B timing and test coverage of every kernel variant).  Read ONCE, when a
// handle is created (or per mz_op_* call), never on the launch path.
struct Knobs {
    int use_glds = 1;       // MZ_USE_GLDS=0: stage through registers instead of LDS-DMA (needs a -DMZ_REG_STAGING build)
    bool wide = true;       // MZ_NO_WIDE=1: force the 256-pixel kernel
    bool fuse = true;       // MZ_NO_FUSE=1: conv2 and the mix as two launches
    bool s16 = true;        // MZ_NO_S16=1: keep 16-bit types on the 32x32x16 kernels
    bool fuse16 = true;     // MZ_NO_FUSE16=1: the fused mix stays on the 32x32x16 kernel
    bool mix16 = true;      // MZ_NO_MIX16=1: C = k * 192 mixes on the general 1x1 kernel
    bool mix16b = true;     // MZ_NO_MIX16B=1: ... on mix16_kernel (blend in accumulator layout, x and z read twice) instead of mix16b_kernel
    int persist = -1;       // MZ_NO_PERSIST=1 -> 0 (one workgroup per tile); MZ_PERSIST_WGS=n -> n; -1 = one per CU
    int kpad_pct = 12;      // MZ_KPAD_PCT=n: the 16x16x32 kernels take Cin whose padding to whole 32-channel chunks is <= n %
    int blk4 = 1;           // MZ_NO_BLK4=1: row-major tile walk inside an image (A/B of the L2 sharing of vertical halos)
    int r = 1;              // MZ_NO_R=1: never use conv3r_kernel (96-channel N tiles, 8 x 48 / 8 x 40 pixel tiles, role-alternating waves: epilogues under the next K loop)
    int geo40 = 1;          // MZ_NO_GEO40=1: conv3r_kernel keeps its 8 x 48 tiles where 8 x 40 tiles would pad fewer pixels
    int r2 = 1;             // MZ_NO_R2=1: Cin = 48 -> 96-channel N tiles (conv1 of the 48-channel models' level-1 block) stays off conv3r_kernel's ragged variant
    int head256 = 1;        // MZ_NO_HEAD256=1: the image head (EPI_FINAL) stays on the 512-pixel per-tile kernel instead of the 256-pixel one
    int t = 1;              // MZ_NO_T=1: never use conv3t_kernel (the same structure for ONE N tile of <= 48 channels, 12 x 64 tiles)
};
static Knobs read_knobs() {
    Knobs k;
    if (const char* e = getenv("MZ_USE_GLDS")) k.use_glds = atoi(e) != 0;
    k.wide = getenv("MZ_NO_WIDE") == nullptr;
    k.fuse = getenv("MZ_NO_FUSE") == nullptr;
    k.s16 = getenv("MZ_NO_S16") == nullptr;
    k.fuse16 = getenv("MZ_NO_FUSE16") == nullptr;
    k.mix16 = getenv("MZ_NO_MIX16") == nullptr;
    k.mix16b = getenv("MZ_NO_MIX16B") == nullptr;
    k.r = getenv("MZ_NO_R") == nullptr;
    k.t = getenv("MZ_NO_T") == nullptr;
    k.head256 = getenv("MZ_NO_HEAD256") == nullptr;
    k.r2 = getenv("MZ_NO_R2") == nullptr;
    k.geo40 = getenv("MZ_NO_GEO40") == nullptr;
    k.blk4 = getenv("MZ_NO_BLK4") == nullptr;
    if (const char* e = getenv("MZ_KPAD_PCT")) k.kpad_pct = atoi(e);
    if (getenv("MZ_NO_PERSIST") != nullptr) k.persist = 0;
    else if (const char* e = getenv("MZ_PERSIST_WGS")) { const int n = atoi(e) / 8 * 8; k.persist = n > 0 ? n : 0; }
    return k;
}

// ------------------------------------------------------------------------------------------------
// model description
// ------------------------------------------------------------------------------------------------
struct ConvW {
    // logical (reference) shape
    int cout = 0, cin = 0, kh = 0, kw = 0;
    // kernel selection
    int mode = MODE_CONV3, taps = 9, nt = 1, ntiles = 1, nchunks = 1, nchunks_real = 1;
    int out_map = OUT_PLAIN, cq = 0, cq_p = 0;
    int in_map = SRC_PLAIN, c0 = 0, cp0 = 0, c1 = 0;
    int n_logical_padded = 0;
    void* packed = nullptr;
    size_t packed_sz = 0;
    // second packing for the 16x16x32-MFMA kernel (16-bit types, wide 3x3 convs that are not the image head)
    void* packed16 = nullptr;
    size_t packed16_sz = 0;
    int nchunks16 = 0;
    void* packed16r = nullptr;  // fused gate weights once more, both halves in accumulator-row order (conv3r_kernel; PackArgs::frag16 = 2)
    void* packed16t = nullptr;  // conv3t_kernel (one N tile of <= 48 channels): three 16-channel fragments per tap; SRC_MIXF: its gate (frag16 = 4)
    size_t packed16t_sz = 0;
    int nchunks16t = 0;
    bool set = false;
};

static void plan_conv(ConvW& c, int dtype, int mode, int cout, int cin, int kh, int kw, int out_map, int in_map,
                      int c0, int c1) {
    const int ck = chunk_channels(dtype);
    c.cout = cout; c.cin = cin; c.kh = kh; c.kw = kw;
    c.mode = mode;
    c.taps = mode == MODE_CONV3 ? 9 : 1;
    c.out_map = out_map;
    c.in_map = in_map;
    if (out_map == OUT_D2S) {
        c.cq = cout / 4;
        c.cq_p = pad16(c.cq);
        c.n_logical_padded = 4 * c.cq_p;
    } else if (out_map == OUT_FINAL) {
        c.n_logical_padded = 16;
    } else {
        c.n_logical_padded = pad16(cout);
    }
    c.nt = choose_nt(c.n_logical_padded);
    c.ntiles = (c.n_logical_padded + 32 * c.nt - 1) / (32 * c.nt);
    if (in_map == SRC_PLAIN) {
        c.c0 = cin; c.cp0 = pad16(cin); c.c1 = 0;
        c.nchunks = c.cp0 / ck;
    } else if (in_map == SRC_CONCAT) {
        c.c0 = c0; c.cp0 = pad16(c0); c.c1 = c1;
        c.nchunks = (c.cp0 + pad16(c1)) / ck;
    } else {  // CRUSH
        c.c0 = cin; c.cp0 = pad16(cin); c.c1 = 0;
        c.nchunks = 4 * c.cp0 / ck;
    }
    c.nchunks_real = c.nchunks;
    if (mode == MODE_GEMM1) {  // the 1x1 kernel consumes S chunks per stage: pad K with zero weights
        const int S = gemm1_chunks_per_stage();
        c.nchunks = (c.nchunks + S - 1) / S * S;
    }
    c.packed_sz = packed_bytes(c.taps, c.nt, c.ntiles, c.nchunks);
    if (mode == MODE_CONV3 && dtype != DT_F32 && in_map == SRC_PLAIN && out_map != OUT_FINAL && c.nt <= 3) {
        c.nchunks16 = (c.cp0 + 31) / 32;
        c.packed16_sz = packed_bytes(c.taps, 2 * c.nt, c.ntiles, c.nchunks16);
    }
    // one N tile of 33..48 channels over whole 32-channel chunks: third packing, for conv3t_kernel (three fragments per tap)
    if (mode == MODE_CONV3 && dtype != DT_F32 && in_map == SRC_PLAIN && out_map == OUT_PLAIN && c.n_logical_padded == 48 && c.cp0 % 32 == 0) {
        c.nchunks16t = c.cp0 / 32;
        c.packed16t_sz = packed_bytes(c.taps, 3, 1, c.nchunks16t);
    }
    // AdaptiveResidualMix with C = k * 192: second packing for mix16_kernel (192-channel N tiles = 12 fragments of 16)
    if (mode == MODE_GEMM1 && dtype != DT_F32 && in_map == SRC_CONCAT && cout % 192 == 0 && c0 == cout && c1 == cout) {
        c.nchunks16 = 2 * cout / 32;
        c.packed16_sz = packed_bytes(1, 12, cout / 192, c.nchunks16);
    }
}

struct BlockW {
    ConvW conv1, conv2, mix;
    ConvW mixf;          // the gate weights once more, packed for the fused conv2 + mix epilogue (SRC_MIXF)
    bool fused = false;  // conv2 keeps all its output channels in one workgroup (<= 96): the mix runs in its epilogue
    float alpha = 0.f;
    bool alpha_set = false;
};

enum SlotKind { SK_CONV, SK_ALPHA, SK_STEM_W, SK_STEM_B, SK_QA_B };
struct Slot {
    std::string name;
    int kind;
    ConvW* conv = nullptr;
    BlockW* block = nullptr;  // for alpha (or skip mixes: alpha stored in skip_alpha)
    float* alpha = nullptr;
    bool* flag = nullptr;
    int64_t shape[4] = {0, 0, 0, 0};
    int ndim = 0;
};

struct ProfRec {
    hipEvent_t a, b;
    double flops, bytes;
    int is_conv3;
    int kind, B, H, W, cin, cout, nt, ntiles, mtiles, n_fast;  // kind: 0 conv3, 1 mix, 2 crush
};

struct mz_handle {
    mz_config cfg;
    int dtype;
    int ch[4], enc[4], dec[4];
    int nhead;
    // weights
    std::vector<std::unique_ptr<BlockW>> enc_blocks[4], dec_blocks[4], head_blocks;
    ConvW crush[3], up[3], skipmix[3];
    float skip_alpha[3] = {0, 0, 0};
    bool skip_alpha_set[3] = {false, false, false};
    std::vector<std::unique_ptr<ConvW>> head_up;
    ConvW qa_conv;
    float* stem_w4 = nullptr;  // [cp0][4]
    float* qa_bias = nullptr;  // [F]
    bool stem_w_set = false, stem_b_set = false, qa_b_set = false;
    void* zero_page = nullptr;
    std::vector<Slot> slots;
    std::unordered_map<std::string, int> slot_index;
    bool device_ready = false;
    Knobs knobs;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    size_t recs_used = 0;
    // tile lists of the role-alternating kernels (Runner::tile_table): one per launch geometry, built on first use
    std::map<std::vector<int>, std::pair<void*, int>> tile_tabs;
    ~mz_handle() {
        for (auto& t : tile_tabs)
            if (t.second.first) (void)hipFree(t.second.first);
    }
};

static void add_slot(mz_handle* h, const std::string& name, int kind, std::initializer_list<int64_t> shape) {
    Slot s;
    s.name = name;
    s.kind = kind;
    s.ndim = (int)shape.size();
    int i = 0;
    for (auto d : shape) s.shape[i++] = d;
    h->slot_index[name] = (int)h->slots.size();
    h->slots.push_back(s);
}

static void add_block(mz_handle* h, BlockW* b, const std::string& prefix, int c) {
    const int hr = h->cfg.hidden_ratio;
    plan_conv(b->conv1, h->dtype, MODE_CONV3, hr * c, c, 3, 3, OUT_PLAIN, SRC_PLAIN, 0, 0);      // model.py:742-744
    plan_conv(b->conv2, h->dtype, MODE_CONV3, c, hr * c, 3, 3, OUT_PLAIN, SRC_PLAIN, 0, 0);      // model.py:746-748
    plan_conv(b->mix, h->dtype, MODE_GEMM1, c, 2 * c, 1, 1, OUT_PLAIN, SRC_CONCAT, c, c);        // model.py:805
    b->fused = b->conv2.ntiles == 1 && b->conv2.nt <= 3;
    if (b->fused) {
        ConvW& f = b->mixf;
        f.cout = c; f.cin = 2 * c; f.kh = f.kw = 1;
        f.mode = MODE_GEMM1; f.taps = 1;
        f.nt = b->conv2.nt; f.ntiles = 1;
        f.out_map = OUT_PLAIN; f.in_map = SRC_MIXF;
        f.c0 = c; f.cp0 = pad16(c); f.c1 = c;
        const int zg = h->dtype == DT_F32 ? 4 : 2;
        f.nchunks = f.nchunks_real = f.cp0 / chunk_channels(h->dtype) + f.nt * zg;
        f.packed_sz = packed_bytes(1, f.nt, 1, f.nchunks);
        if (h->dtype != DT_F32) {  // second packing for the fused epilogue of the 16x16x32 kernel: 2 nt K-steps x 2 nt fragments
            f.nchunks16 = (f.cp0 + 31) / 32 + f.nt;
            f.packed16_sz = packed_bytes(1, 2 * f.nt, 1, f.nchunks16);
            if (f.cp0 == 48) f.packed16t_sz = packed_bytes(1, 3, 1, 3);  // conv3t_kernel's gate: three K steps x three fragments
        }
    }
    add_slot(h, prefix + ".convnet.conv1.weight", SK_CONV, {hr * c, c, 3, 3});
    h->slots.back().conv = &b->conv1;
    add_slot(h, prefix + ".convnet.conv2.weight", SK_CONV, {c, hr * c, 3, 3});
    h->slots.back().conv = &b->conv2;
    add_slot(h, prefix + ".skip.alpha", SK_ALPHA, {});  // a module's own parameters precede its children's
    h->slots.back().alpha = &b->alpha;
    h->slots.back().flag = &b->alpha_set;
    add_slot(h, prefix + ".skip.conv.weight", SK_CONV, {c, 2 * c, 1, 1});
    h->slots.back().conv = &b->mix;
    h->slots.back().block = b;
}

static int validate(const mz_config& c) {
    // Same rejected values as the reference constructor (AssertionError there).
    if (!(c.upscale_ratio == 2 || c.upscale_ratio == 4 || c.upscale_ratio == 8))  // model.py:67-69
        return fail(MZ_ERR_INVALID_ARGUMENT, "Upscale ratio must be one of {2, 4, 8}, but got %d.", c.upscale_ratio);
    if (c.primary_channels <= 3)  // model.py:218-222
        return fail(MZ_ERR_INVALID_ARGUMENT, "Output channels must be greater than input channels.");
    const int ch[4] = {c.primary_channels, c.secondary_channels, c.tertiary_channels, c.quaternary_channels};
    const int ly[4] = {c.primary_layers, c.secondary_layers, c.tertiary_layers, c.quaternary_layers};
    const char* nm[4] = {"primary", "secondary", "tertiary", "quaternary"};
    for (int i = 0; i < 4; ++i) {
        if (ly[i] <= 1)  // model.py:265-275
            return fail(MZ_ERR_INVALID_ARGUMENT, "Number of %s layers must be greater than 1.", nm[i]);
        if (ch[i] <= 0) return fail(MZ_ERR_INVALID_ARGUMENT, "Number of channels must be greater than 0.");  // :737
    }
    if (!(c.hidden_ratio == 1 || c.hidden_ratio == 2 || c.hidden_ratio == 4))  // model.py:738
        return fail(MZ_ERR_INVALID_ARGUMENT, "Hidden ratio must be either 1, 2, or 4.");
    if (c.num_deg_features <= 0)  // model.py:356-358 (intent)
        return fail(MZ_ERR_INVALID_ARGUMENT, "Number of quality assessor features must be greater than 0.");
    return MZ_OK;
}

extern "C" int mz_create(const mz_config* cfg, int dtype, mz_handle** out) {
    if (!cfg || !out) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    if (dtype != MZ_F32 && dtype != MZ_BF16 && dtype != MZ_F16) return fail(MZ_ERR_INVALID_ARGUMENT, "bad dtype %d", dtype);
    int rc = validate(*cfg);
    if (rc) return rc;
    auto* h = new mz_handle();
    h->cfg = *cfg;
    h->dtype = dtype;
    const int ch[4] = {cfg->primary_channels, cfg->secondary_channels, cfg->tertiary_channels, cfg->quaternary_channels};
    const int ly[4] = {cfg->primary_layers, cfg->secondary_layers, cfg->tertiary_layers, cfg->quaternary_layers};
    for (int i = 0; i < 4; ++i) {
        h->ch[i] = ch[i];
        h->enc[i] = (ly[i] + 1) / 2;  // ceil, model.py:277-288
        h->dec[i] = ly[i] / 2;        // floor, model.py:290-300
    }
    h->nhead = cfg->upscale_ratio == 2 ? 1 : (cfg->upscale_ratio == 4 ? 2 : 3);  // model.py:945
    h->knobs = read_knobs();

    // Registry in the reference's state_dict order.
    h->slots.reserve(1024);
    add_slot(h, "stem.conv.weight", SK_STEM_W, {ch[0], 3, 1, 1});
    add_slot(h, "stem.conv.bias", SK_STEM_B, {ch[0]});
    for (int s = 0; s < 4; ++s) {
        for (int i = 0; i < h->enc[s]; ++i) {
            h->enc_blocks[s].emplace_back(new BlockW());
            add_block(h, h->enc_blocks[s].back().get(), "unet.encoder.stage" + std::to_string(s + 1) + "." + std::to_string(i), ch[s]);
        }
    }
    for (int s = 0; s < 3; ++s) {  // model.py:388-390, 857-863
        plan_conv(h->crush[s], dtype, MODE_GEMM1, ch[s + 1], ch[s], 2, 2, OUT_PLAIN, SRC_CRUSH, 0, 0);
        add_slot(h, "unet.encoder.downsample" + std::to_string(s + 1) + ".conv.weight", SK_CONV, {ch[s + 1], ch[s], 2, 2});
        h->slots.back().conv = &h->crush[s];
    }
    plan_conv(h->qa_conv, dtype, MODE_CONV3, cfg->num_deg_features, ch[3], 3, 3, OUT_PLAIN, SRC_PLAIN, 0, 0);  // :1010
    add_slot(h, "unet.encoder.qa_head.conv.weight", SK_CONV, {cfg->num_deg_features, ch[3], 3, 3});
    h->slots.back().conv = &h->qa_conv;
    add_slot(h, "unet.encoder.qa_head.conv.bias", SK_QA_B, {cfg->num_deg_features});
    for (int d = 0; d < 4; ++d) {  // decoder stage1 = coarsest level (model.py:290-300)
        const int lvl = 3 - d;
        for (int i = 0; i < h->dec[lvl]; ++i) {
            h->dec_blocks[d].emplace_back(new BlockW());
            add_block(h, h->dec_blocks[d].back().get(), "unet.decoder.stage" + std::to_string(d + 1) + "." + std::to_string(i), ch[lvl]);
        }
    }
    for (int d = 0; d < 3; ++d) {
        const int cin = ch[3 - d], cout = ch[2 - d];
        plan_conv(h->up[d], dtype, MODE_CONV3, 4 * cout, cin, 3, 3, OUT_D2S, SRC_PLAIN, 0, 0);  // model.py:569-571, 900-911
        add_slot(h, "unet.decoder.upsample" + std::to_string(d + 1) + ".conv.weight", SK_CONV, {4 * cout, cin, 3, 3});
        h->slots.back().conv = &h->up[d];
    }
    for (int d = 0; d < 3; ++d) {
        const int cout = ch[2 - d];
        plan_conv(h->skipmix[d], dtype, MODE_GEMM1, cout, 2 * cout, 1, 1, OUT_PLAIN, SRC_CONCAT, cout, cout);  // model.py:573-575
        add_slot(h, "unet.decoder.skip" + std::to_string(d + 1) + ".alpha", SK_ALPHA, {});
        h->slots.back().alpha = &h->skip_alpha[d];
        h->slots.back().flag = &h->skip_alpha_set[d];
        add_slot(h, "unet.decoder.skip" + std::to_string(d + 1) + ".conv.weight", SK_CONV, {cout, 2 * cout, 1, 1});
        h->slots.back().conv = &h->skipmix[d];
    }
    for (int i = 0; i < h->nhead; ++i) {  // model.py:945-954, 981-983
        h->head_blocks.emplace_back(new BlockW());
        add_block(h, h->head_blocks.back().get(), "head.layers." + std::to_string(i) + ".refiner", ch[0]);
        const bool last = i == h->nhead - 1;
        const int cout = last ? 3 : ch[0];
        h->head_up.emplace_back(new ConvW());
        plan_conv(*h->head_up.back(), dtype, MODE_CONV3, 4 * cout, ch[0], 3, 3, last ? OUT_FINAL : OUT_D2S, SRC_PLAIN, 0, 0);
        add_slot(h, "head.layers." + std::to_string(i) + ".upscale.conv.weight", SK_CONV, {4 * cout, ch[0], 3, 3});
        h->slots.back().conv = h->head_up.back().get();
    }
    *out = h;
    return MZ_OK;
}

static void free_conv(ConvW& c) {
    if (c.packed) (void)hipFree(c.packed);
    c.packed = nullptr;
    if (c.packed16) (void)hipFree(c.packed16);
    c.packed16 = nullptr;
    if (c.packed16r) (void)hipFree(c.packed16r);
    c.packed16r = nullptr;
    if (c.packed16t) (void)hipFree(c.packed16t);
    c.packed16t = nullptr;
}

extern "C" int mz_destroy(mz_handle* h) {
    if (!h) return MZ_OK;
    for (auto& s : h->slots) {
        if (s.kind == SK_CONV && s.conv) free_conv(*s.conv);
        if (s.kind == SK_CONV && s.block && s.conv == &s.block->mix) free_conv(s.block->mixf);
    }
    if (h->stem_w4) (void)hipFree(h->stem_w4);
    if (h->qa_bias) (void)hipFree(h->qa_bias);
    if (h->zero_page) (void)hipFree(h->zero_page);
    for (auto& r : h->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    delete h;
    return MZ_OK;
}

extern "C" int mz_num_weights(const mz_handle* h) { return h ? (int)h->slots.size() : 0; }

extern "C" int mz_weight_info(const mz_handle* h, int index, const char** name, int64_t shape[4]) {
    if (!h || index < 0 || index >= (int)h->slots.size()) return fail(MZ_ERR_INVALID_ARGUMENT, "bad weight index");
    const Slot& s = h->slots[index];
    if (name) *name = s.name.c_str();
    if (shape)
        for (int i = 0; i < 4; ++i) shape[i] = s.shape[i];
    return s.ndim;
}

static int prepare_device(mz_handle* h, hipStream_t st) {
    if (h->device_ready) return MZ_OK;
    int rc = ensure_device_ready();
    if (rc) return rc;
    // zero fills go to the CALLER's stream, like every later use of these buffers (a blocking memset on the NULL stream
    // is not ordered with work on a non-blocking stream)
    HIPCHK(hipMalloc(&h->zero_page, 4096));
    HIPCHK(hipMemsetAsync(h->zero_page, 0, 4096, st));
    const int cp0 = pad16(h->ch[0]);
    HIPCHK(hipMalloc((void**)&h->stem_w4, sizeof(float) * 4 * cp0));
    HIPCHK(hipMemsetAsync(h->stem_w4, 0, sizeof(float) * 4 * cp0, st));
    HIPCHK(hipMalloc((void**)&h->qa_bias, sizeof(float) * std::max(1, h->cfg.num_deg_features)));
    h->device_ready = true;
    return MZ_OK;
}

static int pack_conv(ConvW& c, int dtype, const float* w_dev, hipStream_t s) {
    if (!c.packed) HIPCHK(hipMalloc(&c.packed, c.packed_sz));
    PackArgs p;
    p.w = w_dev; p.dst = c.packed; p.dtype = dtype;
    p.cout = c.cout; p.cin = c.cin; p.kh = c.kh; p.kw = c.kw;
    p.taps = c.taps; p.nt = c.nt; p.ntiles = c.ntiles; p.nchunks = c.nchunks;
    p.out_map = c.out_map; p.cq = c.cq; p.cq_p = c.cq_p;
    p.in_map = c.in_map; p.c0 = c.c0; p.cp0 = c.cp0; p.c1 = c.c1;
    p.frag16 = 0; p.nfr = 0;
    HIPCHK(launch_pack(p, s));
    if (c.packed16_sz) {
        if (!c.packed16) HIPCHK(hipMalloc(&c.packed16, c.packed16_sz));
        p.dst = c.packed16; p.frag16 = 1; p.nchunks = c.nchunks16;
        if (c.in_map == SRC_CONCAT) { p.nt = 6; p.ntiles = c.cout / 192; }  // mix16_kernel: 12 fragments per K step
        HIPCHK(launch_pack(p, s));
        if (c.in_map == SRC_CONCAT && c.nchunks16 == 12) {  // C = 192, mix16b_kernel: rows in B-operand order, own channels first
            if (!c.packed16r) HIPCHK(hipMalloc(&c.packed16r, c.packed16_sz));
            p.dst = c.packed16r; p.frag16 = 3;
            HIPCHK(launch_pack(p, s));
        }
        if (c.in_map == SRC_MIXF && c.nt == 3) {
            if (!c.packed16r) HIPCHK(hipMalloc(&c.packed16r, c.packed16_sz));
            p.dst = c.packed16r; p.frag16 = 2;
            HIPCHK(launch_pack(p, s));
        }
    }
    if (c.packed16t_sz) {  // conv3t_kernel: three 16-channel fragments per tap (its gate: PackArgs::frag16 = 4)
        if (!c.packed16t) HIPCHK(hipMalloc(&c.packed16t, c.packed16t_sz));
        p.dst = c.packed16t; p.nt = c.nt; p.ntiles = 1; p.nfr = 3;
        if (c.in_map == SRC_MIXF) { p.frag16 = 4; p.nchunks = 3; }
        else { p.frag16 = 1; p.nchunks = c.nchunks16t; }
        HIPCHK(launch_pack(p, s));
    }
    c.set = true;
    return MZ_OK;
}

extern "C" int mz_set_weight(mz_handle* h, const char* name, const float* dev_f32, const int64_t* shape, int ndim,
                             void* hip_stream) {
    if (!h || !name || !dev_f32) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    auto it = h->slot_index.find(name);
    if (it == h->slot_index.end()) return fail(MZ_ERR_UNKNOWN_WEIGHT, "unknown parameter '%s'", name);
    Slot& s = h->slots[it->second];
    if (ndim != s.ndim) return fail(MZ_ERR_SHAPE_MISMATCH, "'%s': expected %d dims, got %d", name, s.ndim, ndim);
    for (int i = 0; i < ndim; ++i)
        if (shape[i] != s.shape[i])
            return fail(MZ_ERR_SHAPE_MISMATCH, "'%s': dim %d is %lld, expected %lld", name, i, (long long)shape[i], (long long)s.shape[i]);
    hipStream_t st = (hipStream_t)hip_stream;
    int rc = prepare_device(h, st);
    if (rc) return rc;
    switch (s.kind) {
        case SK_CONV: {
            int rc2 = pack_conv(*s.conv, h->dtype, dev_f32, st);
            if (rc2 == MZ_OK && s.block && s.block->fused && s.conv == &s.block->mix)
                rc2 = pack_conv(s.block->mixf, h->dtype, dev_f32, st);
            return rc2;
        }
        case SK_ALPHA: {
            // sigmoid(alpha) is folded on the host (model.py:833); one 4-byte read at load time.
            float v = 0.f;
            HIPCHK(hipMemcpyAsync(&v, dev_f32, sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            *s.alpha = v;
            *s.flag = true;
            return MZ_OK;
        }
        case SK_STEM_W:
            HIPCHK(launch_pack_stem(dev_f32, nullptr, h->stem_w4, h->ch[0], pad16(h->ch[0]), st));
            h->stem_w_set = true;
            return MZ_OK;
        case SK_STEM_B:
            HIPCHK(launch_pack_stem(nullptr, dev_f32, h->stem_w4, h->ch[0], pad16(h->ch[0]), st));
            h->stem_b_set = true;
            return MZ_OK;
        case SK_QA_B:
            HIPCHK(hipMemcpyAsync(h->qa_bias, dev_f32, sizeof(float) * h->cfg.num_deg_features, hipMemcpyDeviceToDevice, st));
            h->qa_b_set = true;
            return MZ_OK;
    }
    return fail(MZ_ERR_INVALID_ARGUMENT, "bad slot");
}

extern "C" int mz_weights_complete(const mz_handle* h) {
    if (!h) return fail(MZ_ERR_INVALID_ARGUMENT, "null handle");
    for (const Slot& s : h->slots) {
        bool ok = true;
        switch (s.kind) {
            case SK_CONV: ok = s.conv->set; break;
            case SK_ALPHA: ok = *s.flag; break;
            case SK_STEM_W: ok = h->stem_w_set; break;
            case SK_STEM_B: ok = h->stem_b_set; break;
            case SK_QA_B: ok = h->qa_b_set; break;
        }
        if (!ok) return fail(MZ_ERR_MISSING_WEIGHTS, "parameter '%s' has not been set", s.name.c_str());
    }
    return MZ_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------
struct Plan {
    int nb;                // images per micro-batch
    int hs[4], ws[4];      // level sizes
    size_t R[4][3], HID[4], Z[4], U[3];
    size_t HR[3][2], HHID[3], HZ[3];  // head levels 1..nhead-1 (index j-1... stored at j)
    size_t QA;
    size_t total;
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

static void make_plan(const mz_handle* h, int nb, int H, int W, Plan& p) {
    const size_t sz = dtype_size(h->dtype);
    const int hr = h->cfg.hidden_ratio;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes);
        return o;
    };
    p.nb = nb;
    p.hs[0] = H; p.ws[0] = W;
    for (int i = 1; i < 4; ++i) { p.hs[i] = p.hs[i - 1] / 2; p.ws[i] = p.ws[i - 1] / 2; }
    for (int l = 0; l < 4; ++l) {
        const size_t px = (size_t)nb * p.hs[l] * p.ws[l];
        const size_t c = px * pad16(h->ch[l]) * sz;
        for (int k = 0; k < 3; ++k) p.R[l][k] = take(c);
        p.HID[l] = take(px * pad16(hr * h->ch[l]) * sz);
        p.Z[l] = take(c);
        if (l < 3) p.U[l] = take(c);
    }
    for (int j = 1; j < h->nhead; ++j) {
        const size_t px = (size_t)nb * (H << j) * (W << j);
        const size_t c = px * pad16(h->ch[0]) * sz;
        p.HR[j][0] = take(c);
        p.HR[j][1] = take(c);
        p.HHID[j] = take(px * pad16(hr * h->ch[0]) * sz);
        p.HZ[j] = take(c);
    }
    p.QA = take((size_t)nb * p.hs[3] * p.ws[3] * pad16(h->cfg.num_deg_features) * sz);
    p.total = off;
}

static int default_micro_batch(const mz_handle* h, int B, int H, int W, int requested) {
    if (requested > 0) return std::min(B, requested);
    // keep a micro-batch's workspace around <= 48 GiB by default (288 GB of HBM per GPU)
    Plan p;
    make_plan(h, 1, H, W, p);
    const size_t budget = (size_t)48 << 30;
    int nb = (int)std::max<size_t>(1, budget / std::max<size_t>(1, p.total));
    return std::max(1, std::min(B, nb));
}

extern "C" int mz_workspace_bytes(const mz_handle* h, int B, int H, int W, int max_images_in_flight, size_t* bytes) {
    if (!h || !bytes) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    if (B <= 0 || H < 8 || W < 8) return fail(MZ_ERR_INVALID_ARGUMENT, "need B >= 1 and H, W >= 8 (got %d, %d, %d)", B, H, W);
    Plan p;
    make_plan(h, default_micro_batch(h, B, H, W, max_images_in_flight), H, W, p);
    *bytes = p.total;
    return MZ_OK;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// Diagnostic stamp buffer (only -DMZ_DIAG kernel builds write to it, mz_diag.h; MZ_DEBUG_STAMPS=1 allocates it).
static unsigned long long* debug_buffer() {
    static unsigned long long* buf = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        if (getenv("MZ_DEBUG_STAMPS")) {
            if (hipMalloc((void**)&buf, 16 * 64 * 8 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
            else (void)hipMemset(buf, 0, 16 * 64 * 8 * sizeof(unsigned long long));
        }
    }
    return buf;
}
extern "C" int mz_debug_read(unsigned long long* host_dst) {
    unsigned long long* b = debug_buffer();
    if (!b) return -1;
    return hipMemcpy(host_dst, b, 16 * 64 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -6;
}

// workgroups of a persistent launch: one per CU of the CURRENT device, a multiple of 8 (one equal share per XCD).
// Knobs::persist overrides: 0 = one workgroup per tile everywhere (A/B timing); n = force n (tests use 8 / 16 so that
// small images walk several tiles per workgroup).
static int persistent_workgroups(const Knobs& k) {
    if (k.persist >= 0) return k.persist;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    return g_dev_cus[dev];
}

// 1 / sigmoid(alpha) = 1 + e^-alpha for blend_() (mz_device.h), which folds the scale into the reciprocal of the gate's sigmoid:
// rcp(fma(e^-beta, inv_s, inv_s)).  Kept finite: for alpha < -88.7 the exact value overflows to +inf and fma(0, inf, inf) (a gate
// whose e^-beta flushed to 0) would be NaN where the reference (model.py:833-837) returns x; with FLT_MAX the weight is ~0 instead.
static float inv_sigmoid(float alpha) {
    const float v = 1.0f + std::exp(-alpha);
    return std::isfinite(v) ? v : 3.402823466e+38f;
}

// The tiles of a launch in walk order, two words each: {y0 | x0 << 16, image | N tile << 16}.  The order is the group walk of
// mz_device.h (tile_of / tile_rc): ids 0 .. a.grid - 1 in groups of gm pixel tiles x gn N tiles, the tiles of an image in block rows of
// four tile rows where a.blk4 is set, padding ids of partial groups dropped.
static void tile_list(const ConvArgs& a, int th, int tw, std::vector<uint32_t>& t) {
    const int tpi = a.tiles_x * a.tiles_y, gsz = a.gm * a.gn;
    for (int L = 0; L < a.grid; ++L) {
        const int group = L / gsz, within = L % gsz;
        const int gi_n = group / a.groups_m, gi_m = group % a.groups_m;
        const int mt = gi_m * a.gm + within / a.gn, nt = gi_n * a.gn + within % a.gn;
        if (mt >= a.mtiles || nt >= a.ntiles) continue;
        const int b = mt / tpi, trem = mt % tpi;
        int tyi, txi;
        if (!a.blk4) {
            tyi = trem / a.tiles_x; txi = trem % a.tiles_x;
        } else {
            const int bsz = 4 * a.tiles_x, br = trem / bsz, rem = trem % bsz;
            const int rows = std::min(4, a.tiles_y - 4 * br);
            txi = rem / rows; tyi = 4 * br + rem % rows;
        }
        t.push_back((uint32_t)(tyi * th) | (uint32_t)(txi * tw) << 16);
        t.push_back((uint32_t)b | (uint32_t)nt << 16);
    }
}

struct Runner {
    mz_handle* h;
    hipStream_t s;
    int dtype;
    int rc = MZ_OK;
    const Knobs knobs = h->knobs;
    bool wide_tiles = knobs.wide;
    bool no_fuse = !knobs.fuse;
    int io_u8 = 0;                                        // images at both ends are uint8 (mz_forward_u8)
    const float* film_gamma = nullptr;                    // mz_op_conv_film: per-image per-channel affine on the next conv3 call
    const float* film_beta = nullptr;
    bool use_s16 = knobs.s16;
    int persist_wgs = persistent_workgroups(knobs);       // 0: one workgroup per tile everywhere

    void prof_begin(ProfRec*& r, double flops, double bytes, int is_conv3) {
        r = nullptr;
        if (!h || !h->prof) return;
        if (h->recs_used == h->recs.size()) {
            ProfRec n;
            if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
            h->recs.push_back(n);
        }
        r = &h->recs[h->recs_used++];
        r->flops = flops; r->bytes = bytes; r->is_conv3 = is_conv3;
        r->kind = r->B = r->H = r->W = r->cin = r->cout = r->nt = r->ntiles = r->mtiles = r->n_fast = 0;
        (void)hipEventRecord(r->a, s);
    }
    void prof_end(ProfRec* r) {
        if (r) (void)hipEventRecord(r->b, s);
    }

    int check(hipError_t e, const char* what) {
        if (e != hipSuccess && rc == MZ_OK) rc = fail(MZ_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
        return rc;
    }

    void base_args(ConvArgs& a, const ConvW& c) {
        memset(&a, 0, sizeof(a));
        a.wpk = c.packed;
        a.zero = h ? h->zero_page : nullptr;
        a.dbg = debug_buffer();
        a.nchunks = c.nchunks;
        a.nchunks_real = c.nchunks_real;
        a.ntiles = c.ntiles;
        a.use_glds = knobs.use_glds;
    }

    void pick_order(ConvArgs& a, const ConvW& c, double act_bytes, int resident_per_xcd = 32) {
        // Tile groups of gm pixel tiles x gn N tiles (gm * gn ~ the workgroups resident on one XCD): inside a group
        // both operands are shared through the XCD's L2; per group the activations are re-read ntiles/gn times and
        // the weights mtiles/gm times in total.  Pick the shape with the least total re-read traffic.
        const double W = (double)c.packed_sz, A = act_bytes;
        int best_gm = a.mtiles, best_gn = 1;
        double best = 1e300;
        for (int gn = 1; gn <= a.ntiles; ++gn) {
            if (gn > resident_per_xcd) break;
            if (a.ntiles % gn != 0 && gn != a.ntiles) continue;
            int gm = resident_per_xcd / gn;
            if (gm < 1) gm = 1;
            if (gm > a.mtiles) gm = a.mtiles;
            const double groups_n = std::ceil((double)a.ntiles / gn), groups_m = std::ceil((double)a.mtiles / gm);
            const double traffic = A * groups_n + W * groups_m;
            if (traffic < best) { best = traffic; best_gm = gm; best_gn = gn; }
        }
        a.gm = best_gm; a.gn = best_gn;
        const long long groups = (long long)((a.mtiles + a.gm - 1) / a.gm) * ((a.ntiles + a.gn - 1) / a.gn);
        a.grid = (int)(groups * a.gm * a.gn);
        a.groups_m = (a.mtiles + a.gm - 1) / a.gm;
        a.inv_gsz = 1.0f / (float)(a.gm * a.gn);
        a.inv_groups_m = 1.0f / (float)a.groups_m;
        a.inv_gn = 1.0f / (float)a.gn;
        a.inv_tpi = a.tiles_x > 0 ? 1.0f / (float)(a.tiles_x * a.tiles_y) : 1.0f;
        a.inv_tiles_x = a.tiles_x > 0 ? 1.0f / (float)a.tiles_x : 1.0f;
        a.inv_bsz = a.tiles_x > 0 ? 1.0f / (float)(4 * a.tiles_x) : 1.0f;
        a.blk4 = knobs.blk4 && a.tiles_x > 0 && 4 * a.tiles_x < 65536 ? 1 : 0;  // the tile walk inside an image: conv3s_kernel, and the tile lists of conv3r / conv3t
        auto magic = [](long long d) { return d <= 1 ? 0xffffffffu : (uint32_t)(4294967296ULL / (unsigned long long)d); };
        a.mg_gsz = magic((long long)a.gm * a.gn); a.mg_groups_m = magic(a.groups_m); a.mg_gn = magic(a.gn);
        a.mg_tpi = magic(a.tiles_x > 0 ? (long long)a.tiles_x * a.tiles_y : 1); a.mg_tiles_x = magic(a.tiles_x > 0 ? a.tiles_x : 1);
        a.mg_bsz = magic(a.tiles_x > 0 ? 4LL * a.tiles_x : 1);
    }

    // conv3r_kernel / conv3t_kernel: the launch's tiles in walk order as a table in HBM (ConvArgs::tile_tab), so that the kernels'
    // helper role -- the critical path of their short tiles -- reads a tile's coordinates with one scalar load instead of running
    // the divisions of the group walk (tile_of_s / tile_rc_s, mz_device.h) three times per phase.  The order IS that walk's: ids
    // 0 .. grid - 1 in gm x gn groups, the tiles of an image in block rows of four tile rows (blk4), padding ids dropped.  Needs
    // pick_order() done; sets a.tile_tab, a.grid (= tiles listed) and a.persist.  One table per geometry, kept with the handle.
    void tile_table(ConvArgs& a, int th, int tw) {
        if (rc) return;
        if (a.B >= 65536 || a.ntiles >= 65536 || a.tiles_y * th >= 65536 || a.tiles_x * tw >= 65536) {
            rc = fail(MZ_ERR_INVALID_ARGUMENT, "tile table: image, batch or N-tile index beyond 16 bits");
            return;
        }
        const int pad = 4 * ((persist_wgs > 256 ? persist_wgs : 256) / 8) + 8;
        const std::vector<int> key = {th, tw, a.B, a.tiles_x, a.tiles_y, a.ntiles, a.gm, a.gn, a.grid, a.blk4, pad};
        auto it = h->tile_tabs.find(key);
        if (it == h->tile_tabs.end()) {
            std::vector<uint32_t> t;
            t.reserve(2 * ((size_t)a.mtiles * a.ntiles + pad));
            tile_list(a, th, tw, t);
            const int n = (int)(t.size() / 2);
            t.resize(t.size() + 2 * (size_t)pad, 0u);
            void* d = nullptr;
            if (check(hipMalloc(&d, t.size() * 4), "tile table")) return;
            if (check(hipMemcpy(d, t.data(), t.size() * 4, hipMemcpyHostToDevice), "tile table upload")) { (void)hipFree(d); return; }
            it = h->tile_tabs.emplace(key, std::make_pair(d, n)).first;
        }
        a.tile_tab = it->second.first;
        a.grid = it->second.second;
        const int need = (a.grid + 7) / 8 * 8;
        a.persist = need < persist_wgs ? need : persist_wgs;
    }

    // conv3x3, pad 1 (model.py:742-748, 900-909, 1010). epi: STORE / D2S / FINAL
    void conv3(const ConvW& c, const void* in, void* out, int B, int H, int W, int epi, int silu, int Hout, int Wout,
               const void* img = nullptr, int R = 0, int clamp = 0, const void* zero_override = nullptr,
               const ConvW* mixf = nullptr, const void* xin = nullptr, float alpha = 0.f) {
        if (rc) return;
        ConvArgs a;
        base_args(a, c);
        if (zero_override) a.zero = zero_override;
        a.in0 = in; a.out = out;
        a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W;
        a.p0 = c.cp0 * dtype_size(dtype) / 16;
        a.src = SRC_PLAIN;
        // tile shape: the 512-pixel kernels (NT <= 3) in the shape that wastes fewer padded pixels, else 8 x 32
        int mode = MODE_CONV3, th = 8, tw = 32;
        if (c.nt <= 3 && wide_tiles) {
            const long long waste16 = (long long)((H + 15) / 16 * 16) * ((W + 31) / 32 * 32);
            const long long waste8 = (long long)((H + 7) / 8 * 8) * ((W + 63) / 64 * 64);
            if (waste8 <= waste16) { mode = MODE_C3W8; th = 8; tw = 64; }
            else { mode = MODE_C3W16; th = 16; tw = 32; }
        }
        // The image head (12 output channels + PixelShuffle + bicubic skip + clamp) is a per-tile kernel whose load, K loop and long
        // epilogue run one after the other: on 512-pixel tiles (183 KB of LDS) a CU holds ONE workgroup and nothing overlaps; on the
        // 256-pixel kernel several fit and one tile's epilogue runs under another's loads (2160 x 3840, Cin = 96: 2.34 -> 1.60 ms per 3
        // images).  Chosen by dtype and knobs only, never by the image size.
        if (epi == EPI_FINAL && knobs.head256 && dtype != DT_F32) { mode = MODE_CONV3; th = 8; tw = 32; }
        a.tiles_x = (W + tw - 1) / tw; a.tiles_y = (H + th - 1) / th;
        a.mtiles = B * a.tiles_x * a.tiles_y;
        a.epi = epi; a.silu = silu;
        a.cp_out = epi == EPI_D2S ? c.cq_p : pad16(c.cout);
        a.p_out = a.cp_out * dtype_size(dtype) / 16;
        a.Hout = Hout; a.Wout = Wout;
        a.img = img; a.R = R; a.clamp = clamp;
        if (epi == EPI_FINAL) { a.Hi = Hout / R; a.Wi = Wout / R; a.io_u8 = io_u8; }
        double extra_flops = 0.0;
        if (epi == EPI_FUSEDMIX) {
            a.in1 = xin;
            a.p1 = pad16(c.cout) * dtype_size(dtype) / 16;
            a.wmix = mixf->packed;
            a.mix_pieces = mixf->nchunks * mixf->nt;
            {   // room for the 8 compute waves' x fragments next to the gate weights in ring slots 1-2?
                const int a_slot = (mode == MODE_C3W16 ? 2 * 640 : 2 * 672) * 16;
                const int slot = a_slot + 9 * c.nt * 1024;
                const int ncx = a.p1 / 2;
                a.x_via_lds = (a.mix_pieces * 1024 + 8 * ncx * 1024 <= 2 * slot) ? 1 : 0;
            }
            a.mix_scale = 1.0f / (1.0f + std::exp(-alpha));
            a.inv_mix_scale = inv_sigmoid(alpha);
            extra_flops = 2.0 * (double)B * H * W * 2.0 * c.cout * c.cout;
        }
        const double sz = dtype_size(dtype);
        const double px = (double)B * H * W;
        // what conv3r_kernel's plain variants need in common: 16-bit type, 96-channel N tiles, the 16x16x32 packing, K padding within the knob
        const bool r_common = use_s16 && !film_gamma && dtype != DT_F32 && c.nt == 3 && c.packed16 && (epi == EPI_STORE || epi == EPI_D2S) &&
                        persist_wgs > 0 && c.nchunks16 * 32 * 100 <= c.cp0 * (100 + knobs.kpad_pct) &&
                        (double)H * W * 64.0 < 4294967296.0;
        // conv3t_kernel: ONE N tile of 33..48 channels (the level-1 block of the 48-channel models), whole 32-channel chunks, three or six
        // and more of them; 12 x 64 pixel tiles; stores and x loads carry 32-bit offsets inside six planes.  The choice depends on channel
        // counts only (never on H or W): its fused variant sums the gate in another order than conv3s_kernel<.., FUSE> -- equal to <= 1 ulp,
        // not bit for bit --, and a tile of upscale_tiled() must run the kernel the whole image runs.
        const bool t_fuse = epi == EPI_FUSEDMIX && knobs.fuse16 && mixf && mixf->packed16t;
        const bool use_t = knobs.t && use_s16 && !film_gamma && dtype != DT_F32 && c.packed16t && c.ntiles == 1 && persist_wgs > 0 &&
                           (c.nchunks16t == 3 || c.nchunks16t >= 6) && ((epi == EPI_STORE) || t_fuse) &&
                           (double)H * W * 64.0 < 4294967296.0 && 6.0 * H * W * 16.0 < 4294967296.0;
        if (use_t) {
            a.tiles_x = (W + 63) / 64; a.tiles_y = (H + 11) / 12;
            a.mtiles = B * a.tiles_x * a.tiles_y;
            pick_order(a, c, px * c.cp0 * sz);
            a.s16 = 1; a.wpk16 = c.packed16t; a.nchunks16 = c.nchunks16t;
            if (t_fuse) a.wmix16 = mixf->packed16t;
            tile_table(a, 12, 64);
            if (rc) return;
            const double extra_bytes = epi == EPI_FUSEDMIX ? px * c.cout * sz : 0.0;
            ProfRec* r;
            prof_begin(r, 2.0 * px * 9.0 * c.cin * c.cout + extra_flops, px * (c.cin + c.cout) * sz + 9.0 * c.cin * c.cout * sz + extra_bytes, 1);
            if (r) { r->kind = 0; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
            g_last_kernel = t_fuse ? "conv3t_fused" : "conv3t";
            check(launch_conv3t(dtype, a, s), "conv3t launch");
            prof_end(r);
            return;
        }
        // conv3r_kernel's ragged variant: conv1 + SiLU with Cin = 48 (two 32-channel chunks, the second with two real planes) into 96-channel
        // N tiles.  The kernel it replaces (conv3p_kernel: 32x32x16 MFMA, exact 16-channel chunks) sums in another order, so the choice
        // depends on channel counts, dtype and knobs only -- never on H or W.
        const bool use_r2 = knobs.r && knobs.r2 && use_s16 && !film_gamma && dtype != DT_F32 && c.nt == 3 && c.packed16 && epi == EPI_STORE && silu &&
                            persist_wgs > 0 && c.nchunks16 == 2 && c.cp0 == 48 && (double)H * W * 64.0 < 4294967296.0 &&
                            12.0 * H * W * 16.0 < 4294967296.0;
        if (use_r2) {
            a.tiles_x = (W + 47) / 48; a.tiles_y = (H + 7) / 8;
            a.mtiles = B * a.tiles_x * a.tiles_y;
            pick_order(a, c, px * c.cp0 * sz);
            a.s16 = 1; a.wpk16 = c.packed16; a.nchunks16 = 2;
            a.ragged_planes = (c.cp0 - 32) / 8;
            tile_table(a, 8, 48);
            if (rc) return;
            ProfRec* r;
            prof_begin(r, 2.0 * px * 9.0 * c.cin * c.cout, px * (c.cin + c.cout) * sz + 9.0 * c.cin * c.cout * sz, 1);
            if (r) { r->kind = 0; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
            g_last_kernel = "conv3r_ragged";
            check(launch_conv3r(dtype, a, s), "conv3r ragged launch");
            prof_end(r);
            return;
        }
        // padded pixels of conv3r's 8 x 48 tiles -- and of its second geometry, 8 x 40 (five pixel fragments per wave: widths
        // like 120 that 48 does not divide) -- against the better of the 8 x 64 / 16 x 32 tiles.  (All plain variants accumulate in the same
        // order whatever the tile shape: bit-identical, so this choice may depend on H and W.)
        const long long rows8 = (long long)((H + 7) / 8 * 8);
        const long long pad48 = rows8 * ((W + 47) / 48 * 48), pad40 = rows8 * ((W + 39) / 40 * 40);
        const long long pads = (long long)a.tiles_y * th * a.tiles_x * tw;
        // conv3r_kernel: any chunk count >= 3 of four whole planes (its halo loads carry the plane in the scalar offset, which the
        // hardware's range check does not cover); its stores carry 32-bit offsets inside 12 output planes / one D2S target image
        const bool r_ok = knobs.r && r_common && c.nchunks16 >= 3 && a.p0 % 4 == 0 &&
                          (epi == EPI_D2S ? (double)(c.cq_p * dtype_size(dtype) / 16) * Hout * Wout * 16.0 < 4294967296.0
                                          : 12.0 * H * W * 16.0 < 4294967296.0);
        const int geo = (r_ok && knobs.geo40 && pad40 < pad48) ? 1 : 0;
        const int tw_r = geo ? 40 : 48;
        const bool use_r = r_ok && (geo ? pad40 : pad48) <= pads;
        // ... and its fused variant (conv2 + AdaptiveResidualMix, C = 96): six or more chunks (one pixel fragment's gate GEMM and
        // blend per chunk), the gate weights packed in accumulator-row order, x and out within 32-bit offsets
        bool use_rf = knobs.r && knobs.fuse16 && epi == EPI_FUSEDMIX && use_s16 && dtype != DT_F32 && c.nt == 3 && c.ntiles == 1 && c.packed16 &&
                      mixf && mixf->packed16r && (mixf->cp0 + 31) / 32 == c.nt && persist_wgs > 0 && c.nchunks16 >= 6 && a.p0 % 4 == 0 &&
                      c.nchunks16 * 32 * 100 <= c.cp0 * (100 + knobs.kpad_pct) && (double)H * W * 64.0 < 4294967296.0 &&
                      12.0 * H * W * 16.0 < 4294967296.0;
        // (NOT a function of H and W: this kernel and conv3s_kernel<.., FUSE> sum the x half of the gate in different orders inside a
        // 32-wide K step -- equal to <= 1 ulp, not bit for bit -- and a tile of upscale_tiled() must run the kernel the whole image runs,
        // or "tiled == untiled bit for bit" (ultrazoom_amd/tiling.py) breaks.  The plain variants above ARE bit-identical to
        // conv3s_kernel, so their choice may follow the padded-pixel count.)
        if (use_rf) {
            a.tiles_x = (W + 47) / 48; a.tiles_y = (H + 7) / 8;
            a.mtiles = B * a.tiles_x * a.tiles_y;
            pick_order(a, c, px * c.cp0 * sz);
            a.s16 = 1; a.wpk16 = c.packed16; a.nchunks16 = c.nchunks16;
            a.wmix16 = mixf->packed16r;
            tile_table(a, 8, 48);
            if (rc) return;
            ProfRec* r;
            prof_begin(r, 2.0 * px * 9.0 * c.cin * c.cout + extra_flops, px * (c.cin + c.cout) * sz + 9.0 * c.cin * c.cout * sz + px * c.cout * sz, 1);
            if (r) { r->kind = 0; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
            g_last_kernel = "conv3r_fused";
            check(launch_conv3r(dtype, a, s), "conv3r fused launch");
            prof_end(r);
            return;
        }
        if (use_r) {
            a.geo = geo;
            a.tiles_x = (W + tw_r - 1) / tw_r; a.tiles_y = (H + 7) / 8;
            a.mtiles = B * a.tiles_x * a.tiles_y;
            pick_order(a, c, px * c.cp0 * sz);
            a.s16 = 1; a.wpk16 = c.packed16; a.nchunks16 = c.nchunks16;
            tile_table(a, 8, tw_r);
            if (rc) return;
            ProfRec* r;
            prof_begin(r, 2.0 * px * 9.0 * c.cin * c.cout, px * (c.cin + c.cout) * sz + 9.0 * c.cin * c.cout * sz, 1);
            if (r) { r->kind = 0; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
            g_last_kernel = a.geo ? "conv3r_8x40" : "conv3r";
            check(launch_conv3r(dtype, a, s), "conv3r launch");
            prof_end(r);
            return;
        }
        pick_order(a, c, px * c.cp0 * sz);
        const bool fuse16 = epi == EPI_FUSEDMIX && mixf && mixf->packed16 && knobs.fuse16 &&
                            (mixf->cp0 + 31) / 32 == c.nt;  // x K-steps == z K-steps (always so for C <= 96)
        if (mode != MODE_CONV3 && (epi == EPI_STORE || epi == EPI_D2S || fuse16) && persist_wgs > 0) {
            // 16-bit types: the 16x16x32-MFMA kernel (persistent only; 32-bit halo offsets span four planes)
            // ... and only where padding K to whole 32-channel chunks wastes less than the shape gains (~12 %)
            const bool k_fits = c.nchunks16 * 32 * 100 <= c.cp0 * (100 + knobs.kpad_pct);
            if (c.packed16 && use_s16 && k_fits && (double)H * W * 64.0 < 4294967296.0) {
                a.s16 = 1; a.wpk16 = c.packed16; a.nchunks16 = c.nchunks16;
                if (fuse16) a.wmix16 = mixf->packed16;
                const int need = (a.grid + 7) / 8 * 8;
                a.persist = need < persist_wgs ? need : persist_wgs;
            } else if (a.grid > persist_wgs && (double)H * W * 32.0 < 4294967296.0) {
                // conv3p_kernel: 32-bit halo offsets span the two planes of a 16-channel stage; larger images stay on
                // the per-tile kernel (64-bit addresses)
                a.persist = persist_wgs;
            }
        }
        if (film_gamma) {
            if (!a.s16) {
                rc = fail(MZ_ERR_INVALID_ARGUMENT, "the FiLM epilogue exists on the 16x16x32 kernel only: bf16 / fp16, at most 96 output "
                                                   "channels per N tile, input channels within 12.5 %% of a multiple of 32");
                return;
            }
            a.film_gamma = film_gamma; a.film_beta = film_beta;
        }
        // algorithmic bytes: input once, output once, weights once; a fused conv2 + mix also reads the block input x once
        const double extra_bytes = epi == EPI_FUSEDMIX ? px * c.cout * sz : 0.0;
        ProfRec* r;
        prof_begin(r, 2.0 * px * 9.0 * c.cin * c.cout + extra_flops, px * (c.cin + c.cout) * sz + 9.0 * c.cin * c.cout * sz + extra_bytes, 1);
        if (r) { r->kind = 0; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
        g_last_kernel = mode == MODE_CONV3 ? "conv_kernel" : (a.persist > 0 ? (a.s16 ? (epi == EPI_FUSEDMIX ? "conv3s_fused" : "conv3s") : "conv3p")
                                                                            : (epi == EPI_FUSEDMIX ? "conv3w_fused" : "conv3w"));
        check(launch_conv(dtype, mode, c.nt, a, s), "conv3x3 launch");
        prof_end(r);
    }

    // AdaptiveResidualMix (model.py:826-839): out = x + sigmoid(alpha)*sigmoid(W[x;z])*(z - x)
    void mix(const ConvW& c, float alpha, const void* x, const void* z, void* out, int B, int H, int W) {
        if (rc) return;
        ConvArgs a;
        base_args(a, c);
        a.in0 = x; a.in1 = z; a.out = out;
        a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W;
        const int sz = dtype_size(dtype);
        a.p0 = c.cp0 * sz / 16; a.p1 = pad16(c.c1) * sz / 16;
        a.nchunks0 = c.cp0 / chunk_channels(dtype);
        a.src = SRC_CONCAT;
        const long long npix = (long long)B * H * W;
        a.mtiles = (int)((npix + 255) / 256);
        a.epi = EPI_MIX;
        a.cp_out = pad16(c.cout);
        a.p_out = a.cp_out * sz / 16;
        a.mix_scale = 1.0f / (1.0f + std::exp(-alpha));
        a.inv_mix_scale = inv_sigmoid(alpha);
        const bool mix16 = c.packed16 != nullptr && knobs.mix16 &&
                           (double)npix * c.cp0 * sz < 4294967296.0;  // 32-bit buffer offsets inside each tensor
        int mix16b_wgs = knobs.persist > 0 ? knobs.persist : 0;  // persistent (also under MZ_NO_PERSIST=1: it has no per-tile form)
        if (mix16b_wgs == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) mix16b_wgs = g_dev_cus[dev];
        }
        const bool mix16b = mix16 && knobs.mix16b && c.packed16r != nullptr && c.nchunks16 == 12 && mix16b_wgs > 0;  // C = 192
        if (mix16) {  // 192-channel N tiles, x / z straight into MFMA operands (mix16_kernel / mix16b_kernel)
            a.ntiles = c.cout / 192;
            a.wpk16 = mix16b ? c.packed16r : c.packed16;
            a.nchunks16 = c.nchunks16;
        }
        pick_order(a, c, (double)npix * (c.cp0 + pad16(c.c1)) * sz, mix16 ? 32 : 64);
        ProfRec* r;
        prof_begin(r, 2.0 * (double)npix * c.cin * c.cout, (double)npix * 3.0 * c.cout * sz, 0);
        if (r) { r->kind = 1; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
        g_last_kernel = mix16b ? "mix16b" : (mix16 ? "mix16" : "conv_kernel_mix");
        if (mix16b) check(launch_mix16b(dtype, a, s, mix16b_wgs), "mix16b launch");
        else if (mix16) check(launch_mix16(dtype, a, s), "mix16 launch");
        else check(launch_conv(dtype, MODE_GEMM1, c.nt, a, s), "mix launch");
        prof_end(r);
    }

    // PixelCrush (model.py:857-863, 881-882): conv 2x2 stride 2, floors odd sizes
    void crush(const ConvW& c, const void* in, void* out, int B, int H, int W, const void* zero_override = nullptr) {
        if (rc) return;
        ConvArgs a;
        base_args(a, c);
        if (zero_override) a.zero = zero_override;
        a.in0 = in; a.out = out;
        a.B = B; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
        a.p0 = c.cp0 * dtype_size(dtype) / 16;
        a.nchunks0 = c.cp0 / chunk_channels(dtype);
        a.src = SRC_CRUSH;
        const long long npix = (long long)B * a.Ho * a.Wo;
        a.mtiles = (int)((npix + 255) / 256);
        a.epi = EPI_STORE;
        a.cp_out = pad16(c.cout);
        a.p_out = a.cp_out * dtype_size(dtype) / 16;
        const double sz = dtype_size(dtype);
        pick_order(a, c, (double)B * H * W * c.cp0 * sz, 64);
        ProfRec* r;
        prof_begin(r, 2.0 * (double)npix * 4.0 * c.cin * c.cout, ((double)B * H * W * c.cin + (double)npix * c.cout) * sz, 0);
        if (r) { r->kind = 2; r->B = B; r->H = H; r->W = W; r->cin = c.cin; r->cout = c.cout; r->nt = c.nt; r->ntiles = a.ntiles; r->mtiles = a.mtiles; r->n_fast = a.gm * 1000 + a.gn; }
        check(launch_conv(dtype, MODE_GEMM1, c.nt, a, s), "crush launch");
        prof_end(r);
    }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static int forward_micro(mz_handle* h, const char* x, char* out_sr, float* out_qa, int nb, int H, int W, int clamp,
                         char* ws, hipStream_t s, int io_u8) {
    Plan p;
    make_plan(h, nb, H, W, p);
    Runner run{h, s, h->dtype};
    run.io_u8 = io_u8;
    const int r = h->cfg.upscale_ratio;

    auto block = [&](const BlockW& b, const void* xin, void* hid, void* z, void* yout, int hh, int ww) {
        // EncoderBlock / DecoderBlock (model.py:507-511): conv1 -> SiLU -> conv2 -> adaptive mix with the input
        run.conv3(b.conv1, xin, hid, nb, hh, ww, EPI_STORE, 1, 0, 0);
        if (b.fused && run.wide_tiles && !run.no_fuse) {
            // conv2 + AdaptiveResidualMix in one launch (all output channels live in one workgroup)
            run.conv3(b.conv2, hid, yout, nb, hh, ww, EPI_FUSEDMIX, 0, 0, 0, nullptr, 0, 0, nullptr, &b.mixf, xin, b.alpha);
        } else {
            run.conv3(b.conv2, hid, z, nb, hh, ww, EPI_STORE, 0, 0, 0);
            run.mix(b.mix, b.alpha, xin, z, yout, nb, hh, ww);
        }
    };

    // stem (model.py:158): NCHW image -> NHWC features
    char* cur = ws + p.R[0][0];
    if (hipError_t e = launch_stem(h->dtype, x, h->stem_w4, cur, nb, H, W, pad16(h->ch[0]), s, io_u8); e != hipSuccess)
        return fail(MZ_ERR_HIP, "stem launch: %s", hipGetErrorString(e));

    // encoder (model.py:461-484)
    char* feat[4];
    int feat_slot[4];
    for (int l = 0; l < 4; ++l) {
        int slot = 0;
        if (l > 0) {
            cur = ws + p.R[l][0];
            run.crush(h->crush[l - 1], feat[l - 1], cur, nb, p.hs[l - 1], p.ws[l - 1]);
        }
        for (auto& b : h->enc_blocks[l]) {
            char* nxt = ws + p.R[l][slot ^ 1];
            block(*b, cur, ws + p.HID[l], ws + p.Z[l], nxt, p.hs[l], p.ws[l]);
            cur = nxt;
            slot ^= 1;
        }
        feat[l] = cur;
        feat_slot[l] = slot;
    }

    // quality head (model.py:482, 1026-1032); upscale() discards it (model.py:175)
    if (out_qa) {
        const int F = h->cfg.num_deg_features;
        run.conv3(h->qa_conv, feat[3], ws + p.QA, nb, p.hs[3], p.ws[3], EPI_STORE, 0, 0, 0);
        if (run.rc) return run.rc;
        if (hipError_t e = launch_qa_reduce(h->dtype, ws + p.QA, h->qa_bias, out_qa, nb, p.hs[3] * p.ws[3], pad16(F), F, s);
            e != hipSuccess)
            return fail(MZ_ERR_HIP, "qa reduce launch: %s", hipGetErrorString(e));
    }

    // decoder (model.py:691-724)
    cur = feat[3];
    int slot = feat_slot[3];
    for (int d = 0; d < 4; ++d) {
        const int l = 3 - d;
        if (d > 0) {
            // SubpixelConv2d (model.py:926-930) + crop_feature_maps zero pad (model.py:650-689) + skip mix (:701)
            const ConvW& up = h->up[d - 1];
            char* u = ws + p.U[l];
            run.conv3(up, cur, u, nb, p.hs[l + 1], p.ws[l + 1], EPI_D2S, 0, p.hs[l], p.ws[l]);
            if (run.rc) return run.rc;
            if (hipError_t e = launch_zero_border(h->dtype, u, nb, p.hs[l], p.ws[l], up.cq_p, 2 * p.hs[l + 1], 2 * p.ws[l + 1], s);
                e != hipSuccess)
                return fail(MZ_ERR_HIP, "zero border launch: %s", hipGetErrorString(e));
            // pick a level-l buffer that is not the saved encoder feature
            slot = (feat_slot[l] + 1) % 3;
            char* dst = ws + p.R[l][slot];
            run.mix(h->skipmix[d - 1], h->skip_alpha[d - 1], feat[l], u, dst, nb, p.hs[l], p.ws[l]);
            cur = dst;
        }
        for (auto& b : h->dec_blocks[d]) {
            int nslot = (slot + 1) % 3;
            if (d > 0 && nslot == feat_slot[l]) nslot = (nslot + 1) % 3;  // (the encoder feature is dead after the skip mix, but keep it simple)
            char* nxt = ws + p.R[l][nslot];
            block(*b, cur, ws + p.HID[l], ws + p.Z[l], nxt, p.hs[l], p.ws[l]);
            cur = nxt;
            slot = nslot;
        }
    }

    // head (model.py:968-972, 997-1001) + bicubic skip + residual add + clamp (model.py:156,162,177)
    int hh = H, ww = W;
    for (int i = 0; i < h->nhead; ++i) {
        char *hid, *z, *y;
        if (i == 0) {
            hid = ws + p.HID[0]; z = ws + p.Z[0];
            int nslot = (slot + 1) % 3;
            y = ws + p.R[0][nslot];
        } else {
            hid = ws + p.HHID[i]; z = ws + p.HZ[i]; y = ws + p.HR[i][1];
        }
        block(*h->head_blocks[i], cur, hid, z, y, hh, ww);
        const bool last = i == h->nhead - 1;
        if (last) {
            run.conv3(*h->head_up[i], y, out_sr, nb, hh, ww, EPI_FINAL, 0, 2 * hh, 2 * ww, x, r, clamp);
        } else {
            char* nxt = ws + p.HR[i + 1][0];
            run.conv3(*h->head_up[i], y, nxt, nb, hh, ww, EPI_D2S, 0, 2 * hh, 2 * ww);
            cur = nxt;
            hh *= 2; ww *= 2;
        }
    }
    return run.rc;
}

static int forward_impl(mz_handle* h, const void* x, void* out_sr, float* out_qa, int B, int H, int W, int clamp,
                        void* workspace, size_t workspace_bytes, int max_images_in_flight, void* hip_stream, int io_u8) {
    if (!h || !x || !out_sr || !workspace) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    if (B <= 0 || H < 8 || W < 8) return fail(MZ_ERR_INVALID_ARGUMENT, "need B >= 1 and H, W >= 8 (got %d, %d, %d)", B, H, W);
    int rc = mz_weights_complete(h);
    if (rc) return rc;
    const int nbmax = default_micro_batch(h, B, H, W, max_images_in_flight);
    Plan p;
    make_plan(h, nbmax, H, W, p);
    if (workspace_bytes < p.total)
        return fail(MZ_ERR_WORKSPACE_TOO_SMALL, "workspace too small: %zu bytes given, %zu needed", workspace_bytes, p.total);
    const size_t sz = io_u8 ? 1 : dtype_size(h->dtype);
    const int r = h->cfg.upscale_ratio;
    const size_t in_img = (size_t)3 * H * W * sz;
    const size_t out_img = (size_t)3 * H * r * W * r * sz;
    for (int b0 = 0; b0 < B; b0 += nbmax) {
        const int nb = std::min(nbmax, B - b0);
        rc = forward_micro(h, (const char*)x + b0 * in_img, (char*)out_sr + b0 * out_img,
                           out_qa ? out_qa + (size_t)b0 * h->cfg.num_deg_features : nullptr, nb, H, W, clamp,
                           (char*)workspace, (hipStream_t)hip_stream, io_u8);
        if (rc) return rc;
    }
    return MZ_OK;
}

extern "C" int mz_forward(mz_handle* h, const void* x, void* out_sr, float* out_qa, int B, int H, int W, int clamp,
                          void* workspace, size_t workspace_bytes, int max_images_in_flight, void* hip_stream) {
    return forward_impl(h, x, out_sr, out_qa, B, H, W, clamp, workspace, workspace_bytes, max_images_in_flight, hip_stream, 0);
}

extern "C" int mz_forward_u8(mz_handle* h, const uint8_t* x, uint8_t* out_sr, float* out_qa, int B, int H, int W,
                             void* workspace, size_t workspace_bytes, int max_images_in_flight, void* hip_stream) {
    return forward_impl(h, x, out_sr, out_qa, B, H, W, /*clamp (implied by the uint8 store)*/ 1, workspace, workspace_bytes,
                        max_images_in_flight, hip_stream, 1);
}

// ------------------------------------------------------------------------------------------------
// operator-level entry points (tests)
// ------------------------------------------------------------------------------------------------
extern "C" int mz_padded_channels(int c) { return pad16(c); }

struct TempBuf {
    void* p = nullptr;
    ~TempBuf() {
        if (p) (void)hipFree(p);
    }
};

extern "C" int mz_op_conv(int dtype, int kind, const void* in0, const void* in1, const float* w_dev_f32, float alpha,
                          void* out, int B, int H, int W, int cin, int cout, int Hout, int Wout, int silu,
                          void* hip_stream) {
    int rc = ensure_device_ready();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    ConvW c;
    switch (kind) {
        case 0: plan_conv(c, dtype, MODE_CONV3, cout, cin, 3, 3, OUT_PLAIN, SRC_PLAIN, 0, 0); break;
        case 1: plan_conv(c, dtype, MODE_CONV3, cout, cin, 3, 3, OUT_D2S, SRC_PLAIN, 0, 0); break;
        case 2: plan_conv(c, dtype, MODE_GEMM1, cout, cin, 2, 2, OUT_PLAIN, SRC_CRUSH, 0, 0); break;
        case 3: plan_conv(c, dtype, MODE_GEMM1, cout, 2 * cout, 1, 1, OUT_PLAIN, SRC_CONCAT, cout, cout); break;
        default: return fail(MZ_ERR_INVALID_ARGUMENT, "bad op kind %d", kind);
    }
    TempBuf zero, packed, packed16, packed16r, packed16t;
    HIPCHK(hipMalloc(&zero.p, 4096));
    HIPCHK(hipMemsetAsync(zero.p, 0, 4096, s));
    rc = pack_conv(c, dtype, w_dev_f32, s);
    packed.p = c.packed;
    packed16.p = c.packed16;
    packed16r.p = c.packed16r;   // kind 3, C = 192: the second packing of the gate weights (mix16b_kernel)
    packed16t.p = c.packed16t;   // kind 0, one N tile of 33..48 channels: conv3t_kernel's packing
    if (rc) return rc;
    // a throw-away handle carries the zero page / staging choice for Runner
    mz_handle fake;
    fake.zero_page = zero.p;
    fake.knobs = read_knobs();
    Runner run{&fake, s, dtype};
    switch (kind) {
        case 0: run.conv3(c, in0, out, B, H, W, EPI_STORE, silu, 0, 0); break;
        case 1:
            run.conv3(c, in0, out, B, H, W, EPI_D2S, 0, Hout, Wout);
            if (!run.rc) {
                hipError_t e = launch_zero_border(dtype, out, B, Hout, Wout, c.cq_p, 2 * H, 2 * W, s);
                if (e != hipSuccess) run.rc = fail(MZ_ERR_HIP, "zero border: %s", hipGetErrorString(e));
            }
            break;
        case 2: run.crush(c, in0, out, B, H, W); break;
        case 3: run.mix(c, alpha, in0, in1, out, B, H, W); break;
    }
    fake.zero_page = nullptr;
    HIPCHK(hipStreamSynchronize(s));
    return run.rc;
}

// conv2 of a block + AdaptiveResidualMix with the block input in ONE launch (model.py:773-778 second half, 826-839): the fused
// kernels of the 16-bit modes (conv3r_kernel / conv3s_kernel / conv3w_kernel with FUSE), for C <= 96.
//   hid [B, cin, H, W] (conv1's activated output), x [B, cout, H, W] (the block input), w2 [cout, cin, 3, 3], wmix [cout, 2 cout, 1, 1]
extern "C" int mz_op_conv_mix(int dtype, const void* hid, const void* x, const float* w2_dev_f32, const float* wmix_dev_f32, float alpha,
                              void* out, int B, int H, int W, int cin, int cout, void* hip_stream) {
    int rc = ensure_device_ready();
    if (rc) return rc;
    if (!hid || !x || !w2_dev_f32 || !wmix_dev_f32 || !out) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    hipStream_t s = (hipStream_t)hip_stream;
    ConvW c2;
    plan_conv(c2, dtype, MODE_CONV3, cout, cin, 3, 3, OUT_PLAIN, SRC_PLAIN, 0, 0);
    if (!(c2.ntiles == 1 && c2.nt <= 3)) return fail(MZ_ERR_INVALID_ARGUMENT, "the fused conv2 + mix needs all output channels in one N tile (cout <= 96)");
    ConvW f;  // as plan_model()'s BlockW::mixf
    f.cout = cout; f.cin = 2 * cout; f.kh = f.kw = 1;
    f.mode = MODE_GEMM1; f.taps = 1;
    f.nt = c2.nt; f.ntiles = 1;
    f.out_map = OUT_PLAIN; f.in_map = SRC_MIXF;
    f.c0 = cout; f.cp0 = pad16(cout); f.c1 = cout;
    const int zg = dtype == DT_F32 ? 4 : 2;
    f.nchunks = f.nchunks_real = f.cp0 / chunk_channels(dtype) + f.nt * zg;
    f.packed_sz = packed_bytes(1, f.nt, 1, f.nchunks);
    if (dtype != DT_F32) {
        f.nchunks16 = (f.cp0 + 31) / 32 + f.nt;
        f.packed16_sz = packed_bytes(1, 2 * f.nt, 1, f.nchunks16);
        if (f.cp0 == 48) f.packed16t_sz = packed_bytes(1, 3, 1, 3);
    }
    TempBuf zero, p0, p1, p2, p3, p4, p5, p6;
    HIPCHK(hipMalloc(&zero.p, 4096));
    HIPCHK(hipMemsetAsync(zero.p, 0, 4096, s));
    rc = pack_conv(c2, dtype, w2_dev_f32, s);
    p0.p = c2.packed; p1.p = c2.packed16; p5.p = c2.packed16t;
    if (rc) return rc;
    rc = pack_conv(f, dtype, wmix_dev_f32, s);
    p2.p = f.packed; p3.p = f.packed16; p4.p = f.packed16r; p6.p = f.packed16t;
    if (rc) return rc;
    mz_handle fake;
    fake.zero_page = zero.p;
    fake.knobs = read_knobs();
    Runner run{&fake, s, dtype};
    run.conv3(c2, hid, out, B, H, W, EPI_FUSEDMIX, 0, 0, 0, nullptr, 0, 0, nullptr, &f, x, alpha);
    fake.zero_page = nullptr;
    HIPCHK(hipStreamSynchronize(s));
    return run.rc;
}

// a17: conv3x3 -> gamma[b, c] * y + beta[b, c] -> optional SiLU, the per-channel modulation of a FiLM /
// control module.  The reference snapshot has no such module (README.md:86-129 describes library version 0.2.x): nothing to
// be parity-checked against, so this operator is checked against the build's own CPU restatement only ("parity unpinned").
extern "C" int mz_op_conv_film(int dtype, const void* in0, const float* w_dev_f32, const float* gamma_dev_f32,
                               const float* beta_dev_f32, void* out, int B, int H, int W, int cin, int cout, int silu,
                               void* hip_stream) {
    int rc = ensure_device_ready();
    if (rc) return rc;
    if (!in0 || !w_dev_f32 || !gamma_dev_f32 || !beta_dev_f32 || !out) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    if (dtype != DT_BF16 && dtype != DT_F16) return fail(MZ_ERR_INVALID_ARGUMENT, "the FiLM epilogue is implemented for bf16 / fp16");
    hipStream_t s = (hipStream_t)hip_stream;
    ConvW c;
    plan_conv(c, dtype, MODE_CONV3, cout, cin, 3, 3, OUT_PLAIN, SRC_PLAIN, 0, 0);
    TempBuf zero, packed, packed16, packed16t, gpad, bpad;
    HIPCHK(hipMalloc(&zero.p, 4096));
    HIPCHK(hipMemsetAsync(zero.p, 0, 4096, s));
    rc = pack_conv(c, dtype, w_dev_f32, s);
    packed.p = c.packed;
    packed16.p = c.packed16;
    packed16t.p = c.packed16t;
    if (rc) return rc;
    // gamma / beta [B][cout] -> [B][padded cout], pad channels zero
    const int cp = pad16(cout);
    HIPCHK(hipMalloc(&gpad.p, sizeof(float) * (size_t)B * cp));
    HIPCHK(hipMalloc(&bpad.p, sizeof(float) * (size_t)B * cp));
    HIPCHK(hipMemsetAsync(gpad.p, 0, sizeof(float) * (size_t)B * cp, s));
    HIPCHK(hipMemsetAsync(bpad.p, 0, sizeof(float) * (size_t)B * cp, s));
    HIPCHK(hipMemcpy2DAsync(gpad.p, sizeof(float) * cp, gamma_dev_f32, sizeof(float) * cout, sizeof(float) * cout, B, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpy2DAsync(bpad.p, sizeof(float) * cp, beta_dev_f32, sizeof(float) * cout, sizeof(float) * cout, B, hipMemcpyDeviceToDevice, s));
    mz_handle fake;
    fake.zero_page = zero.p;
    fake.knobs = read_knobs();
    Runner run{&fake, s, dtype};
    run.film_gamma = (const float*)gpad.p;
    run.film_beta = (const float*)bpad.p;
    run.conv3(c, in0, out, B, H, W, EPI_STORE, silu, 0, 0);
    fake.zero_page = nullptr;
    HIPCHK(hipStreamSynchronize(s));
    return run.rc;
}

extern "C" int mz_op_stem(int dtype, const void* x, const float* w_dev_f32, const float* b_dev_f32, void* out, int B,
                          int H, int W, int cout, void* hip_stream) {
    int rc = ensure_device_ready();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    const int cp = pad16(cout);
    TempBuf w4;
    HIPCHK(hipMalloc(&w4.p, sizeof(float) * 4 * cp));
    HIPCHK(hipMemsetAsync(w4.p, 0, sizeof(float) * 4 * cp, s));
    HIPCHK(launch_pack_stem(w_dev_f32, b_dev_f32, (float*)w4.p, cout, cp, s));
    HIPCHK(launch_stem(dtype, x, (const float*)w4.p, out, B, H, W, cp, s));
    HIPCHK(hipStreamSynchronize(s));
    return MZ_OK;
}

extern "C" int mz_op_final(int dtype, const void* feat, const void* img, const float* w_dev_f32, void* out, int B, int H,
                           int W, int cin, int R, int clamp, void* hip_stream) {
    int rc = ensure_device_ready();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    ConvW c;
    plan_conv(c, dtype, MODE_CONV3, 12, cin, 3, 3, OUT_FINAL, SRC_PLAIN, 0, 0);
    TempBuf zero, packed, packed16;
    HIPCHK(hipMalloc(&zero.p, 4096));
    HIPCHK(hipMemsetAsync(zero.p, 0, 4096, s));
    rc = pack_conv(c, dtype, w_dev_f32, s);
    packed.p = c.packed;
    packed16.p = c.packed16;
    if (rc) return rc;
    mz_handle fake;
    fake.zero_page = zero.p;
    fake.knobs = read_knobs();
    Runner run{&fake, s, dtype};
    run.conv3(c, feat, out, B, H, W, EPI_FINAL, 0, 2 * H, 2 * W, img, R, clamp);
    fake.zero_page = nullptr;
    HIPCHK(hipStreamSynchronize(s));
    return run.rc;
}

// ------------------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------------------
extern "C" const char* mz_last_error(void) { return g_err; }
extern "C" const char* mz_debug_last_kernel(void) { return g_last_kernel; }

// Host-only (no GPU): the tile list Runner::tile_table() uploads for a launch of B images of tiles_y x tiles_x tiles of th x tw pixels,
// ntiles N tiles, walked in groups of gm x gn (blk4: block rows of four tile rows).  Writes at most cap entries of two words to out
// and returns the number of tiles listed (tests/test_cabi_cpu.py checks that every tile appears exactly once).
extern "C" int mz_debug_tile_list(int B, int tiles_y, int tiles_x, int ntiles, int gm, int gn, int blk4, int th, int tw, unsigned int* out, int cap) {
    if (B < 1 || tiles_x < 1 || tiles_y < 1 || ntiles < 1 || gm < 1 || gn < 1 || th < 1 || tw < 1 || !out || cap < 0) return MZ_ERR_INVALID_ARGUMENT;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.mtiles = B * tiles_x * tiles_y; a.ntiles = ntiles;
    a.gm = gm < a.mtiles ? gm : a.mtiles; a.gn = gn < ntiles ? gn : ntiles; a.blk4 = blk4 ? 1 : 0;
    a.groups_m = (a.mtiles + a.gm - 1) / a.gm;
    a.grid = a.groups_m * ((ntiles + a.gn - 1) / a.gn) * a.gm * a.gn;
    std::vector<uint32_t> t;
    tile_list(a, th, tw, t);
    const int n = (int)(t.size() / 2);
    for (int i = 0; i < n && i < cap; ++i) { out[2 * i] = t[2 * i]; out[2 * i + 1] = t[2 * i + 1]; }
    return n;
}
extern "C" const char* mz_version(void) { return "mewzoom_hip 0.1 (gfx950)"; }

extern "C" double mz_flops_per_image(const mz_handle* h, int H, int W) {
    if (!h) return 0.0;
    const int hr = h->cfg.hidden_ratio;
    int hs[4] = {H, 0, 0, 0}, ws[4] = {W, 0, 0, 0};
    for (int i = 1; i < 4; ++i) { hs[i] = hs[i - 1] / 2; ws[i] = ws[i - 1] / 2; }
    double macs = 3.0 * h->ch[0] * H * W;
    auto blk = [&](double c) { return (18.0 * hr + 2.0) * c * c; };
    for (int l = 0; l < 4; ++l) macs += (h->enc[l] + h->dec[l]) * blk(h->ch[l]) * hs[l] * ws[l];
    for (int l = 0; l < 3; ++l) {
        macs += 4.0 * h->ch[l] * h->ch[l + 1] * hs[l + 1] * ws[l + 1];
        macs += 9.0 * h->ch[l + 1] * 4.0 * h->ch[l] * hs[l + 1] * ws[l + 1];
        macs += 2.0 * h->ch[l] * h->ch[l] * hs[l] * ws[l];
    }
    macs += 9.0 * h->ch[3] * h->cfg.num_deg_features * hs[3] * ws[3];
    double hh = H, ww = W;
    for (int i = 0; i < h->nhead; ++i) {
        const double cout = (i == h->nhead - 1) ? 3 : h->ch[0];
        macs += blk(h->ch[0]) * hh * ww + 9.0 * h->ch[0] * 4.0 * cout * hh * ww;
        hh *= 2; ww *= 2;
    }
    return 2.0 * macs;
}

extern "C" int mz_profile_enable(mz_handle* h, int on) {
    if (!h) return fail(MZ_ERR_INVALID_ARGUMENT, "null handle");
    h->prof = on != 0;
    h->recs_used = 0;
    return MZ_OK;
}

extern "C" int mz_profile_dump(mz_handle* h, const char* path) {
    // One CSV row per profiled launch since the last reset (does not reset).
    if (!h || !path) return fail(MZ_ERR_INVALID_ARGUMENT, "null argument");
    FILE* f = fopen(path, "w");
    if (!f) return fail(MZ_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    fprintf(f, "kind,B,H,W,cin,cout,nt,ntiles,mtiles,n_fast,ms,gflop,tflops,alg_GBps\n");
    for (size_t i = 0; i < h->recs_used; ++i) {
        ProfRec& r = h->recs[i];
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        const char* kn = r.kind == 0 ? "conv3" : (r.kind == 1 ? "mix" : "crush");
        fprintf(f, "%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.3f,%.1f,%.1f\n", kn, r.B, r.H, r.W, r.cin, r.cout, r.nt, r.ntiles,
                r.mtiles, r.n_fast, ms, r.flops / 1e9, r.flops / (ms * 1e-3) / 1e12, r.bytes / (ms * 1e-3) / 1e9);
    }
    fclose(f);
    return MZ_OK;
}

extern "C" int mz_profile_read(mz_handle* h, double* conv_ms, double* conv_flops, double* conv_launches, double* other_ms,
                               double* conv_bytes) {
    if (!h) return fail(MZ_ERR_INVALID_ARGUMENT, "null handle");
    double cm = 0, cf = 0, cl = 0, om = 0, cb = 0;
    for (size_t i = 0; i < h->recs_used; ++i) {
        ProfRec& r = h->recs[i];
        HIPCHK(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.is_conv3) { cm += ms; cf += r.flops; cl += 1; cb += r.bytes; }
        else om += ms;
    }
    if (conv_ms) *conv_ms = cm;
    if (conv_flops) *conv_flops = cf;
    if (conv_launches) *conv_launches = cl;
    if (other_ms) *other_ms = om;
    if (conv_bytes) *conv_bytes = cb;
    h->recs_used = 0;
    return MZ_OK;
}
