// conv3q_kernel: 3x3 convolution (pad 1, stride 1) on v_mfma_f32_16x16x32_{bf16,f16} with ONE compute wave per SIMD.
//
// Why another generation: in conv3s_kernel (two 64 px x 96 ch compute waves per SIMD, 168 VGPRs) the K
// loop pays for its fragment reads (10 x 1 KB per 24 MFMAs) and a 64-pixel-wide tile wastes 6.7 % of the MFMAs of the
// 240-pixel-wide level.  Here
//   * a workgroup is 8 waves of 256 registers: on every SIMD ONE compute wave and ONE loader wave;
//   * compute wave tile = 96 pixels x 96 channels = 6 pixel fragments x 6 channel fragments (144 accumulator registers): 12
//     fragment reads per 36 MFMAs (0.33 per MFMA instead of 0.42);
//   * the pixel tile is 8 rows x 48 columns (wave w = rows 2w, 2w+1): 1920, 960, 480 and 240 are multiples of 48;
//   * the four loader waves issue all LDS-DMA (halo image by buffer_load ... lds with hardware zero fill outside the image,
//     weights by global_load ... lds), a quarter of the pieces each.  (A first cut without loader waves -- 4 waves of 512
//     registers issuing the DMA between their own MFMAs -- lost 25 % to the DMA issue: a wave's issue is in order, and an
//     LDS-DMA instruction holds it for 60 - 100 cycles.  The VMEM issue of a DIFFERENT wave does not hold the MFMA issue.)
//   * the weight half-chunks turn on THREE LDS slots and the halo image of chunk u + 1 is published at the middle of chunk
//     u, so that every fragment the next half-step starts with has been readable for a whole half-step: the fragment
//     pipeline (weights two 12-MFMA groups ahead, pixel fragments one tap ahead) runs straight through the barriers,
//     across chunk and tile boundaries; a barrier only publishes loads and guards slot reuse.
//
// LDS map (NT = 3): [halo 0 | halo 1] 2 x 32 KB (4 planes x 512 entries, 10 x 50 halo pixels) + 3 weight half-chunk slots of
// 28 KB = 148 KB.  Weights use the packing of conv3s_kernel (pack_kernel with frag16 = 1): group G = (tap G / NT, channel
// fragment pair G % NT) = pieces 2G, 2G + 1 of the chunk.
//
// Half-step protocol (h = global half-step counter, u = h / 2 = chunk counter of the workgroup; barrier B_h opens half-step h):
//   loader, after B_h : weights(h + 2) -> slot (h + 2) % 3 (all waves are past half-step h - 1, whose slot this was);
//                       h even: halo(u + 1) -> halo slot (u + 1) & 1 (all waves are past chunk u - 1);
//                       s_waitcnt vmcnt(0); B_{h+1}
//   compute, after B_h: groups of half-step h; the fragment requests of its last groups already address weights(h + 1)
//                       (published by B_h) and, in the second half of a chunk, tap 0 of halo(u + 1) (published by B_{2u+1})
#include "mz_device.h"

namespace mz {

namespace q3 {

constexpr int TH = 8, TW = 48;
constexpr int ROWW = 50;
constexpr int NPIX = 10 * ROWW;      // 500 halo pixels
constexpr int PLANE_ENT = 512;       // padded: 4 planes = 32 whole DMA instructions
constexpr int A_PLANE = PLANE_ENT * 16;
constexpr int A_SLOT = 4 * A_PLANE;  // 32 KB
constexpr int NPF = 6;               // pixel fragments per wave: 2 rows x 3
constexpr int NT = 3, NF = 6, BN = 96;
constexpr int NG = 9 * NT;           // groups per 32-channel chunk
constexpr int G0 = (NG + 1) / 2;     // groups in the first half (14)
constexpr int P0 = 2 * G0, P1 = 2 * (NG - G0);  // 1-KiB pieces of the two halves (28, 26)
constexpr int B_SLOT = P0 * 1024;
constexpr int B_BASE = 2 * A_SLOT;
constexpr int LDS_BYTES = B_BASE + 3 * B_SLOT;  // 151552

// byte offset of pixel fragment pf of tap (dy, dx) inside one plane of the halo image, relative to the wave's first row
template <int TAP, int PF> constexpr int a_off() {
    constexpr int DY = TAP / 3, DX = TAP % 3;
    return ((DY + PF / 3) * ROWW + DX + 16 * (PF % 3)) * 16;
}

struct Frag {
    u32x4 x[2][NPF];  // [tap parity][pixel fragment]; a chunk has 9 taps: see end_of_chunk()
    u32x4 w[3][2];    // [group % 3][channel fragment of the pair]: requested TWO groups (24 MFMAs) ahead; 27 groups per chunk
                      // = 0 mod 3, so the rotation is the same in every chunk
};

// lgkmcnt(N): everything but the N youngest LDS reads has landed (LDS returns in order); the registers named "+v" are
// the ones the following MFMAs use
template <int N> __device__ __forceinline__ void wait_w(u32x4& w0, u32x4& w1) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w0), "+v"(w1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_wx(u32x4& w0, u32x4& w1, u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3, u32x4& x4, u32x4& x5) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(w0), "+v"(w1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5)
                 : "n"(N)
                 : "memory");
}

// LDS read addresses of one lane: halo image of the current / next chunk, weight half-slot of the current / next half
struct Bases {
    uint32_t a_cur, a_nxt, b_cur, b_nxt;
};

// Group GC of a chunk: tap t = GC / 3, channel-fragment pair n = GC % 3: 12 MFMAs.  While they issue, the wave requests the
// weight pair of group GC + 2 and (n < 2) three pixel fragments of tap t + 1 -- from the NEXT half-slot / halo image where the
// group or tap index runs past this half / chunk.  Tap t reads pixel buffer (t + XP) & 1: a chunk has 9 taps, so the parity XP
// of a chunk's tap 0 flips from chunk to chunk; the chunk loop is unrolled by two (XP = 0, then XP = 1) and the kernel takes
// EVEN chunk counts only (the host guards; an odd tail behind a branch made hipcc spill 8 - 12 VGPRs, a register copy at the
// end of every chunk instead of the unroll cost 2 %).
template <class TT, int GC, int XP, bool ZERO_C, int M>
__device__ __forceinline__ void group_mfmas(f32x4 (&acc)[NPF][NF], Frag& f, const Bases& bs) {
    if constexpr (M < 12) {
        constexpr int t = GC / 3, n = GC % 3, xp = (t + XP) & 1, xq = xp ^ 1;
        constexpr int H = GC < G0 ? 0 : 1;
        constexpr int Gs = H == 0 ? 0 : G0, Ge = H == 0 ? G0 : NG;
        constexpr int T = GC + 2;      // group whose weights are requested now
        constexpr bool t_here = T < Ge;
        constexpr int t_idx = t_here ? T - Gs : T - Ge;  // its index inside its half
        constexpr int k = M / 6, pf = M % 6;
        if constexpr (ZERO_C) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 r;
            if constexpr (TT::IS_BF16)
                r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f.w[GC % 3][k]), __builtin_bit_cast(bf16x8_t, f.x[xp][pf]), zero, 0, 0, 0);
            else
                r = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, f.w[GC % 3][k]), __builtin_bit_cast(f16x8_t, f.x[xp][pf]), zero, 0, 0, 0);
            acc[pf][2 * n + k] = r;
        } else {
            mma16<TT>(acc[pf][2 * n + k], f.w[GC % 3][k], f.x[xp][pf]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M < 2) {
            f.w[T % 3][M] = lds_read128<(2 * t_idx + M) * 1024>(t_here ? bs.b_cur : bs.b_nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n < 2 && M >= 2 && M < 5) {
            constexpr int pfn = 3 * n + (M - 2);
            if constexpr (t + 1 < 9) f.x[xq][pfn] = lds_read128<a_off<t + 1, pfn>()>(bs.a_cur);
            else f.x[xq][pfn] = lds_read128<a_off<0, pfn>()>(bs.a_nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        group_mfmas<TT, GC, XP, ZERO_C, M + 1>(acc, f, bs);
    }
}

// groups [G, GE) of one half-step; ZERO_T: the chunk is the first of a tile (tap 0 writes the accumulators instead of adding)
template <class TT, int G, int GE, int XP, bool ZERO_T>
__device__ __forceinline__ void groups(f32x4 (&acc)[NPF][NF], Frag& f, const Bases& bs) {
    if constexpr (G < GE) {
        constexpr int t = G / 3, n = G % 3;
        __builtin_amdgcn_sched_barrier(0);
        group_mfmas<TT, G, XP, (ZERO_T && G < 3), 0>(acc, f, bs);
        // what the NEXT group needs (also across the end of this half-step: the stream continues behind the barrier).
        // LDS reads return in order.  Reads requested per group: n = 0, 1: two weight + three pixel fragments, n = 2: two weight
        // fragments.  The next group's weight pair was requested first thing in the PREVIOUS group.
        constexpr int wn = (G + 1) % 3, xn = (t + 1 + XP) & 1;
        if constexpr (n == 2)  // a new tap starts: its six pixel fragments (requested in this tap's groups 0 and 1) must be in
            wait_wx<2>(f.w[wn][0], f.w[wn][1], f.x[xn][0], f.x[xn][1], f.x[xn][2], f.x[xn][3], f.x[xn][4], f.x[xn][5]);
        else if constexpr (n == 1)  // younger than the pair: the previous group's three pixel fragments + this group's five
            wait_w<8>(f.w[wn][0], f.w[wn][1]);
        else                        // the previous group (n = 2) requested the pair only: this group's five may be outstanding
            wait_w<5>(f.w[wn][0], f.w[wn][1]);
        groups<TT, G + 1, GE, XP, ZERO_T>(acc, f, bs);
    }
}
// accumulators of one pixel fragment -> plane-major tensor (entry16(), mz_device.h).  EPI_STORE: the lane's entry of channel-fragment
// pair n lies at plane (nbase / 8 + 4 n + lane_cu), lane_cu = 2 (g & 1) + (g >> 1): one 64-bit base per pixel fragment, then a
// uniform stride of four planes per pair.
template <class TT, int EPI, bool SILU>
__device__ __forceinline__ void store_pf(const ConvArgs& a, f32x4 (&accpf)[NF], int lane, int nbase, int b, int py, int px) {
    constexpr bool d2s = EPI == EPI_D2S;
    const int g = lane >> 4;
    const long long plane_o = d2s ? (long long)a.Hout * a.Wout * 16 : (long long)a.H * a.W * 16;
    char* const obase = (char*)a.out + (long long)b * a.p_out * plane_o;
    const bool inside = py < a.H && px < a.W;
    const int lane_cu = 2 * (g & 1) + (g >> 1);
    if constexpr (!d2s) {
        char* dst = obase + (long long)((nbase >> 3) + lane_cu) * plane_o + ((long long)py * a.W + px) * 16;
        const long long stride = 4 * plane_o;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const u32x4 o = entry16<TT, SILU>(accpf[2 * n], accpf[2 * n + 1]);
            const int nch = nbase + (4 * n + lane_cu) * 8;
            if (inside && nch < a.cp_out) *(u32x4*)dst = o;
            dst += stride;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const u32x4 o = entry16<TT, SILU>(accpf[2 * n], accpf[2 * n + 1]);
            const int nch = nbase + (4 * n + lane_cu) * 8;
            if (!inside || nch >= 4 * a.cp_out) continue;
            const int ij = nch / a.cp_out;
            const int ch = nch - ij * a.cp_out;
            const int Y = 2 * py + (ij >> 1), X = 2 * px + (ij & 1);
            *(u32x4*)(obase + (ch >> 3) * plane_o + ((long long)Y * a.Wout + X) * 16) = o;
        }
    }
}

}  // namespace q3

template <class TT>
__global__ __launch_bounds__(512) void conv3q_kernel(const ConvArgs a) {
    using namespace q3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3 compute, 4..7 loaders
    const int nchunks = a.nchunks16;                          // 32-channel chunks; even (launch_conv3q refuses odd counts)

    // ---- tile walk (as conv3s_kernel: an XCD's contiguous id range, strided by the workgroups of that XCD) ----
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3, step = gridDim.x >> 3;
    const int q = a.grid >> 3, rem = a.grid & 7;
    const int cnt = q + (xcd < rem ? 1 : 0);
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    auto seek = [&](int i, int& mt, int& nt) __attribute__((always_inline)) {
        while (i < cnt && !tile_of(a, base + i, mt, nt)) i += step;
        return i;
    };
    int mtile = 0, ntile = 0;
    int cur = seek(pos, mtile, ntile);
    if (cur >= cnt) return;  // uniform over the workgroup

    const int tpi = a.tiles_x * a.tiles_y;
    auto tile_origin = [&](int mt, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        b = fdiv(mt, tpi, a.inv_tpi);
        const int trem = mt - b * tpi;
        int tyi, txi;
        tile_rc(a, trem, tyi, txi);
        y0 = tyi * TH;
        x0 = txi * TW;
    };

    if (w >= 4) {
        // =========================== loader waves ===========================
        const int lw = w - 4;
        // halo DMA addressing: this wave's pieces j = lw + 4 i cover halo entries [64 j, 64 j + 64).  roff = byte offset of the
        // entry relative to the tile's halo origin (y0 - 1, x0 - 1) inside plane 0 of the chunk; pyx = its (row, column)
        const long long plane_in = (long long)a.H * a.W * 16;
        uint32_t roff[8], pyx[8], aoff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 64 * (lw + 4 * i) + lane;
            const int plane = e >> 9, p = e & 511;
            const int py = p / ROWW, px = p - py * ROWW;
            roff[i] = (((uint32_t)plane * (uint32_t)a.H + (uint32_t)py) * (uint32_t)a.W + (uint32_t)px) * 16u;
            pyx[i] = p < NPIX ? ((uint32_t)py << 16) | (uint32_t)px : 0xffffffffu;
        }
        const char* img_l = nullptr;  // image of the tile being loaded
        auto set_load_tile = [&](int mt) __attribute__((always_inline)) {
            int b, y0, x0;
            tile_origin(mt, b, y0, x0);
            img_l = (const char*)a.in0 + (long long)b * a.p0 * plane_in;
            const uint32_t delta = ((uint32_t)(y0 - 1) * (uint32_t)a.W + (uint32_t)(x0 - 1)) * 16u;  // mod 2^32
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gy = y0 - 1 + (int)(pyx[i] >> 16), gx = x0 - 1 + (int)(pyx[i] & 0xffff);
                const bool ok = pyx[i] != 0xffffffffu && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                aoff[i] = ok ? roff[i] + delta : 0xffffffffu;  // beyond the descriptor: the hardware returns zeros
            }
        };
        // load cursor: walks (tile, chunk) in the workgroup's order
        const size_t chunk_bytes = (size_t)(P0 + P1) * 1024;
        int l_pos = cur, l_mt = mtile, l_nt = ntile, l_kc = 0;
        bool l_ok = true;
        const char* l_w = (const char*)a.wpk16 + (size_t)l_nt * nchunks * chunk_bytes + lane * 16;
        set_load_tile(l_mt);
        auto issue_halo = [&](char* dst) __attribute__((always_inline)) {
            if (!l_ok) return;
            const int planes = a.p0 - 4 * l_kc < 4 ? a.p0 - 4 * l_kc : 4;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(img_l + 4LL * l_kc * plane_in), 0, (int)(uint32_t)(planes * plane_in), 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (lw + 4 * i) * 1024), 16,
                                                         (int)aoff[i], 0, 0, 0);
        };
        auto issue_weights = [&](int half, char* dst) __attribute__((always_inline)) {
            if (!l_ok) return;
            const char* src = l_w + (size_t)l_kc * chunk_bytes + (half ? P0 * 1024 : 0);
            const int pieces = half ? P1 : P0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int j = lw + 4 * i;
                if (j < pieces) glds16(src + j * 1024, dst + j * 1024);
            }
        };
        auto advance_load = [&]() __attribute__((always_inline)) {  // after both halves of chunk l_kc have been issued
            if (++l_kc == nchunks) {
                l_kc = 0;
                l_pos = seek(l_pos + step, l_mt, l_nt);
                l_ok = l_pos < cnt;
                if (l_ok) {
                    l_w = (const char*)a.wpk16 + (size_t)l_nt * nchunks * chunk_bytes + lane * 16;
                    set_load_tile(l_mt);
                }
            }
        };
        int nhalf = 0;  // half-steps of this workgroup = 2 * nchunks * (its tiles)
        {
            int mt_, nt_;
            for (int i = cur; i < cnt; i = seek(i + step, mt_, nt_)) nhalf += 2 * nchunks;
        }
        char* hal_nxt = smem + A_SLOT;
        char* hal_cur = smem;
        char* ws_cur = smem + B_BASE;
        char* ws_nxt = smem + B_BASE + B_SLOT;
        char* ws_nn = smem + B_BASE + 2 * B_SLOT;
        // prologue: chunk 0 of the first tile (halo + both weight halves), published by B_0
        issue_halo(hal_cur);
        issue_weights(0, ws_cur);
        issue_weights(1, ws_nxt);
        advance_load();
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int h = 0; h < nhalf; h += 2) {
            // first half of a chunk: the NEXT chunk's halo image and first weight half
            issue_halo(hal_nxt);
            issue_weights(0, ws_nn);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            { char* t_ = ws_cur; ws_cur = ws_nxt; ws_nxt = ws_nn; ws_nn = t_; }
            // second half: the next chunk's second weight half
            issue_weights(1, ws_nn);
            advance_load();
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            { char* t_ = ws_cur; ws_cur = ws_nxt; ws_nxt = ws_nn; ws_nn = t_; }
            { char* t_ = hal_cur; hal_cur = hal_nxt; hal_nxt = t_; }
        }
        return;
    }

    // =========================== compute waves ===========================
    __builtin_amdgcn_s_barrier();  // B_0
    const int g = lane >> 4, c = lane & 15;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = g * A_PLANE + ((2 * w) * ROWW + c) * 16;
    Bases bs;
    bs.a_cur = lds_base + a_lane;
    bs.a_nxt = lds_base + A_SLOT + a_lane;
    uint32_t b0 = lds_base + B_BASE + lane * 16, b1 = b0 + B_SLOT, b2 = b1 + B_SLOT;  // read bases of the three weight slots
    bs.b_cur = b0;
    bs.b_nxt = b1;

    // prime the fragment stream: tap 0 of chunk 0 and the weight pairs of groups 0 and 1
    Frag f;
    f.x[0][0] = lds_read128<a_off<0, 0>()>(bs.a_cur);
    f.x[0][1] = lds_read128<a_off<0, 1>()>(bs.a_cur);
    f.x[0][2] = lds_read128<a_off<0, 2>()>(bs.a_cur);
    f.x[0][3] = lds_read128<a_off<0, 3>()>(bs.a_cur);
    f.x[0][4] = lds_read128<a_off<0, 4>()>(bs.a_cur);
    f.x[0][5] = lds_read128<a_off<0, 5>()>(bs.a_cur);
    f.w[0][0] = lds_read128<0 * 1024>(bs.b_cur);
    f.w[0][1] = lds_read128<1 * 1024>(bs.b_cur);
    f.w[1][0] = lds_read128<2 * 1024>(bs.b_cur);
    f.w[1][1] = lds_read128<3 * 1024>(bs.b_cur);
    wait_wx<0>(f.w[0][0], f.w[0][1], f.x[0][0], f.x[0][1], f.x[0][2], f.x[0][3], f.x[0][4], f.x[0][5]);
    wait_w<0>(f.w[1][0], f.w[1][1]);

    f32x4 acc[NPF][NF];
    auto half_tail = [&]() __attribute__((always_inline)) {  // B_{h+1}; rotate the weight half-slots: cur <- nxt <- nn <- cur
        __builtin_amdgcn_s_barrier();
        const uint32_t u_ = b0; b0 = b1; b1 = b2; b2 = u_;
        bs.b_cur = b0; bs.b_nxt = b1;
    };
    auto chunk_tail = [&]() __attribute__((always_inline)) {
        half_tail();
        const uint32_t v_ = bs.a_cur; bs.a_cur = bs.a_nxt; bs.a_nxt = v_;
    };
    while (cur < cnt) {
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (a peeled first chunk whose tap 0 WRITES the accumulators would save these 144 moves, but hipcc then spills 45
        // VGPRs at the junction of the peeled body and the loop: 216 values are live across it and only 256 registers exist)
        for (int kc = 0; kc + 1 < nchunks; kc += 2) {
            groups<TT, 0, G0, 0, false>(acc, f, bs);
            half_tail();
            groups<TT, G0, NG, 0, false>(acc, f, bs);
            chunk_tail();
            groups<TT, 0, G0, 1, false>(acc, f, bs);
            half_tail();
            groups<TT, G0, NG, 1, false>(acc, f, bs);
            chunk_tail();
        }
        // ---- epilogue: this wave's 96 pixels x 96 channels ----
        int b, y0, x0;
        tile_origin(mtile, b, y0, x0);
        const int nbase = ntile * BN;
        if (a.epi == EPI_D2S) {
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) store_pf<TT, EPI_D2S, false>(a, acc[pf], lane, nbase, b, y0 + 2 * w + pf / 3, x0 + 16 * (pf % 3) + c);
        } else if (a.silu) {
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) store_pf<TT, EPI_STORE, true>(a, acc[pf], lane, nbase, b, y0 + 2 * w + pf / 3, x0 + 16 * (pf % 3) + c);
        } else {
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf) store_pf<TT, EPI_STORE, false>(a, acc[pf], lane, nbase, b, y0 + 2 * w + pf / 3, x0 + 16 * (pf % 3) + c);
        }
        cur = seek(cur + step, mtile, ntile);
    }
}

size_t conv3q_lds_bytes() { return q3::LDS_BYTES; }

hipError_t init_conv3q() {
    hipError_t e = hipFuncSetAttribute((const void*)conv3q_kernel<TBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, q3::LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)conv3q_kernel<TF16>, hipFuncAttributeMaxDynamicSharedMemorySize, q3::LDS_BYTES);
}

// a.persist workgroups of 512 threads; a.tiles_x / tiles_y / mtiles describe 8 x 48 tiles; NT must be 3
hipError_t launch_conv3q(int dtype, const ConvArgs& a, hipStream_t s) {
    if (a.persist <= 0 || (a.persist & 7) || (a.nchunks16 & 1)) return hipErrorInvalidValue;
    switch (dtype) {
        case DT_BF16: hipLaunchKernelGGL(conv3q_kernel<TBF16>, dim3(a.persist), dim3(512), q3::LDS_BYTES, s, a); break;
        case DT_F16: hipLaunchKernelGGL(conv3q_kernel<TF16>, dim3(a.persist), dim3(512), q3::LDS_BYTES, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mz
