// conv3t_kernel: 3x3 convolution (pad 1, stride 1) for N tiles of 48 output channels on v_mfma_f32_16x16x32_{bf16,f16} -- the
// level-1 block of the 48-channel models (BASELINE configs[0] / [1]: conv2 96 -> 48 + AdaptiveResidualMix, reference
// model.py:746-748, 773-778, 826-839), which conv3s_kernel<NT = 2> computed with N padded 48 -> 64 and its epilogue exposed.
//
// The structure is conv3r_kernel's (mz_conv3r.h: two teams of four waves that alternate between the compute role and the
// loader + epilogue role from tile to tile), re-shaped around THREE channel fragments:
//   * wave tile 12 pixel fragments x 3 channel fragments = 36 accumulators (144 registers, as conv3r's 6 x 6): 192 pixels x 48
//     channels per wave; workgroup tile 12 rows x 64 columns (wave w: rows 3 w .. 3 w + 2); 1080, 540, 2160 are multiples of 12 and
//     1920, 960, 3840 of 64: no padded pixel on the 16:9 sizes.
//   * a tap is two GROUPS of 18 MFMAs (6 pixel fragments x 3 channel fragments).  Fragment stream: the six pixel fragments of the
//     next group are requested during the first six MFMAs of the current one (two buffers, flipping per group: 18 groups per chunk,
//     so every chunk starts on buffer 0), the three weight fragments of the next tap during the tap's first group (three buffers,
//     tap % 3: 9 taps per chunk, so no parity to carry either).  Same summation order as conv3s / conv3r (chunk, tap, one 32-channel
//     MFMA): bit-identical sums.
//   * weights: 27 KB per 32-channel chunk, three segments of three taps = three barriers per chunk.  With three steps per chunk and
//     three slots, segment s ALWAYS lives in slot s: the chunk's 27 pieces have fixed LDS addresses, nothing rotates.
//   * LDS: halo image 14 x 66 pixels x 4 planes (15 DMA pieces per plane, 60 KB) double-buffered + 27 KB weights (+ 9 KB gate
//     weights, resident for the whole launch) = 147 / 156 KB.
//   * epilogue entries (16 bytes = 8 channels of one pixel per lane): 48 channels = six planes.  Channel fragments 0 and 1 of a pixel
//     fragment pair up as in conv3r (planes 0..3); fragment 2 (planes 4, 5) pairs with fragment 2 of the NEIGHBOURING pixel fragment:
//     after v_permlane16_swap lane row g holds plane 4 + (g >> 1) of pixel fragment 2 k + (g & 1).  18 entries per wave and tile.
//   * EPI_FUSEDMIX: per pixel fragment a "unit": z rounded to the storage type and packed; gate GEMM beta = W [x ; z] as three K steps
//     of two 16-channel fragments each in accumulator-row order -- (x0, x1), (x2, z0), (z1, z2): x arrives in accumulator layout, so
//     pairs of fragments ARE B operands as they stand (gate weights packed to match, PackArgs::frag16 = 4) --; blend; entries.  The
//     x of a step's units is requested in the step BEFORE, ahead of that step's LDS-DMA: the step's closing vmcnt wait covers it and
//     no unit ever waits for HBM.  Twelve units over the tile's first three chunks (three chunks: 0 + 2 + 2, 1 + 1 + 2, 1 + 1 + 2
//     per step) or six (0 + 1 + 1 per chunk: the steps that carry the halo DMA stay free).
// Requirements (the host guards): one N tile (Cout <= 48), Cin a multiple of 32 with three or >= six chunks, 32-bit offsets inside
// four input planes / six output planes.
#pragma once
#include "mz_conv3r.h"

namespace mz {
namespace t3 {

using r3::ic;
using r3::lane_now;

constexpr int FPR = 4;                      // pixel fragments per tile row
constexpr int RPW = 3;                      // tile rows per wave
constexpr int TH = 4 * RPW, TW = 16 * FPR;  // 12 x 64
constexpr int ROWW = TW + 2;
constexpr int NPIX = (TH + 2) * ROWW;           // 924 halo pixels
constexpr int PLANE_PIECES = (NPIX + 63) / 64;  // 15 DMA pieces of 64 entries per plane
constexpr int PLANE_ENT = 64 * PLANE_PIECES;
constexpr int A_PLANE = PLANE_ENT * 16;
constexpr int A_SLOT = 4 * A_PLANE;             // 60 KB
constexpr int NPF = 4 * 3, NF = 3, BN = 16 * NF;
constexpr int GPF = 6;                          // pixel fragments per group
constexpr int B_BASE = 2 * A_SLOT;
constexpr int CHUNK_PIECES = 9 * NF;            // 27 KB of weights per 32-channel chunk
constexpr int SEG_PIECES = 3 * NF;              // three taps
constexpr int MIX_BASE = B_BASE + CHUNK_PIECES * 1024;
constexpr int MIX_PIECES = 3 * NF;              // gate weights: three K steps x three fragments
// 1 KB behind everything else: the target of DUMMY halo pieces.  Every wave issues the same NUMBER of halo pieces (wave 3, whose share of
// a plane is one piece short, issues a piece nobody reads), so that a chunk's first step can close with a COUNTED vmcnt wait that leaves
// the halo image -- issued LAST in the step, needed two steps later -- in flight: it comes from HBM, and waiting for it at the end of the
// step it was requested in cost the helper 1 - 2 k cycles per chunk (tools/stamp_probe_t.py).
constexpr int dump_base(bool fuse) { return fuse ? MIX_BASE + MIX_PIECES * 1024 : MIX_BASE; }
constexpr int lds_bytes(bool fuse) { return dump_base(fuse) + 1024; }
constexpr int HALO_PIECES = 16;                 // halo pieces a wave issues per chunk (four planes x four, dummies included)
static_assert(lds_bytes(true) <= 160 * 1024, "LDS of a CU");
static_assert(NPF == RPW * FPR && NPF == 2 * GPF, "two groups of six pixel fragments per tap");

// byte offset of pixel fragment pf of tap (dy, dx) inside one plane of the halo image, relative to the wave's first row
template <int TAP, int PF> constexpr int a_off() {
    constexpr int DY = TAP / 3, DX = TAP % 3;
    return ((DY + PF / FPR) * ROWW + DX + 16 * (PF % FPR)) * 16;
}

struct Frag {
    u32x4 x[2][GPF];  // [group parity][pixel fragment of the group]
    u32x4 w[3][NF];   // [tap % 3][channel fragment]
};

template <int N> __device__ __forceinline__ void wait_x(u32x4 (&x)[GPF]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_xw(u32x4 (&x)[GPF], u32x4 (&w)[NF]) {
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2])
                 : "n"(N)
                 : "memory");
}

// Group GC = 2 t + h of a chunk: tap t, pixel fragments 6 h .. 6 h + 5, all three channel fragments: 18 MFMAs, pixel-fragment major
// with the channel fragments in serpentine order (one operand changes per MFMA, the B operand every third: conv3r's finding that the
// power-limited chip holds a higher clock that way); odd groups walk their pixel fragments backwards.
// Requests issued meanwhile: M = 0..5 the next group's six pixel fragments, M = 6..8 of a tap's first group the next tap's weights.
template <class TT, int GC, bool ZERO_C, int M>
__device__ __forceinline__ void group_mfmas(f32x4 (&acc)[NPF][NF], Frag& f, const uint32_t a_cur, const uint32_t a_nxt, const uint32_t b_lane) {
    if constexpr (M < GPF * NF) {
        constexpr int t = GC / 2, h = GC % 2, xp = GC & 1, xq = xp ^ 1;
        constexpr int j = M / NF, i = M % NF;
        constexpr int pl = h ? GPF - 1 - j : j, pf = GPF * h + pl, nf = (j & 1) ? NF - 1 - i : i;
        if constexpr (ZERO_C) {  // a tile's first tap WRITES the accumulators (C = 0): nobody clears 144 registers per tile
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if constexpr (TT::IS_BF16)
                acc[pf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f.w[t % 3][nf]), __builtin_bit_cast(bf16x8_t, f.x[xp][pl]), zero, 0, 0, 0);
            else
                acc[pf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, f.w[t % 3][nf]), __builtin_bit_cast(f16x8_t, f.x[xp][pl]), zero, 0, 0, 0);
        } else {
            mma16<TT>(acc[pf][nf], f.w[t % 3][nf], f.x[xp][pl]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M < GPF) {
            if constexpr (h == 0) f.x[xq][M] = lds_read128<a_off<t, GPF + M>()>(a_cur);
            else if constexpr (t + 1 < 9) f.x[xq][M] = lds_read128<a_off<(t + 1 < 9 ? t + 1 : 0), M>()>(a_cur);
            else f.x[xq][M] = lds_read128<a_off<0, M>()>(a_nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (h == 0 && M >= GPF && M < GPF + NF) {
            constexpr int tn = (t + 1) % 9;  // (tap 0 of the NEXT chunk: the same LDS address, refilled by the loader two steps ago)
            f.w[(t + 1) % 3][M - GPF] = lds_read128<(NF * tn + (M - GPF)) * 1024>(b_lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        group_mfmas<TT, GC, ZERO_C, M + 1>(acc, f, a_cur, a_nxt, b_lane);
    }
}

// groups [G, GE) of one step (= three taps)
template <class TT, int G, int GE>
__device__ __forceinline__ void groups(f32x4 (&acc)[NPF][NF], Frag& f, const uint32_t a_cur, const uint32_t a_nxt, const uint32_t b_lane, const bool first) {
    if constexpr (G < GE) {
        constexpr int t = G / 2, h = G % 2;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (G < 2) {  // tap 0 of a chunk that may be the tile's first
            if (first) group_mfmas<TT, G, true, 0>(acc, f, a_cur, a_nxt, b_lane);
            else group_mfmas<TT, G, false, 0>(acc, f, a_cur, a_nxt, b_lane);
        } else {
            group_mfmas<TT, G, false, 0>(acc, f, a_cur, a_nxt, b_lane);
        }
        // what the next group needs (LDS reads return in order): its six pixel fragments were requested first in this group; behind
        // them, in a tap's first group, three weight fragments that only the next TAP needs
        if constexpr (h == 0) wait_x<NF>(f.x[(G + 1) & 1]);
        else wait_xw<0>(f.x[(G + 1) & 1], f.w[(t + 1) % 3]);
        groups<TT, G + 1, GE>(acc, f, a_cur, a_nxt, b_lane, first);
    }
}

// where the epilogue work of a finished tile goes: unit / entry ranges per (chunk, step).  SHORT = the tile has three chunks.
// fused: units = pixel fragments [first, first + count)
template <bool SHORT> constexpr int fuse_first(int k, int sg) {
    if (SHORT) return k == 0 ? (sg == 0 ? 0 : (sg == 1 ? 0 : 2)) : 4 * k + (sg == 0 ? 0 : (sg == 1 ? 1 : 2));
    return 2 * k + (sg == 2 ? 1 : 0);
}
template <bool SHORT> constexpr int fuse_count(int k, int sg) {
    if (SHORT) return k > 2 ? 0 : (k == 0 ? (sg == 0 ? 0 : 2) : (sg == 2 ? 2 : 1));
    return k > 5 ? 0 : (sg == 0 ? 0 : 1);
}
// plain: entries [first, first + count) of 18
template <bool SHORT> constexpr int plain_first(int k, int sg) { return SHORT ? 6 * k + 2 * sg : 3 * k + sg; }
template <bool SHORT> constexpr int plain_count(int k, int sg) { return SHORT ? (k > 2 ? 0 : 2) : (k > 5 ? 0 : 1); }

}  // namespace t3

// EPI: EPI_STORE (SILU: with the activation) or EPI_FUSEDMIX
template <class TT, int EPI, bool SILU>
__global__ __launch_bounds__(512) void conv3t_kernel(const ConvArgs a) {
    using namespace t3;
    constexpr bool FUSE = EPI == EPI_FUSEDMIX;
    static_assert(EPI == EPI_STORE || EPI == EPI_FUSEDMIX, "plain / SiLU store or the fused mix");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int team = w >> 2, wq = w & 3;  // wq: SIMD = tile rows 3 wq .. 3 wq + 2 (compute) = loader index
    const int nchunks = a.nchunks16;       // 32-channel chunks: 3, or >= 6 (the host guards)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- tile walk (as conv3r_kernel: the host's tile list, an XCD's contiguous range of it strided by the workgroups of that XCD;
    // a tile's coordinates are one scalar load, requested two tiles ahead at the start of a helper phase) ----
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3, step = gridDim.x >> 3;
    const int q = a.grid >> 3, rem = a.grid & 7;
    const int cnt = q + (xcd < rem ? 1 : 0);
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    if (pos >= cnt) return;  // uniform over the workgroup
    typedef uint32_t TabE __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) TabE* TabPtr;
    const TabPtr tab = (TabPtr)(uintptr_t)a.tile_tab + base;
    struct TileE { uint32_t yx, bn; };  // y0 | x0 << 16, image (the N tile is always 0 here)
    auto tile_at = [&](int i) __attribute__((always_inline)) {
        const TabE e = tab[i];
        return TileE{e[0], e[1]};
    };
    // a_pos = position of the tile computed in the current phase.  A team in the helper role holds eD = the tile it computed last (its
    // epilogue runs now), eA = tile a_pos, eB = tile a_pos + step (its own next one).
    int a_pos = pos;
    TileE eD, eA, eB;
    auto advance = [&]() __attribute__((always_inline)) { a_pos += step; };
    auto tile_origin = [&](const TileE e, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        b = (int)(e.bn & 0xffffu);
        y0 = (int)(e.yx & 0xffffu);
        x0 = (int)(e.yx >> 16);
    };

    int us = 0;  // halo slot of the chunk that is about to start
    f32x4 acc[NPF][NF];
    Frag f;
    RS_DECL;  // diagnostic builds (-DMZ_DIAG, mz_diag.h; tools/stamp_probe_t.py): counters 0 / 1 = K-loop cycles / tiles, 2 = final epilogue;
              // step class c = (chunk's first step ? 0 : 1) + (epilogue work ? 2 : 0): 4 c + 4 .. + 7 = request + DMA issue / epilogue /
              // vmcnt wait / barrier, 24 + c = steps of the class; 3 = phase start (offsets, epilogue setup)
    const long long plane_in = (long long)a.H * a.W * 16;
    const char* const wbase = (const char*)a.wpk16;  // one N tile: [chunk][tap][fragment][64 lanes][16 B]

    // ------------------------------------------------------------------------------------------------
    // loader role
    // ------------------------------------------------------------------------------------------------
    // Halo image of a chunk: four planes of 15 pieces; wave wq issues the in-plane pieces wq, wq + 4, wq + 8 (and wq + 12 < 15) of every
    // plane.  The plane goes into the instruction's scalar offset (Cin % 32 == 0: every chunk has its four planes; the range check does
    // not see scalar offsets), so a tile costs four per-lane offsets.
    uint32_t hoff[4];
    const char* img_l = nullptr;
    auto set_load_tile = [&](const TileE e) __attribute__((always_inline)) {
        int b, y0, x0;
        tile_origin(e, b, y0, x0);
        img_l = (const char*)a.in0 + (long long)b * a.p0 * plane_in;
        const int lane_ = lane_now();
        // tiles whose whole halo (and the row of pad entries behind it) lies inside the image need no per-entry bounds test
        const bool interior = y0 >= 1 && x0 >= 1 && y0 + TH + 2 <= a.H && x0 + TW + 1 <= a.W;
        const uint32_t delta = ((uint32_t)(y0 - 1) * (uint32_t)a.W + (uint32_t)(x0 - 1)) * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = 64 * (wq + 4 * i) + lane_;
            const int py = (p * 993) >> 16, px = p - py * ROWW;  // p / 66 for p < 1024
            uint32_t o = ((uint32_t)py * (uint32_t)a.W + (uint32_t)px) * 16u + delta;
            if (!interior) {
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool ok = (p < NPIX) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
                o = ok ? o : 0xffffffffu;  // beyond the descriptor: the hardware returns zeros
            }
            hoff[i] = o;
        }
    };
    auto halo_piece = [&](auto pl_tag, auto i_tag, const __amdgpu_buffer_rsrc_t rsrc, char* dst) __attribute__((always_inline)) {
        constexpr int pl = decltype(pl_tag)::value, i = decltype(i_tag)::value;
        if constexpr (i < 3) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (pl * PLANE_PIECES + wq + 4 * i) * 1024), 16,
                                                     (int)hoff[i], (int)((uint32_t)pl * (uint32_t)plane_in), 0, 0);
        } else {
            // in-plane piece wq + 12 exists for wq < 3; wave 3 issues a dummy in its place (all lanes out of range: no memory traffic, zeros
            // into the dump area), so that every wave issues four pieces per plane
            const bool real = wq < PLANE_PIECES - 12;
            char* const d = real ? dst + (pl * PLANE_PIECES + wq + 12) * 1024 : smem + dump_base(FUSE);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)d, 16, real ? (int)hoff[3] : -1,
                                                     (int)((uint32_t)pl * (uint32_t)plane_in), 0, 0);
        }
    };
    // planes [2 half, 2 half + 2) of chunk kc's image -> halo slot `slot`: eight pieces per wave
    auto halo_half = [&](auto half_tag, int kc, int slot) __attribute__((always_inline)) {
        constexpr int P = 2 * decltype(half_tag)::value;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(img_l + 4LL * kc * plane_in), 0, (int)(uint32_t)(4 * plane_in), 0x00020000);
        uint32_t h_off = (uint32_t)slot * (uint32_t)A_SLOT;
        asm volatile("" : "+s"(h_off));  // (opaque: no piece addresses kept alive -- and spilled -- from chunk to chunk)
        char* const dst = smem + h_off;
        halo_piece(ic<P>{}, ic<0>{}, rsrc, dst); halo_piece(ic<P>{}, ic<1>{}, rsrc, dst); halo_piece(ic<P>{}, ic<2>{}, rsrc, dst); halo_piece(ic<P>{}, ic<3>{}, rsrc, dst);
        halo_piece(ic<P + 1>{}, ic<0>{}, rsrc, dst); halo_piece(ic<P + 1>{}, ic<1>{}, rsrc, dst); halo_piece(ic<P + 1>{}, ic<2>{}, rsrc, dst); halo_piece(ic<P + 1>{}, ic<3>{}, rsrc, dst);
    };
    // weight segment sgm (three taps, nine pieces) of chunk kc -> its fixed place in LDS; wave wq issues pieces wq, wq + 4, wq + 8 (< 9)
    auto weight_segment = [&](int kc, int sgm) __attribute__((always_inline)) {
        const uint32_t lo = (uint32_t)lane_now() * 16u;
        const char* src = wbase + ((size_t)kc * CHUNK_PIECES + (size_t)sgm * SEG_PIECES + wq) * 1024u + lo;
        char* dst = smem + B_BASE + (sgm * SEG_PIECES + wq) * 1024;
        glds16(src, dst);
        glds16(src + 4096, dst + 4096);
        if (wq == 0) glds16(src + 8192, dst + 8192);
    };

    // ---- epilogue of the finished tile eD ----
    __amdgpu_buffer_rsrc_t orsrc;
    u32x4 xrsrc = {0u, 0u, 0u, 0u};  // FUSE: buffer descriptor of the block input x (in1) of the finished tile's image, for inline-asm loads
    uint32_t e_pix = 0, eoff1 = 0, eoff2 = 0, x_lane = 0;
    int e_c = 0, e_c2 = 0, e_y = 0;
    auto epi_setup = [&]() __attribute__((always_inline)) {
        const int lane_ = lane_now();
        const int g = lane_ >> 4, c = lane_ & 15;
        int d_b, d_y0, d_x0;
        tile_origin(eD, d_b, d_y0, d_x0);
        e_c = d_x0 + c;
        e_c2 = e_c + 16 * (g & 1);  // entries of channel fragment 2: the lane's pixel belongs to pixel fragment 2 k + (g & 1)
        e_y = d_y0 + RPW * wq;
        const long long plane_o = (long long)a.H * a.W * 16;
        orsrc = __builtin_amdgcn_make_buffer_rsrc((char*)a.out + (long long)d_b * a.p_out * plane_o, 0, (int)(uint32_t)(a.p_out * plane_o), 0x00020000);
        e_pix = ((uint32_t)e_y * (uint32_t)a.W + (uint32_t)e_c) * 16u;
        eoff1 = (uint32_t)(2 * (g & 1) + (g >> 1)) * (uint32_t)plane_o;           // fragments 0, 1: plane 2 (g & 1) + (g >> 1) (entry16(), mz_device.h)
        eoff2 = (uint32_t)(4 + (g >> 1)) * (uint32_t)plane_o + (uint32_t)(g & 1) * 256u;  // fragment 2: plane 4 + (g >> 1), 16 pixels on for odd lane rows
        if constexpr (FUSE) {
            // x in ACCUMULATOR layout: channels 16 nf + 4 g .. + 3 of the lane's pixel = 8 bytes (g & 1) of plane 2 nf + (g >> 1)
            const unsigned long long xb = (unsigned long long)(uintptr_t)((const char*)a.in1 + (long long)d_b * a.p1 * plane_o);
            xrsrc[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb);
            xrsrc[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((xb >> 32) & 0xffffu));
            xrsrc[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a.p1 * plane_o));  // bytes: loads beyond them return zeros
            xrsrc[3] = 0x00020000u;
            x_lane = (uint32_t)(g >> 1) * (uint32_t)plane_o + (uint32_t)(g & 1) * 8u;
        }
    };
    // byte offset of pixel fragment pf relative to e_pix
    auto pf_off = [&](int pf) __attribute__((always_inline)) { return (uint32_t)(pf / FPR) * (uint32_t)a.W * 16u + (uint32_t)(16 * (pf % FPR)) * 16u; };

    // FUSE state: x of up to four pixel fragments (ring by pf & 3: requested one step ahead), z of the unit in work; both as packed
    // pairs in accumulator layout: [channel fragment][2 words]
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ f_x[4][NF];  // (64-bit elements: the inline-asm loads below write them in place)
    uint32_t f_z[NF][2];
    uint32_t held[2];  // fragment-2 words of an even pixel fragment, waiting for its odd neighbour
    auto fuse_x = [&](auto pf_tag) __attribute__((always_inline)) {  // request x of pixel fragment pf
        constexpr int pf = decltype(pf_tag)::value;
        const bool inside = e_y + pf / FPR < a.H && e_c + 16 * (pf % FPR) < a.W;
        const long long plane_o = (long long)a.H * a.W * 16;
        const uint32_t off = e_pix + pf_off(pf) + x_lane;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const uint32_t o = inside ? off + (uint32_t)(2 * nf) * (uint32_t)plane_o : 0xffffffffu;  // planes >= p1 fall out of range: zeros
            // Inline asm: hipcc's waitcnt pass must not see these loads.  It would wait for them in front of their first use with what it
            // can count -- which is vmcnt(0) once conditional DMA pieces lie in between, i.e. for the halo image a step's closing wait
            // deliberately leaves in flight.  The request is older than that step's DMA, so the closing wait of the step it is issued in
            // covers it; x_landed() marks the spot from which the values may be used.
            // The destination is the ring element itself: a copy behind the asm would read the register before the data arrives.
            u32x2_& dst = f_x[pf & 3][nf];
            const u32x4& xr = xrsrc;  // (named: an asm operand alone does not make a generic lambda capture it)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(o), "s"(xr) : "memory");
        }
    };
    // x of pixel fragment pf, requested by fuse_x() at least one closing wait ago: from here on the registers hold it
    auto x_landed = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            u32x2_& r = f_x[pf & 3][nf];  // (named first: an asm operand alone does not make a generic lambda capture the array)
            asm volatile("" : "+v"(r));
        }
    };
    // values of an entry, packed: out of place (accumulator elements are read where they lie), activation / blend as inline-asm pairs
    // of scalar-f32 chains (r3::silu_pair_to / blend_pair_to: no packed-f32 arithmetic beside the partner's MFMA stream)
    auto packed_frag = [&](auto pf_tag, auto nf_tag, uint32_t (&o)[2]) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value, nf = decltype(nf_tag)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v0 = acc[pf][nf][2 * h], v1 = acc[pf][nf][2 * h + 1];
            if constexpr (FUSE) {
                float x0, x1, z0, z1;
                r3::unpack2r<TT>(f_x[pf & 3][nf][h], x0, x1);
                r3::unpack2r<TT>(f_z[nf][h], z0, z1);
                r3::blend_pair_to(v0, v1, acc[pf][nf][2 * h], acc[pf][nf][2 * h + 1], x0, x1, z0, z1, a.inv_mix_scale);
            }
            if constexpr (!FUSE && SILU) r3::silu_pair_to(v0, v1, acc[pf][nf][2 * h], acc[pf][nf][2 * h + 1]);
            o[h] = pack_pair<TT>(f32x2{v0, v1});
        }
    };
    auto store_entry = [&](const uint32_t (&pa)[2], const uint32_t (&pb)[2], uint32_t off) __attribute__((always_inline)) {
        u32x4 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const auto sw = __builtin_amdgcn_permlane16_swap(pa[h], pb[h], false, false);
            o[h] = sw[0];
            o[2 + h] = sw[1];
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)off, 0, 0);
    };
    // entry of channel fragments 0, 1 of pixel fragment pf
    auto entry01 = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
        uint32_t pa[2], pb[2];
        packed_frag(pf_tag, ic<0>{}, pa);
        packed_frag(pf_tag, ic<1>{}, pb);
        const bool inside = e_y + pf / FPR < a.H && e_c + 16 * (pf % FPR) < a.W;
        store_entry(pa, pb, inside ? e_pix + pf_off(pf) + eoff1 : 0xffffffffu);
    };
    // entry of channel fragment 2 of the pixel fragments 2 k (pa) and 2 k + 1 (pb)
    auto entry2 = [&](auto k_tag, const uint32_t (&pa)[2], const uint32_t (&pb)[2]) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value;
        const bool inside = e_y + (2 * k) / FPR < a.H && e_c2 + 16 * ((2 * k) % FPR) < a.W;
        store_entry(pa, pb, inside ? e_pix + pf_off(2 * k) + eoff2 : 0xffffffffu);
    };
    // plain variants: entry E of 18 = pair E / 3: fragments 0, 1 of its two pixel fragments, then fragment 2 of both
    auto plain_entry = [&](auto e_tag) __attribute__((always_inline)) {
        constexpr int E = decltype(e_tag)::value;
        constexpr int k = E / 3, r = E % 3;
        if constexpr (r < 2) {
            entry01(ic<2 * k + r>{});
        } else {
            uint32_t pa[2], pb[2];
            packed_frag(ic<2 * k>{}, ic<2>{}, pa);
            packed_frag(ic<2 * k + 1>{}, ic<2>{}, pb);
            entry2(ic<k>{}, pa, pb);
        }
    };
    // fused unit = pixel fragment pf: z rounded and packed, gate GEMM (9 MFMAs into the partner's stream), blend, entries
    auto fuse_unit = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
        // gate weights: nine fragments in LDS; the first two K steps' are requested before anything else, the third behind the first
        // step's MFMAs (the unit is short of VALU issue and of time; three serial LDS round trips cost a sixth of it)
        const uint32_t mix_lane = lds_base + MIX_BASE + (uint32_t)lane_now() * 16u;
        u32x4 wa[NF], wb[NF], wc[NF];
        auto gate_reads = [&](auto s_tag, u32x4 (&wv)[NF]) __attribute__((always_inline)) {
            constexpr int S = decltype(s_tag)::value;
            wv[0] = lds_read128<(NF * S + 0) * 1024>(mix_lane); wv[1] = lds_read128<(NF * S + 1) * 1024>(mix_lane); wv[2] = lds_read128<(NF * S + 2) * 1024>(mix_lane);
        };
        gate_reads(ic<0>{}, wa);
        gate_reads(ic<1>{}, wb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const f32x4 z = acc[pf][nf];
            uint32_t q0 = pack_pair<TT>(f32x2{z[0], z[1]}), q1 = pack_pair<TT>(f32x2{z[2], z[3]});  // (round to nearest even, as a store of z would)
            asm volatile("" : "+v"(q0), "+v"(q1));  // opaque: no pack -> unpack forwarding that would keep the floats alive
            f_z[nf][0] = q0;
            f_z[nf][1] = q1;
        }
        // B operands: K step s = fragments 2 s, 2 s + 1 of [x0 x1 x2 z0 z1 z2], each two packed words in accumulator-row order
        const u32x4 b0 = {f_x[pf & 3][0][0], f_x[pf & 3][0][1], f_x[pf & 3][1][0], f_x[pf & 3][1][1]};
        const u32x4 b1 = {f_x[pf & 3][2][0], f_x[pf & 3][2][1], f_z[0][0], f_z[0][1]};
        const u32x4 b2 = {f_z[1][0], f_z[1][1], f_z[2][0], f_z[2][1]};
        auto gate_mfmas = [&](auto s_tag, u32x4 (&wv)[NF], const u32x4& b) __attribute__((always_inline)) {
            constexpr int S = decltype(s_tag)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                if constexpr (S == 0) {  // K step 0 WRITES beta (C = 0)
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (TT::IS_BF16) acc[pf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv[nf]), __builtin_bit_cast(bf16x8_t, b), zero, 0, 0, 0);
                    else acc[pf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wv[nf]), __builtin_bit_cast(f16x8_t, b), zero, 0, 0, 0);
                } else {
                    mma16<TT>(acc[pf][nf], wv[nf], b);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2])::"memory");
        gate_mfmas(ic<0>{}, wa, b0);
        gate_reads(ic<2>{}, wc);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2])::"memory");
        gate_mfmas(ic<1>{}, wb, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wc[0]), "+v"(wc[1]), "+v"(wc[2])::"memory");
        gate_mfmas(ic<2>{}, wc, b2);
        // MFMA result -> VALU read is a SOFTWARE hazard on this chip (8 passes: 11 wait states), and hipcc's hazard recogniser does not
        // look inside inline asm: the blend below reads beta from inline-asm chains (r3::blend_pair_to).  Without these wait states the
        // first pair of a unit now and then blended with a stale beta (intermittent, under the partner's MFMA stream only).
        asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        entry01(pf_tag);
        if constexpr ((pf & 1) == 0) {
            packed_frag(pf_tag, ic<2>{}, held);
        } else {
            uint32_t pb[2];
            packed_frag(pf_tag, ic<2>{}, pb);
            entry2(ic<pf / 2>{}, held, pb);
        }
    };

    // fragment stream of a tile's first group: tap 0, pixel fragments 0..5, the three weight fragments of tap 0
    auto prime = [&](int aslot) __attribute__((always_inline)) {
        const int lane_ = lane_now();
        const int g = lane_ >> 4, c = lane_ & 15;
        const uint32_t ab = lds_base + aslot * A_SLOT + g * A_PLANE + ((RPW * wq) * ROWW + c) * 16;
        const uint32_t bb = lds_base + B_BASE + lane_ * 16;
        f.x[0][0] = lds_read128<a_off<0, 0>()>(ab);
        f.x[0][1] = lds_read128<a_off<0, 1>()>(ab);
        f.x[0][2] = lds_read128<a_off<0, 2>()>(ab);
        f.x[0][3] = lds_read128<a_off<0, 3>()>(ab);
        f.x[0][4] = lds_read128<a_off<0, 4>()>(ab);
        f.x[0][5] = lds_read128<a_off<0, 5>()>(ab);
        f.w[0][0] = lds_read128<0 * 1024>(bb);
        f.w[0][1] = lds_read128<1 * 1024>(bb);
        f.w[0][2] = lds_read128<2 * 1024>(bb);
        wait_xw<0>(f.x[0], f.w[0]);
    };

    // One step (weight segment sg of chunk k) of the loader role while the partner team computes it.
    //   WK: 0 no epilogue work; 1 plain entries [ES, ES + EN); 2 fused units [ES, ES + EN); XS / XN: fused: the units of the NEXT step,
    //   whose x is requested here, ahead of the DMA
    //   last: the tile's last chunk (its loads belong to the next tile tB)
    bool okB = false;
    auto loader_step = [&](auto wk_tag, auto es_tag, auto en_tag, auto xs_tag, auto xn_tag, auto sg_tag, auto last_tag, int k) __attribute__((always_inline)) {
        constexpr int WK = decltype(wk_tag)::value, ES = decltype(es_tag)::value, EN = WK == 0 ? 0 : decltype(en_tag)::value;
        constexpr int XS = decltype(xs_tag)::value, XN = decltype(xn_tag)::value, sg = decltype(sg_tag)::value;
        constexpr bool last = decltype(last_tag)::value != 0;
        static_assert(EN <= 2 && XN <= 2, "at most two units / entries per step");
        // VMEM instructions this step issues BEHIND its weight pieces: in a chunk's first step the halo image of the next chunk (the same
        // count in every wave), then the stores.  The step closes with vmcnt(that many) (loads, stores and LDS-DMA retire in issue order):
        // the weight segment has landed -- the compute waves prefetch out of it before the next barrier --, the halo image may stay in
        // flight: the closing wait of the chunk's SECOND step covers it, a step before the compute waves first read it.
        constexpr int VM_AFTER = (sg == 0 ? HALO_PIECES : 0) +
                                 (WK == 1 ? EN : (WK == 2 ? (EN > 0 ? 1 + (ES & 1) : 0) + (EN > 1 ? 1 + ((ES + 1) & 1) : 0) : 0));
        static_assert(VM_AFTER < 64, "vmcnt is a 6-bit counter");
        [[maybe_unused]] constexpr int rs_c = (sg == 0 ? 0 : 1) + (EN > 0 ? 2 : 0);
        RS_BEGIN();
        if constexpr (FUSE && WK == 2) {  // x of this step's units: requested a step ago, covered by that step's closing wait
            if constexpr (EN > 0) x_landed(ic<ES>{});
            if constexpr (EN > 1) x_landed(ic<(EN > 1 ? ES + 1 : ES)>{});
        }
        if constexpr (FUSE && XN > 0) {
            fuse_x(ic<XS>{});
            if constexpr (XN > 1) fuse_x(ic<(XN > 1 ? XS + 1 : XS)>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- this step's DMA: weight segment (sg + 2) % 3 -- of this chunk in the first step, of the next chunk otherwise --, then, in a
        //      chunk's first step, the next chunk's halo image (of tB's first chunk in a tile's last chunk; without a tB the offsets still
        //      describe tA: a harmless re-load into the free slot keeps the piece count) ----
        if constexpr (sg == 0) weight_segment(k, 2);
        else weight_segment(last ? 0 : k + 1, sg - 1);
        if constexpr (sg == 0) {
            halo_half(ic<0>{}, last ? 0 : k + 1, us ^ 1);
            halo_half(ic<1>{}, last ? 0 : k + 1, us ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        RS_LAP(4 * rs_c + 4);
        // ---- this step's share of the finished tile's epilogue, behind the DMA issue ----
        if constexpr (WK == 1) {
            if constexpr (EN > 0) plain_entry(ic<ES>{});
            if constexpr (EN > 1) plain_entry(ic<(EN > 1 ? ES + 1 : ES)>{});
        } else if constexpr (WK == 2) {
            if constexpr (EN > 0) fuse_unit(ic<ES>{});
            if constexpr (EN > 1) fuse_unit(ic<(EN > 1 ? ES + 1 : ES)>{});
        }
        if constexpr (sg == 2) {
            if constexpr (last) {
                // the epilogue is complete: the fragment stream of the next tile.  Plain variants: its first tap WRITES the accumulators; the
                // fused variant clears them here (with the first-tap form hipcc spills 218 registers around the tile loop, as in conv3r)
                if constexpr (FUSE) {
#pragma unroll
                    for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                prime(us ^ 1);
            } else {
                if (k + 2 == nchunks && okB) set_load_tile(eB);  // the next step requests tB's first halo image
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        RS_FENCE();
        RS_LAP(4 * rs_c + 5);
        wait_vmcnt<VM_AFTER>();
        RS_LAP(4 * rs_c + 6);
        __builtin_amdgcn_s_barrier();
        RS_LAP(4 * rs_c + 7);
        RS_COUNT(24 + rs_c);
        if constexpr (sg == 2) us ^= 1;  // the chunk is complete
    };
    // chunk K of a tile's epilogue schedule (K compile-time, k = its run-time twin), SHORT: the tile has three chunks
    auto epi_chunk = [&](auto short_tag, auto k_tag, auto last_tag, int k) __attribute__((always_inline)) {
        constexpr bool SHORT = decltype(short_tag)::value != 0;
        constexpr int K = decltype(k_tag)::value;
        if constexpr (FUSE) {
            // (units of the step after this chunk's last one: the next chunk's first step)
            loader_step(ic<2>{}, ic<fuse_first<SHORT>(K, 0)>{}, ic<fuse_count<SHORT>(K, 0)>{}, ic<fuse_first<SHORT>(K, 1)>{}, ic<fuse_count<SHORT>(K, 1)>{}, ic<0>{}, last_tag, k);
            loader_step(ic<2>{}, ic<fuse_first<SHORT>(K, 1)>{}, ic<fuse_count<SHORT>(K, 1)>{}, ic<fuse_first<SHORT>(K, 2)>{}, ic<fuse_count<SHORT>(K, 2)>{}, ic<1>{}, last_tag, k);
            loader_step(ic<2>{}, ic<fuse_first<SHORT>(K, 2)>{}, ic<fuse_count<SHORT>(K, 2)>{}, ic<fuse_first<SHORT>(K + 1, 0)>{}, ic<fuse_count<SHORT>(K + 1, 0)>{}, ic<2>{}, last_tag, k);
        } else {
            loader_step(ic<1>{}, ic<plain_first<SHORT>(K, 0)>{}, ic<plain_count<SHORT>(K, 0)>{}, ic<0>{}, ic<0>{}, ic<0>{}, last_tag, k);
            loader_step(ic<1>{}, ic<plain_first<SHORT>(K, 1)>{}, ic<plain_count<SHORT>(K, 1)>{}, ic<0>{}, ic<0>{}, ic<1>{}, last_tag, k);
            loader_step(ic<1>{}, ic<plain_first<SHORT>(K, 2)>{}, ic<plain_count<SHORT>(K, 2)>{}, ic<0>{}, ic<0>{}, ic<2>{}, last_tag, k);
        }
    };
    auto plain_chunk = [&](auto last_tag, int k) __attribute__((always_inline)) {
        loader_step(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, last_tag, k);
        loader_step(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<1>{}, last_tag, k);
        loader_step(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<2>{}, last_tag, k);
    };
    auto plain_chunks = [&](int k0) __attribute__((always_inline)) {  // chunks k0 .. nchunks - 1 without epilogue work
        for (int k = k0; k + 1 < nchunks; ++k) plain_chunk(ic<0>{}, k);
        plain_chunk(ic<1>{}, nchunks - 1);
    };
    auto loader_phase = [&](auto epi_tag) __attribute__((always_inline)) {
        constexpr bool DO_EPI = decltype(epi_tag)::value != 0;
        okB = a_pos + step < cnt;
        // the two tiles behind eB: eA / eB of this team's NEXT helper phase (two phases on), requested now, taken over at the end of this phase
        const TileE eA2 = tile_at(a_pos + 2 * step), eB2 = tile_at(a_pos + 3 * step);
        RS_BEGIN();
        set_load_tile(eA);
        if constexpr (DO_EPI) {
            epi_setup();
            RS_FENCE();
            RS_LAP(3);
            if (nchunks == 3) {
                epi_chunk(ic<1>{}, ic<0>{}, ic<0>{}, 0);
                epi_chunk(ic<1>{}, ic<1>{}, ic<0>{}, 1);
                epi_chunk(ic<1>{}, ic<2>{}, ic<1>{}, 2);
            } else {  // six or more
                epi_chunk(ic<0>{}, ic<0>{}, ic<0>{}, 0);
                epi_chunk(ic<0>{}, ic<1>{}, ic<0>{}, 1);
                epi_chunk(ic<0>{}, ic<2>{}, ic<0>{}, 2);
                epi_chunk(ic<0>{}, ic<3>{}, ic<0>{}, 3);
                epi_chunk(ic<0>{}, ic<4>{}, ic<0>{}, 4);
                if (nchunks == 6) {
                    epi_chunk(ic<0>{}, ic<5>{}, ic<1>{}, 5);
                } else {
                    epi_chunk(ic<0>{}, ic<5>{}, ic<0>{}, 5);
                    plain_chunks(6);
                }
            }
        } else {
            plain_chunks(0);
        }
        // this team computes eB next (its epilogue runs in the helper phase after that)
        eD = eB; eA = eA2; eB = eB2;
        asm volatile("" ::"s"(eA.yx), "s"(eA.bn), "s"(eB.yx), "s"(eB.bn));  // (landed here: no scalar load in flight beside the K loop's counted waits)
    };

    // ------------------------------------------------------------------------------------------------
    // compute role: the K loop of tile tA
    // ------------------------------------------------------------------------------------------------
    auto compute_phase = [&]() __attribute__((always_inline)) {
        const int lane_ = lane_now();
        const int g = lane_ >> 4, c = lane_ & 15;
        const uint32_t a_lane = lds_base + g * A_PLANE + ((RPW * wq) * ROWW + c) * 16;
        uint32_t a_cur = a_lane + us * A_SLOT, a_nxt = a_lane + (us ^ 1) * A_SLOT;
        const uint32_t b_lane = lds_base + B_BASE + lane_ * 16;
        RS_BEGIN();
        for (int kc = 0; kc < nchunks; ++kc) {
            const bool first = !FUSE && kc == 0;  // (the fused variant clears its accumulators in the helper role: see there)
            groups<TT, 0, 6>(acc, f, a_cur, a_nxt, b_lane, first);
            __builtin_amdgcn_s_barrier();
            groups<TT, 6, 12>(acc, f, a_cur, a_nxt, b_lane, false);
            __builtin_amdgcn_s_barrier();
            groups<TT, 12, 18>(acc, f, a_cur, a_nxt, b_lane, false);
            __builtin_amdgcn_s_barrier();
            const uint32_t v_ = a_cur; a_cur = a_nxt; a_nxt = v_;
        }
        // (the last group requested fragments of a "next chunk" nobody computes here: its closing wait was lgkmcnt(0), they have landed;
        // the registers are dead from here on)
        RS_LAP(0);
        RS_COUNT(1);
        us ^= nchunks & 1;
    };
    auto final_epilogue = [&]() __attribute__((always_inline)) {
        epi_setup();
        if constexpr (FUSE) {
            // (no partner, no DMA: x one unit ahead, an explicit wait for it -- all but the next unit's three loads and this unit's stores --)
            auto unit = [&](auto pf_tag) __attribute__((always_inline)) {
                constexpr int pf = decltype(pf_tag)::value;
                if constexpr (pf + 1 < NPF) fuse_x(ic<(pf + 1 < NPF ? pf + 1 : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (pf + 1 < NPF) wait_vmcnt<NF>(); else wait_vmcnt<0>();
                x_landed(pf_tag);
                fuse_unit(pf_tag);
            };
            fuse_x(ic<0>{});
            unit(ic<0>{}); unit(ic<1>{}); unit(ic<2>{}); unit(ic<3>{}); unit(ic<4>{}); unit(ic<5>{});
            unit(ic<6>{}); unit(ic<7>{}); unit(ic<8>{}); unit(ic<9>{}); unit(ic<10>{}); unit(ic<11>{});
        } else {
            plain_entry(ic<0>{}); plain_entry(ic<1>{}); plain_entry(ic<2>{}); plain_entry(ic<3>{}); plain_entry(ic<4>{}); plain_entry(ic<5>{});
            plain_entry(ic<6>{}); plain_entry(ic<7>{}); plain_entry(ic<8>{}); plain_entry(ic<9>{}); plain_entry(ic<10>{}); plain_entry(ic<11>{});
            plain_entry(ic<12>{}); plain_entry(ic<13>{}); plain_entry(ic<14>{}); plain_entry(ic<15>{}); plain_entry(ic<16>{}); plain_entry(ic<17>{});
        }
    };

    // ------------------------------------------------------------------------------------------------
    if (team == 1) {
        // prologue: chunk 0 of the first tile (halo image + its first two weight segments), published by B_0
        eA = tile_at(a_pos);
        eB = tile_at(a_pos + step);
        eD = eA;  // (unused: the first helper phase has no epilogue)
        set_load_tile(eA);
        halo_half(ic<0>{}, 0, 0);
        halo_half(ic<1>{}, 0, 0);
        weight_segment(0, 0);
        weight_segment(0, 1);
        if constexpr (FUSE) {  // the gate weights, resident for the whole launch: nine pieces
            const char* msrc = (const char*)a.wmix16 + (uint32_t)lane_now() * 16u;
            glds16(msrc + (size_t)wq * 1024, smem + MIX_BASE + wq * 1024);
            glds16(msrc + (size_t)(wq + 4) * 1024, smem + MIX_BASE + (wq + 4) * 1024);
            if (wq == 0) glds16(msrc + (size_t)8 * 1024, smem + MIX_BASE + 8 * 1024);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // B_0
        loader_phase(ic<0>{});  // (its last step clears the fused variant's accumulators and primes the fragment stream)
        advance();
        if (a_pos >= cnt) { RS_DUMP(); return; }
    } else {
        // its first helper phase (after tile a_pos) works on the tiles behind it
        eD = tile_at(a_pos);
        eA = tile_at(a_pos + step);
        eB = tile_at(a_pos + 2 * step);
        asm volatile("" ::"s"(eD.yx), "s"(eD.bn), "s"(eA.yx), "s"(eA.bn), "s"(eB.yx), "s"(eB.bn));
        __builtin_amdgcn_s_barrier();  // B_0
        if constexpr (FUSE) {
#pragma unroll
            for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        prime(0);
    }
    for (;;) {
        compute_phase();
        advance();
        if (a_pos >= cnt) {
            RS_BEGIN();
            final_epilogue();
            RS_LAP(2);
            RS_DUMP();
            return;
        }
        loader_phase(ic<1>{});
        advance();
        if (a_pos >= cnt) { RS_DUMP(); return; }
    }
}

}  // namespace mz
