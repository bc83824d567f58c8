// conv3r_kernel ("relay"): 3x3 convolution (pad 1, stride 1) on v_mfma_f32_16x16x32_{bf16,f16} whose two waves per SIMD
// ALTERNATE roles from tile to tile, so that a tile's store / SiLU / mix epilogue runs under the NEXT tile's K loop.
//
// Why (EXPERIMENTS.md section 5, round-2 stamps): in round 2's conv3q_kernel (retired in round 4; history up to commit 40ea7e4) every SIMD
// held one compute wave (96 px x 96 channels, 144
// accumulator registers) and one loader wave whose 253 registers sit idle.  A tile of the Cin = 96 full-resolution layers is
// only 3 - 6 K chunks long, and its epilogue (3.9 - 7.4 k cycles, store-path and SiLU bound) runs with the matrix pipe idle:
// those layers reach 0.46 of the MFMA peak where the deep layers reach 0.65.
//
// Here the workgroup is two TEAMS of four waves (team X = waves 0..3, team Y = waves 4..7; waves w and w + 4 share a SIMD).
// Tile i of the workgroup is computed by team i & 1.  While one team runs the K loop of tile i (exactly that kernel's compute
// wave: same tile shape, same fragment stream, same summation order => bit-identical sums), the other team
//   * issues all LDS-DMA of those half-steps (its loader role: halo image one chunk ahead, weight segments two
//     steps ahead -- near the end of the tile these already belong to tile i + 1),
//   * finishes ITS OWN previous tile i - 1 out of its accumulators: SiLU / pack / stores, a few 16-byte entries per step,
//     placed behind the step's DMA issue,
//   * in the tile's last step primes its fragment registers for tile i + 1 (a tile's first tap WRITES the accumulators -- MFMA with
//     C = 0 --, so nobody clears them; the fused variant, which spills with that, clears them here),
// and at the tile boundary the teams swap.  The matrix pipe of a SIMD sees one uninterrupted K-loop stream; both register
// files hold accumulators; a tile's K loop starts on a freshly primed wave, so odd chunk counts need no tap-parity carry.
//
// Stores and LDS-DMA share vmcnt (gfx9: loads, stores and LDS-DMA retire in issue order), so a step's stores are issued
// BEHIND its DMA and the step ends with s_waitcnt vmcnt(#stores of this step): the DMA has landed, the stores may still be
// in flight across the barrier.  For that count to be exact every store is a buffer_store whose out-of-image lanes carry
// an out-of-range offset (dropped by the hardware's range check): the instruction is always issued.
//
// The helper role is the critical path of the short tiles (3 chunks: C = 96 inputs; the fused variant), and every VECTOR instruction
// it issues costs ~9 cycles beside the partner's MFMA stream (transcendentals and v_permlane16_swap 16, packed f32 40:
// tools/microbench/mb_coissue.hip), while scalar instructions issue at full rate.  Hence, in this file:
//   * the activation and the fused variant's blend are inline asm of scalar-f32 instructions, two interleaved chains per block,
//     OUT of place (accumulator elements are read where they lie; hipcc's SLP vectoriser would pair them into v_pk_* otherwise);
//   * an epilogue entry packs to the storage type first and swaps the PACKED words between the two fragments of a pair (two
//     v_permlane16_swap per 16-byte entry, on fresh registers: no accumulator copies, no hazard s_nops);
//   * halo DMA offsets are two per lane and tile: the plane of a piece goes into the instruction's scalar offset (Cin % 32 == 0:
//     the hardware's range check does not see scalar offsets, so every chunk must have its four planes; the host guards);
//   * the lane index behind a step's weight pieces is worked out once per step, the piece index goes into the scalar base.
//
// NSEG = weight segments (= barriers) per 32-channel chunk: 2 (14 + 13 groups, 3 x 28 KB weight slots) or 3
// (9 + 9 + 9 groups = tap rows, 3 x 18 KB slots).  Shipped: 3 everywhere (mz_conv3r.hip).  With three steps per chunk the compute
// role first touches the NEXT halo image in the chunk's third step, so the helper issues that image BEHIND the first step's weight
// pieces and its closing wait leaves the eight pieces in flight (HALO_LATE in loader_step()): the lines have two steps to come out of
// HBM instead of one, and the helper of the 3- and 6-chunk layers no longer sits in front of vmcnt (96 -> 192 -6 %, 192 -> 384 -3.5 %,
// deep layers unchanged; EXPERIMENTS.md R4.9).  NSEG = 3 also leaves 42 KB of LDS free: the fused variant (EPI_FUSEDMIX) keeps the
// 36 KB of AdaptiveResidualMix gate weights resident there for the whole launch.
//
// LDS map: [halo 0 | halo 1] 2 x 32 KB + 3 weight slots (+ fused: 36 KB gate weights).
#pragma once
#include <type_traits>
#include "mz_device.h"
#include "mz_diag.h"

namespace mz {
namespace r3 {

// Pixel tile of a workgroup: 8 rows, wave w owns rows 2 w and 2 w + 1.  GEO 0: 8 x 48, six pixel fragments per wave (three per row);
// GEO 1: 8 x 40, five pixel fragments per wave -- fragment 2 STRADDLES the wave's two rows (lanes c < 8: row 0, columns 32 + c; lanes
// c >= 8: row 1, columns c - 8), which costs one more per-lane LDS base and per-lane store offset and nothing else: widths like 120
// (cfg2's level 4: 67 x 120) that 48 does not divide lose 7.5 % of their MFMAs to padded pixels instead of 29 %.  Both halo images
// are padded to 512 entries per plane (eight DMA pieces), so the LDS map and the loader role are the same.
constexpr int TH = 8;
template <int GEO> struct Geo {
    static_assert(GEO == 0 || GEO == 1, "8 x 48 or 8 x 40");
    static constexpr int TW = GEO == 0 ? 48 : 40;
    static constexpr int NPF = GEO == 0 ? 6 : 5;   // pixel fragments per wave
    static constexpr int ROWW = TW + 2;
    static constexpr int NPIX = 10 * ROWW;         // 500 / 420 halo pixels
    static constexpr int DIV_MAGIC = GEO == 0 ? 1311 : 1561;  // (p * magic) >> 16 = p / ROWW for p < 512
    static constexpr int X0 = 3, X1 = NPF - 3;     // pixel fragments of the next tap requested in a tap's first / second group
    // position of pixel fragment pf relative to the lane's pixel c of the wave's first row; STRADDLE: per lane (see above)
    static constexpr bool straddle(int pf) { return GEO == 1 && pf == 2; }
    static constexpr int dy(int pf) { return GEO == 0 ? pf / 3 : (pf >= 3 ? 1 : 0); }
    static constexpr int dx(int pf) { return GEO == 0 ? 16 * (pf % 3) : (pf < 2 ? 16 * pf : (pf == 2 ? 0 : 8 + 16 * (pf - 3))); }
};
constexpr int PLANE_ENT = 512;       // padded: 4 planes = 32 whole DMA instructions
constexpr int A_PLANE = PLANE_ENT * 16;
constexpr int A_SLOT = 4 * A_PLANE;  // 32 KB
constexpr int NT = 3, NF = 6, BN = 96;
constexpr int NG = 9 * NT;           // groups per 32-channel chunk
constexpr int B_BASE = 2 * A_SLOT;
constexpr int CHUNK_PIECES = 2 * NG; // 1-KiB pieces of a chunk's packed weights
constexpr int MIX_PIECES = 4 * NT * NT;  // gate weights of the fused mix: 2 NT K-steps x NF fragments

template <int NSEG> struct Seg {
    static_assert(NSEG == 2 || NSEG == 3, "two or three weight segments per chunk");
    static constexpr int start(int s) { return NSEG == 2 ? (s <= 0 ? 0 : (s == 1 ? (NG + 1) / 2 : NG)) : (NG / 3) * s; }
    static constexpr int of(int G) { return NSEG == 2 ? (G < start(1) ? 0 : 1) : G / (NG / 3); }
    static constexpr int pieces(int s) { return 2 * (start(s + 1) - start(s)); }
    static constexpr int SLOT = pieces(0) * 1024;  // the largest segment is the first
    static constexpr int MIX_BASE = B_BASE + 3 * SLOT;   // fused variant, 36 KB: the gate weights
    static constexpr int lds_bytes(bool fuse) { return fuse ? MIX_BASE + MIX_PIECES * 1024 : B_BASE + 3 * SLOT; }
};

// byte offset of pixel fragment pf of tap (dy, dx) inside one plane of the halo image, relative to the wave's first row
// (straddling fragment: relative to the lane's SECOND base, which holds its per-lane displacement)
template <int GEO, int TAP, int PF> constexpr int a_off() {
    using GG = Geo<GEO>;
    constexpr int DY = TAP / 3, DX = TAP % 3;
    return ((DY + GG::dy(PF)) * GG::ROWW + DX + GG::dx(PF)) * 16;
}

template <int NPF> struct Frag {
    u32x4 x[2][NPF];  // [tap parity][pixel fragment]
    u32x4 w[3][2];    // [group % 3][channel fragment of the pair]: requested TWO groups (24 MFMAs) ahead
};

template <int N> __device__ __forceinline__ void wait_w(u32x4& w0, u32x4& w1) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w0), "+v"(w1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_wx(u32x4& w0, u32x4& w1, u32x4 (&x)[6]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(w0), "+v"(w1), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5])
                 : "n"(N)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_wx(u32x4& w0, u32x4& w1, u32x4 (&x)[5]) {
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : "+v"(w0), "+v"(w1), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4])
                 : "n"(N)
                 : "memory");
}

// LDS read addresses of one lane: halo image of the current / next chunk, weight slot of the current / next step
struct Bases {
    uint32_t a_cur, a_nxt, b_cur, b_nxt;
    uint32_t a2_cur, a2_nxt;  // GEO 1: the same for the straddling pixel fragment (unused, and optimised away, otherwise)
};
template <int GEO, int TAP, int PF> __device__ __forceinline__ u32x4 read_x(const Bases& bs, const bool next_image) {
    if constexpr (Geo<GEO>::straddle(PF)) return lds_read128<a_off<GEO, TAP, PF>()>(next_image ? bs.a2_nxt : bs.a2_cur);
    else return lds_read128<a_off<GEO, TAP, PF>()>(next_image ? bs.a_nxt : bs.a_cur);
}

// Group GC of a chunk: tap t = GC / 3, channel-fragment pair n = GC % 3: 12 MFMAs.  While they issue, the wave requests the
// weight pair of group GC + 2 and (n < 2) three pixel fragments of tap t + 1 -- from the NEXT slot / halo image where the
// group or tap index runs past this segment / chunk.  Tap t reads pixel buffer (t + XP) & 1 (a chunk has 9 taps, so the
// parity of a chunk's tap 0 flips from chunk to chunk; XP = chunk index & 1 inside the tile).
template <class TT, int NSEG, int GEO, int GC, int XP, bool ZERO_C, int M>
__device__ __forceinline__ void group_mfmas(f32x4 (&acc)[Geo<GEO>::NPF][NF], Frag<Geo<GEO>::NPF>& f, const Bases& bs) {
    using GG = Geo<GEO>;
    constexpr int NPF = GG::NPF;
    if constexpr (M < 2 * NPF) {
        using S = Seg<NSEG>;
        constexpr int t = GC / 3, n = GC % 3, xp = (t + XP) & 1, xq = xp ^ 1;
        constexpr int sg = S::of(GC);
        constexpr int Gs = S::start(sg), Ge = S::start(sg + 1);
        constexpr int T = GC + 2;      // group whose weights are requested now
        constexpr bool t_here = T < Ge;
        constexpr int t_idx = t_here ? T - Gs : T - Ge;  // its index inside its segment
        // Order of a group's 12 MFMAs: pixel-fragment major, the channel pair in serpentine order, and the odd pair of a tap walks the
        // pixel fragments backwards -- from one MFMA to the next exactly ONE operand changes, and the B operand (pixels) only every
        // second time (18 -> 16 changes per tap instead of 36).  Same FLOPs, same registers, same sums (every accumulator still sees
        // its MFMAs in K order), but the chip is power-limited in this loop and holds a higher clock: deep layers -2 %, whole forward
        // -1 % against the channel-major raster order (tools/microbench/mb_order.hip).
        constexpr int pf = (n & 1) ? NPF - 1 - M / 2 : M / 2, k = ((M / 2) & 1) ? 1 - (M & 1) : (M & 1);
        if constexpr (ZERO_C) {  // a tile's first tap WRITES the accumulators (C = 0): nobody has to clear 144 registers per tile
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if constexpr (TT::IS_BF16)
                acc[pf][2 * n + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f.w[GC % 3][k]), __builtin_bit_cast(bf16x8_t, f.x[xp][pf]), zero, 0, 0, 0);
            else
                acc[pf][2 * n + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, f.w[GC % 3][k]), __builtin_bit_cast(f16x8_t, f.x[xp][pf]), zero, 0, 0, 0);
        } else {
            mma16<TT>(acc[pf][2 * n + k], f.w[GC % 3][k], f.x[xp][pf]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M < 2) {
            f.w[T % 3][M] = lds_read128<(2 * t_idx + M) * 1024>(t_here ? bs.b_cur : bs.b_nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n < 2 && M >= 2 && M < 2 + (n == 0 ? GG::X0 : GG::X1)) {
            constexpr int pfn = GG::X0 * n + (M - 2);
            if constexpr (t + 1 < 9) f.x[xq][pfn] = read_x<GEO, (t + 1 < 9 ? t + 1 : 0), pfn>(bs, false);
            else f.x[xq][pfn] = read_x<GEO, 0, pfn>(bs, true);
            __builtin_amdgcn_sched_barrier(0);
        }
        group_mfmas<TT, NSEG, GEO, GC, XP, ZERO_C, M + 1>(acc, f, bs);
    }
}

// groups [G, GE) of one step
template <class TT, int NSEG, int GEO, int G, int GE, int XP>
__device__ __forceinline__ void groups(f32x4 (&acc)[Geo<GEO>::NPF][NF], Frag<Geo<GEO>::NPF>& f, const Bases& bs, bool first) {
    using GG = Geo<GEO>;
    if constexpr (G < GE) {
        constexpr int t = G / 3, n = G % 3;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (G < 3 && XP == 0) {  // tap 0 of a chunk that may be the tile's first (a tile starts on parity 0)
            if (first) group_mfmas<TT, NSEG, GEO, G, XP, true, 0>(acc, f, bs);
            else group_mfmas<TT, NSEG, GEO, G, XP, false, 0>(acc, f, bs);
        } else {
            group_mfmas<TT, NSEG, GEO, G, XP, false, 0>(acc, f, bs);
        }
        // what the NEXT group needs (also across the end of this step: the stream continues behind the barrier).  LDS reads
        // return in order.  Reads requested per group: n = 0: two weight + X0 pixel fragments, n = 1: two + X1, n = 2: two weight fragments.
        constexpr int wn = (G + 1) % 3, xn = (t + 1 + XP) & 1;
        if constexpr (n == 2)
            wait_wx<2>(f.w[wn][0], f.w[wn][1], f.x[xn]);
        else if constexpr (n == 1)
            wait_w<GG::X0 + 2 + GG::X1>(f.w[wn][0], f.w[wn][1]);  // (the weights of group n = 2 were requested at the head of group n = 0)
        else
            wait_w<2 + GG::X0>(f.w[wn][0], f.w[wn][1]);
        groups<TT, NSEG, GEO, G + 1, GE, XP>(acc, f, bs, first);
    }
}

// gate GEMM of the fused mix (as conv3s_kernel<.., FUSE>): 2 NT K-steps of NF weight fragments, walked in half steps of NT
// fragments; the next half step's fragments are requested before the current one's MFMAs are issued
template <int H, int I> __device__ __forceinline__ void gate_reads(u32x4 (&wv)[NT], uint32_t addr) {
    if constexpr (I < NT) {
        constexpr int ks = H >> 1, part = H & 1;
        wv[I] = lds_read128<(ks * 2 * NT + part * NT + I) * 1024>(addr);
        gate_reads<H, I + 1>(wv, addr);
    }
}
template <int N> __device__ __forceinline__ void gate_wait(u32x4 (&wv)[NT]) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]) : "n"(N) : "memory");
}
template <class TT, int H>
__device__ __forceinline__ void gate_halves(f32x4 (&beta)[NF], const u32x4 (&xf)[NT], const u32x4 (&zf)[NT], u32x4 (&wa)[NT],
                                            u32x4 (&wb)[NT], uint32_t addr) {
    if constexpr (H < 4 * NT) {
        constexpr int ks = H >> 1, part = H & 1;
        constexpr bool more = H + 1 < 4 * NT;
        if constexpr (more) gate_reads<H + 1, 0>(wb, addr);
        gate_wait<(more ? NT : 0)>(wa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if constexpr (ks == 0) mma16_first<TT>(beta[part * NT + i], wa[i], xf[0]);  // (the gate's first K step writes beta: nobody clears it)
            else if constexpr (ks < NT) mma16<TT>(beta[part * NT + i], wa[i], xf[ks]);
            else mma16<TT>(beta[part * NT + i], wa[i], zf[ks - NT]);
        }
        __builtin_amdgcn_sched_barrier(0);
        gate_halves<TT, H + 1>(beta, xf, zf, wb, wa, addr);
    }
}
template <class TT> __device__ __forceinline__ void unpack2r(uint32_t v, float& lo, float& hi) {
    if constexpr (TT::IS_BF16) {
        lo = __builtin_bit_cast(float, v << 16);
        hi = __builtin_bit_cast(float, v & 0xffff0000u);
    } else {
        lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(v & 0xffff));
        hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16));
    }
}
template <class TT> __device__ __forceinline__ uint32_t pack2r(float lo, float hi) {
    if constexpr (TT::IS_BF16) return pack_bf16(lo, hi);
    else return pack_f16(lo, hi);
}

template <int V> using ic = std::integral_constant<int, V>;

// lane index, recomputed where it is needed (v_mbcnt): no register holds it across the K loop, whose 253 registers are all taken,
// and nothing derived from it can be hoisted out of the tile loop (and spilled)
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// v * sigmoid(v) of TWO values: v_mul, v_exp, v_add, v_rcp, v_mul each, as ONE inline-asm block with the two chains
// interleaved.  Inline asm because hipcc's SLP vectoriser pairs the multiplies and adds into v_pk_* (7 x slower beside the
// partner's MFMA stream); one block because hipcc's hazard recogniser does not look inside inline asm: a VALU instruction that
// reads the result of a transcendental one needs ONE wait state, which the other chain's instruction provides (no s_nop).
// The same operations in the same order as silu2() / sigmoidf_(): identical bits.
// OUT of place (the inputs stay untouched: accumulator elements need no copy into scratch registers first; the results double as
// the chains' temporaries)
__device__ __forceinline__ void silu_pair_to(float& ra, float& rb, const float a, const float b) {
    asm("v_mul_f32 %0, 0xbfb8aa3b, %2\n\t"
        "v_mul_f32 %1, 0xbfb8aa3b, %3\n\t"
        "v_exp_f32 %0, %0\n\t"
        "v_exp_f32 %1, %1\n\t"
        "v_add_f32 %0, 1.0, %0\n\t"
        "v_add_f32 %1, 1.0, %1\n\t"
        "v_rcp_f32 %0, %0\n\t"
        "v_rcp_f32 %1, %1\n\t"
        "v_mul_f32 %0, %2, %0\n\t"
        "v_mul_f32 %1, %3, %1"
        : "=&v"(ra), "=&v"(rb)
        : "v"(a), "v"(b));
}

// x + sigmoid(alpha) sigmoid(beta) (z - x) of TWO values, out of place: blend_()'s operations in blend_()'s order
// (mz_device.h: v_mul, v_exp, v_fma, v_rcp, v_sub, v_fma; identical bits), as one inline-asm block of two interleaved chains for the
// same reasons as silu_pair_to() -- left to hipcc, the SLP vectoriser pairs the adds and fmas into v_pk_add_f32 / v_pk_fma_f32.
__device__ __forceinline__ void blend_pair_to(float& o0, float& o1, const float b0, const float b1, const float x0, const float x1,
                                              const float z0, const float z1, const float inv_s) {
    float d0, d1;
    asm("v_mul_f32 %0, 0xbfb8aa3b, %4\n\t"
        "v_mul_f32 %1, 0xbfb8aa3b, %5\n\t"
        "v_exp_f32 %0, %0\n\t"
        "v_exp_f32 %1, %1\n\t"
        "v_fma_f32 %0, %0, %10, %10\n\t"
        "v_fma_f32 %1, %1, %10, %10\n\t"
        "v_rcp_f32 %0, %0\n\t"
        "v_rcp_f32 %1, %1\n\t"
        "v_sub_f32 %2, %8, %6\n\t"
        "v_sub_f32 %3, %9, %7\n\t"
        "v_fma_f32 %0, %0, %2, %6\n\t"
        "v_fma_f32 %1, %1, %3, %7"
        : "=&v"(o0), "=&v"(o1), "=&v"(d0), "=&v"(d1)
        : "v"(b0), "v"(b1), "v"(x0), "v"(x1), "v"(z0), "v"(z1), "s"(inv_s));
}

}  // namespace r3

// EPI: EPI_STORE (SILU: with the activation), EPI_D2S, EPI_FUSEDMIX (needs NSEG = 3).  SILU is a template parameter: a run-time
// branch around the activation made hipcc copy every value twice more on its way through the epilogue.
// RAG: tiles of exactly TWO chunks whose second chunk lacks planes (Cin = 48: conv1 of the 48-channel models' level-1 block): the pieces of
// the missing planes are issued with every lane out of range -- the hardware writes zeros into LDS, the step's piece count stays what the
// closing vmcnt expects --, the weights are packed with K padded to 64.  A quarter of the MFMAs multiply zeros; the tile is bound by its
// 18 SiLU entries in the helper role either way (three per step), which here run under the K loop instead of behind it.
template <class TT, int NSEG, int EPI, bool SILU, int GEO = 0, bool RAG = false>
__global__ __launch_bounds__(512) void conv3r_kernel(const ConvArgs a) {
    using namespace r3;
    using S = Seg<NSEG>;
    using GG = Geo<GEO>;
    constexpr int NPF = GG::NPF, TW = GG::TW, ROWW = GG::ROWW, NPIX = GG::NPIX;
    constexpr int NE = NPF * NT;  // 16-byte epilogue entries per wave and tile
    constexpr bool FUSE = EPI == EPI_FUSEDMIX;
    static_assert(!FUSE || GEO == 0, "the fused variant is built for the 8 x 48 tile");
    constexpr int B_SLOT = S::SLOT;
    static_assert(!FUSE || NSEG == 3, "the gate weights need the LDS that three weight segments leave free");
    static_assert(!RAG || (NSEG == 3 && EPI == EPI_STORE && GEO == 0), "the ragged two-chunk variant: three steps per chunk, plain stores, 8 x 48 tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int team = w >> 2, wq = w & 3;  // wq: SIMD = tile rows 2 wq, 2 wq + 1 (compute) = loader index
    const int nchunks = a.nchunks16;       // 32-channel chunks, >= 3 (RAG: exactly 2; the host guards)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- tile walk: the host lists the launch's tiles in walk order (mz_host.cpp: tile_table(); a.grid entries of eight bytes, the
    // group walk of mz_device.h with the padding ids dropped).  XCD x owns the x-th eighth of the list, its workgroups stride through it.
    // Tile coordinates come out of the table with ONE scalar load per tile, requested two tiles ahead at the start of a helper phase:
    // the divisions of tile_of_s() / tile_rc_s() (three chains of ~50 scalar instructions per phase, at ~5 cycles each in the role
    // that is the critical path of the short tiles) are gone from the kernel. ----
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3, step = gridDim.x >> 3;
    const int q = a.grid >> 3, rem = a.grid & 7;
    const int cnt = q + (xcd < rem ? 1 : 0);
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    if (pos >= cnt) return;  // uniform over the workgroup
    // (constant address space: a wave-uniform load from it is an s_load whose wait hipcc places itself; the table is padded behind its
    // last entry, so entries past an XCD's range may be loaded -- and are never used: a_pos + .. < cnt decides)
    typedef uint32_t TabE __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) TabE* TabPtr;
    const TabPtr tab = (TabPtr)(uintptr_t)a.tile_tab + base;
    struct TileE { uint32_t yx, bn; };  // y0 | x0 << 16, image | N tile << 16
    auto tile_at = [&](int i) __attribute__((always_inline)) {
        const TabE e = tab[i];
        return TileE{e[0], e[1]};
    };
    // a_pos = position of the tile computed in the current phase (by this team or by its partner).  A team in the helper role holds
    // eD = the tile it computed last (its epilogue runs now), eA = tile a_pos, eB = tile a_pos + step (its own next one).
    int a_pos = pos;
    TileE eD, eA, eB;
    auto advance = [&]() __attribute__((always_inline)) { a_pos += step; };
    auto tile_origin = [&](const TileE e, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        b = (int)(e.bn & 0xffffu);
        y0 = (int)(e.yx & 0xffffu);
        x0 = (int)(e.yx >> 16);
    };
    auto tile_nt = [](const TileE e) __attribute__((always_inline)) { return (int)(e.bn >> 16); };

    // slot counters of the step / chunk that is about to start: weights(h) live in slot hs = h % 3, halo(u) in slot us = u & 1
    int hs = 0, us = 0;
    auto next3 = [](int v) __attribute__((always_inline)) { return v == 2 ? 0 : v + 1; };

    f32x4 acc[NPF][NF];
    Frag<NPF> f;
    RS_DECL;  // diagnostic builds: counters 0/1 = K-loop cycles / tiles; 4 c .. 4 c + 3 = DMA issue / epilogue / vmcnt wait / barrier of loader
              // step class c = 1 + 2 (epilogue step) + (not a chunk's first step), 20 + c = steps of the class

    const size_t chunk_bytes = (size_t)CHUNK_PIECES * 1024;
    const long long plane_in = (long long)a.H * a.W * 16;

    // ------------------------------------------------------------------------------------------------
    // loader role
    // ------------------------------------------------------------------------------------------------
    // Halo DMA offsets of this wave's 8 pieces (pieces wq + 4 i = entries [64 (wq + 4 i), + 64) of the 4-plane image) for the tile
    // being loaded.  Piece wq + 4 i lies in plane i >> 1 at plane entries p = 64 (wq + 4 (i & 1)) + lane: the PLANE term is the
    // same for every lane and tile and goes into the instruction's scalar offset, so a tile costs two per-lane offsets.  (Every
    // vector instruction of a loader wave issues ~9 cycles apart beside the partner's MFMA stream: the eight-offset version of
    // this function and a hand-off of its results to the other team through LDS took ~1.9 k cycles per tile, tools/stamp_probe_r.py.)  The hardware's range check
    // may see the per-lane offset only, so all four planes of every chunk must exist: Cin % 32 == 0 (the host guards).
    uint32_t hoff[2];
    const char* img_l = nullptr;
    auto set_load_tile = [&](const TileE e) __attribute__((always_inline)) {
        int b, y0, x0;
        tile_origin(e, b, y0, x0);
        img_l = (const char*)a.in0 + (long long)b * a.p0 * plane_in;
        const int lane_ = lane_now();
        // Tiles whose whole 10 x 50 halo (and the row of pad entries behind it) lies inside the image need no per-entry bounds
        // test.  (Pad entries p >= 500 are never read by the compute waves: whatever in-range bytes they fetch are harmless, and
        // an out-of-range offset reads zeros.)
        const bool interior = y0 >= 1 && x0 >= 1 && y0 + TH + 2 <= a.H && x0 + TW + 1 <= a.W;
        const uint32_t delta = ((uint32_t)(y0 - 1) * (uint32_t)a.W + (uint32_t)(x0 - 1)) * 16u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = 64 * (wq + 4 * j) + lane_;
            const int py = (p * GG::DIV_MAGIC) >> 16, px = p - py * ROWW;  // p / ROWW for p < 512
            uint32_t o = ((uint32_t)py * (uint32_t)a.W + (uint32_t)px) * 16u + delta;
            if (!interior) {
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool ok = (p < NPIX) & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
                o = ok ? o : 0xffffffffu;  // beyond the descriptor: the hardware returns zeros
            }
            hoff[j] = o;
        }
    };
    // LDS-DMA, one 1-KiB piece at a time (the pieces of a step are interleaved with epilogue arithmetic: an LDS-DMA instruction
    // holds the issuing wave for ~50 cycles -- the CU's L1 -> LDS path --, which the VALU work between two pieces hides).
    // halo piece i (0..7) of this wave: entries [64 (wq + 4 i), + 64) of the image
    // (rag: the image of a chunk that lacks planes -- RAG only --: the pieces of planes >= a.ragged_planes carry an out-of-range offset)
    auto halo_piece = [&](auto i_tag, const __amdgpu_buffer_rsrc_t rsrc, char* dst, [[maybe_unused]] const bool rag = false) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;
        uint32_t off = hoff[i & 1];
        if constexpr (RAG) {
            if (rag && (i >> 1) >= a.ragged_planes) off = 0xffffffffu;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (wq + 4 * i) * 1024), 16,
                                                 (int)off, (int)((uint32_t)(i >> 1) * (uint32_t)plane_in), 0, 0);
    };
    auto halo_rsrc = [&](int kc) __attribute__((always_inline)) {  // the four planes of chunk kc (all exist: Cin % 32 == 0)
        return __builtin_amdgcn_make_buffer_rsrc((void*)(img_l + 4LL * kc * plane_in), 0, (int)(uint32_t)(4 * plane_in), 0x00020000);
    };
    // weight piece j = wq + 4 i of a segment of `pieces` pieces (s0 = the segment in HBM: wave-uniform; dst = its slot).
    // lo = 16 lane_now(), worked out ONCE per step: the piece index goes into the scalar base; the wave's piece COUNT is compared with
    // the compile-time i (per-piece indices wq + 4 i would be six more scalars for hipcc to keep alive -- and spill -- across steps)
    auto wseg_count = [&](int pieces) __attribute__((always_inline)) { return pieces > wq ? (pieces - wq + 3) >> 2 : 0; };
    auto wseg_piece = [&](auto i_tag, const char* s0, int mine, char* dst, uint32_t lo) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;
        if (i < mine) glds16(s0 + (size_t)wq * 1024u + (size_t)i * 4096u + lo, dst + wq * 1024 + i * 4096);
    };
    constexpr int WP = (S::pieces(0) + 3) / 4;  // weight pieces per wave and step, at most
    auto wsrc_of = [&](int nt) __attribute__((always_inline)) {
        return (const char*)a.wpk16 + (size_t)nt * nchunks * chunk_bytes;
    };

    // ---- epilogue of the finished tile `done`, one 16-byte entry (pixel fragment pf, channel-fragment pair n) at a time ----
    // per tile and lane: pix = byte offset of the lane's pixel of fragment 0 inside a plane (D2S: of its 2 x 2 target block),
    // eoff[n] = offset of the entry of pair n relative to pix, or 0xffffffff where the channel does not exist
    __amdgpu_buffer_rsrc_t orsrc;
    u32x4 xrsrc = {0u, 0u, 0u, 0u};  // FUSE: buffer descriptor of the block input x (in1) of the finished tile's image, for inline-asm loads
    uint32_t x_lane = 0;           // FUSE: offset of the lane's 8 bytes of channel fragment 0 relative to e_pix
    uint32_t e_pix = 0, eoff[NT] = {0, 0, 0};
    int e_c = 0, e_y = 0;
    [[maybe_unused]] int e_sdy = 0, e_sdx = 0;     // GEO 1: row / column displacement of the lane's pixel of the straddling fragment
    [[maybe_unused]] uint32_t e_soff = 0;          // ... and its byte offset relative to e_pix
    auto epi_setup = [&]() __attribute__((always_inline)) {
        const int lane_ = lane_now();
        const int g = lane_ >> 4, c = lane_ & 15;
        const int lane_cu = 2 * (g & 1) + (g >> 1);  // 16-byte unit of the lane inside a channel-fragment pair's 4 planes (entry16())
        int d_b, d_y0, d_x0;
        tile_origin(eD, d_b, d_y0, d_x0);
        const int d_nbase = tile_nt(eD) * BN;
        e_c = d_x0 + c;
        e_y = d_y0 + 2 * wq;
        if constexpr (GEO == 1) {
            e_sdy = c >= 8 ? 1 : 0;
            e_sdx = c >= 8 ? -8 : 32;
        }
        if constexpr (EPI == EPI_D2S) {
            const long long plane_o = (long long)a.Hout * a.Wout * 16;
            orsrc = __builtin_amdgcn_make_buffer_rsrc((char*)a.out + (long long)d_b * a.p_out * plane_o, 0,
                                                      (int)(uint32_t)(a.p_out * plane_o), 0x00020000);
            e_pix = ((uint32_t)(2 * e_y) * (uint32_t)a.Wout + (uint32_t)(2 * e_c)) * 16u;
            if constexpr (GEO == 1) e_soff = (uint32_t)((2 * e_sdy * a.Wout + 2 * e_sdx) * 16);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int nch = d_nbase + (4 * n + lane_cu) * 8;
                const int ij = nch / a.cp_out, ch = nch - ij * a.cp_out;
                eoff[n] = nch < 4 * a.cp_out
                              ? (uint32_t)(ch >> 3) * (uint32_t)plane_o + ((uint32_t)(ij >> 1) * (uint32_t)a.Wout + (uint32_t)(ij & 1)) * 16u
                              : 0xffffffffu;
            }
        } else {
            const long long plane_o = (long long)a.H * a.W * 16;
            const int p_first = d_nbase >> 3;
            const int planes = a.p_out - p_first < 4 * NT ? a.p_out - p_first : 4 * NT;
            orsrc = __builtin_amdgcn_make_buffer_rsrc((char*)a.out + ((long long)d_b * a.p_out + p_first) * plane_o, 0,
                                                      (int)(uint32_t)(planes * plane_o), 0x00020000);
            e_pix = ((uint32_t)e_y * (uint32_t)a.W + (uint32_t)e_c) * 16u;
            if constexpr (GEO == 1) e_soff = (uint32_t)((e_sdy * a.W + e_sdx) * 16);
#pragma unroll
            for (int n = 0; n < NT; ++n) eoff[n] = (uint32_t)(4 * n + lane_cu) * (uint32_t)plane_o;  // planes that do not exist fall out of range
            if constexpr (FUSE) {
                // x in ACCUMULATOR layout: channels 16 nf + 4 g .. + 3 of the lane's pixel = 8 bytes (g & 1) of plane 2 nf + (g >> 1)
                const unsigned long long xb = (unsigned long long)(uintptr_t)((const char*)a.in1 + (long long)d_b * a.p1 * plane_o);
                xrsrc[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb);
                xrsrc[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((xb >> 32) & 0xffffu));
                xrsrc[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a.p1 * plane_o));  // bytes: loads beyond them return zeros
                xrsrc[3] = 0x00020000u;
                x_lane = (uint32_t)(g >> 1) * (uint32_t)plane_o + (uint32_t)(g & 1) * 8u;
            }
        }
    };
    // entry E = 3 pf + n: the activation OUT of place (accumulator elements are read where they lie), the values packed to the
    // storage type, and only then v_permlane16_swap pairs the two fragments' quads into the lane's 8 channels (as entry16(),
    // mz_device.h) -- on the PACKED words: two swaps per entry instead of four, on fresh registers (hipcc copied every accumulator
    // element before swapping it in place: with the hazard s_nops a quarter of the helper's epilogue instructions, each ~9 cycles
    // beside the partner's MFMA stream).  Identical bits: activation and rounding are per element, the swap only moves lanes.
    // SiLU in SCALAR f32 instructions: the same operations in the same order as silu2(), but no packed-f32 arithmetic --
    // v_pk_mul_f32 / v_pk_add_f32 issue 7 x slower while the SIMD's other wave streams MFMAs (tools/microbench/mb_coissue.hip:
    // 40 cycles each against 9 for v_mul_f32 and 16 for v_exp_f32 / v_rcp_f32)
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ f_xq[2][NF];    // x of pixel fragment pf in buffer pf & 1 (requested ONE CHUNK AHEAD of its use: the loads come from HBM):
                           // accumulator layout, two packed pairs per channel fragment (64-bit elements: the inline-asm loads write them in place)
    u32x4 f_zb[NT];        // its z as B operands (live from part A to part D of a chunk)
    auto entry_words = [&](auto e_tag, u32x4& o) __attribute__((always_inline)) {
        constexpr int E = decltype(e_tag)::value;
        constexpr int pf = E / NT, n = E % NT;
        uint32_t pa[2], pb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float a0 = acc[pf][2 * n][2 * h], a1 = acc[pf][2 * n][2 * h + 1], b0 = acc[pf][2 * n + 1][2 * h], b1 = acc[pf][2 * n + 1][2 * h + 1];
            if constexpr (FUSE) {
                // part D of the fused variant: the accumulators hold the gate beta; x and z of the pixel fragment in work are in
                // f_xq / f_zb (accumulator layout, packed pairs)
                float xa[2], za[2], xc[2], zc[2];
                unpack2r<TT>(f_xq[pf & 1][2 * n][h], xa[0], xa[1]);
                unpack2r<TT>(f_zb[n][h], za[0], za[1]);
                unpack2r<TT>(f_xq[pf & 1][2 * n + 1][h], xc[0], xc[1]);
                unpack2r<TT>(f_zb[n][2 + h], zc[0], zc[1]);
                blend_pair_to(a0, a1, acc[pf][2 * n][2 * h], acc[pf][2 * n][2 * h + 1], xa[0], xa[1], za[0], za[1], a.inv_mix_scale);
                blend_pair_to(b0, b1, acc[pf][2 * n + 1][2 * h], acc[pf][2 * n + 1][2 * h + 1], xc[0], xc[1], zc[0], zc[1], a.inv_mix_scale);
            }
            if constexpr (EPI == EPI_STORE && SILU) {
                silu_pair_to(a0, a1, acc[pf][2 * n][2 * h], acc[pf][2 * n][2 * h + 1]);
                silu_pair_to(b0, b1, acc[pf][2 * n + 1][2 * h], acc[pf][2 * n + 1][2 * h + 1]);
            }
            pa[h] = pack_pair<TT>(f32x2{a0, a1});
            pb[h] = pack_pair<TT>(f32x2{b0, b1});
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const auto sw = __builtin_amdgcn_permlane16_swap(pa[h], pb[h], false, false);
            o[h] = sw[0];
            o[2 + h] = sw[1];
        }
    };
    auto epi_store = [&](auto e_tag, const u32x4& o) __attribute__((always_inline)) {
        constexpr int E = decltype(e_tag)::value;
        constexpr int pf = E / NT, n = E % NT;
        bool inside;
        uint32_t off;
        if constexpr (GG::straddle(pf)) {
            inside = e_y + e_sdy < a.H && e_c + e_sdx < a.W;
            off = e_pix + e_soff + eoff[n];
        } else {
            inside = e_y + GG::dy(pf) < a.H && e_c + GG::dx(pf) < a.W;
            if constexpr (EPI == EPI_D2S) off = e_pix + (uint32_t)(2 * GG::dy(pf)) * (uint32_t)a.Wout * 16u + (uint32_t)(2 * GG::dx(pf)) * 16u + eoff[n];
            else off = e_pix + (uint32_t)GG::dy(pf) * (uint32_t)a.W * 16u + (uint32_t)GG::dx(pf) * 16u + eoff[n];
        }
        if (!inside || eoff[n] == 0xffffffffu) off = 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, (int)off, 0, 0);
    };
    // ---- FUSE (conv2 + AdaptiveResidualMix, model.py:826-839), one pixel fragment pf of the finished tile in three parts:
    //   A: x (the block input, a.in1) arrives in ACCUMULATOR layout, 8 bytes per channel fragment and lane; z is rounded to the
    //      storage type and packed into MFMA B operands (two 16-channel accumulator fragments = one 32-wide K step);
    //   C: gate beta = Wx.x + Wz.z on the MFMA: both halves of the gate weights are packed in accumulator-row order
    //      (PackArgs::frag16 = 2), so a pair of x fragments IS a B operand too -- x is fetched once; the 36 KB of gate weights
    //      stay in LDS for the whole launch;
    //   D: blend x + sigmoid(alpha) sigmoid(beta) (z - x) into the accumulator registers, then the three entries as usual.
    // The arithmetic is conv3s_kernel<.., FUSE>'s, operation for operation, EXCEPT the summation order of the x half of the gate inside a
    // 32-wide K step (accumulator-row order here, plane order there): equal to <= 1 ulp of the output, >= 98 % bit-equal
    // (tests/test_conv3r_gpu.py); which of the two runs therefore depends on channel counts only, never on H or W (mz_host.cpp).
    auto fuse_x = [&](auto pf_tag) __attribute__((always_inline)) {  // request x of pixel fragment pf
        constexpr int pf = decltype(pf_tag)::value;
        const bool inside = e_y + GG::dy(pf) < a.H && e_c + GG::dx(pf) < a.W;
        const long long plane_o = (long long)a.H * a.W * 16;
        uint32_t off = e_pix + (uint32_t)GG::dy(pf) * (uint32_t)a.W * 16u + (uint32_t)GG::dx(pf) * 16u + x_lane;
        if (!inside) off = 0xffffffffu;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const uint32_t o = inside ? off + (uint32_t)(2 * nf) * (uint32_t)plane_o : 0xffffffffu;  // planes >= p1 fall out of range: zeros
            // Inline asm (as conv3t_kernel): hipcc's waitcnt pass must not see these loads.  It cannot count the conditionally issued DMA
            // pieces that follow them and waited with vmcnt(0) in front of the gate step -- for the NEXT pixel fragment's loads, issued a
            // few hundred cycles earlier, whose lines come out of HBM.  The request is a whole chunk older than its use: the closing
            // vmcnt(0) of the gate step it precedes covers it; x_landed() marks the spot from which the values may be used.
            // The destination is the ring element itself: a copy behind the asm would read the register before the data arrives.
            u32x2_& dst = f_xq[pf & 1][nf];
            const u32x4& xr = xrsrc;  // (named: an asm operand alone does not make a generic lambda capture it)
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(o), "s"(xr) : "memory");
        }
    };
    auto x_landed = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            u32x2_& r = f_xq[pf & 1][nf];  // (named first: an asm operand alone does not make a generic lambda capture the array)
            asm volatile("" : "+v"(r));
        }
    };
    auto fuse_z = [&](auto pf_tag) __attribute__((always_inline)) {  // z of pixel fragment pf as B operands
        constexpr int pf = decltype(pf_tag)::value;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const f32x4 za = acc[pf][2 * m], zc = acc[pf][2 * m + 1];
            u32x4 q;
            q[0] = pack2r<TT>(za[0], za[1]); q[1] = pack2r<TT>(za[2], za[3]);
            q[2] = pack2r<TT>(zc[0], zc[1]); q[3] = pack2r<TT>(zc[2], zc[3]);
            asm volatile("" : "+v"(q));  // opaque: no pack -> unpack forwarding that would keep the floats alive
            f_zb[m] = q;
        }
    };
    // part A of chunk pf: x of the NEXT pixel fragment (fragment 0's: see loader_step()), z of this one
    auto fuse_a = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
        if constexpr (pf + 1 < NPF) fuse_x(ic<(pf + 1 < NPF ? pf + 1 : 0)>{});
        fuse_z(pf_tag);
    };
    auto fuse_c = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
        const int lane_ = lane_now();
        const uint32_t mix_lane = lds_base + S::MIX_BASE + lane_ * 16;
        u32x4 xb[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) xb[m] = u32x4{f_xq[pf & 1][2 * m][0], f_xq[pf & 1][2 * m][1], f_xq[pf & 1][2 * m + 1][0], f_xq[pf & 1][2 * m + 1][1]};
        u32x4 wa[NT], wb[NT];
        gate_reads<0, 0>(wa, mix_lane);
        gate_halves<TT, 0>(acc[pf], xb, f_zb, wa, wb, mix_lane);
        // MFMA result -> VALU read is a software hazard (8 passes: 11 wait states) that hipcc does not see into inline asm for: the blend
        // (blend_pair_to) reads beta from inline-asm chains.  In the tile loop a whole step lies between gate and blend; in the final
        // epilogue they follow each other directly (conv3t_kernel met the stale read there).
        asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto entry_whole = [&](auto e_tag) __attribute__((always_inline)) {
        if constexpr (decltype(e_tag)::value < NE) {
            u32x4 o;
            entry_words(e_tag, o);
            epi_store(e_tag, o);
        }
    };

    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // fragment stream of a tile's first groups: tap 0 of chunk 0 and the weight pairs of groups 0 and 1
    auto prime = [&](int wslot, int aslot) __attribute__((always_inline)) {
        const int lane_ = lane_now();
        const int g = lane_ >> 4, c = lane_ & 15;
        Bases pb;
        pb.a_cur = pb.a_nxt = lds_base + aslot * A_SLOT + g * A_PLANE + ((2 * wq) * ROWW + c) * 16;
        pb.a2_cur = pb.a2_nxt = pb.a_cur + (uint32_t)((c >= 8 ? ROWW - 8 : 32) * 16);  // (GEO 1: the straddling fragment)
        pb.b_cur = pb.b_nxt = 0;
        const uint32_t bb = lds_base + B_BASE + wslot * B_SLOT + lane_ * 16;
        f.x[0][0] = read_x<GEO, 0, 0>(pb, false);
        f.x[0][1] = read_x<GEO, 0, 1>(pb, false);
        f.x[0][2] = read_x<GEO, 0, 2>(pb, false);
        f.x[0][3] = read_x<GEO, 0, 3>(pb, false);
        f.x[0][4] = read_x<GEO, 0, 4>(pb, false);
        if constexpr (NPF > 5) f.x[0][NPF - 1] = read_x<GEO, 0, NPF - 1>(pb, false);
        f.w[0][0] = lds_read128<0 * 1024>(bb);
        f.w[0][1] = lds_read128<1 * 1024>(bb);
        f.w[1][0] = lds_read128<2 * 1024>(bb);
        f.w[1][1] = lds_read128<3 * 1024>(bb);
        wait_wx<0>(f.w[0][0], f.w[0][1], f.x[0]);
        wait_w<0>(f.w[1][0], f.w[1][1]);
    };

    // One step of the loader role while the partner team computes chunk k of tile tA.
    //   ES = first of this step's entries of the finished tile's epilogue, EN = how many (0..3)
    //   last: the tile's last chunk (its loads already belong to the next tile tB);  pre: the chunk before it
    const char* wA = nullptr;
    const char* wB = nullptr;
    bool okB = false;
    auto loader_step = [&](auto wk_tag, auto es_tag, auto en_tag, auto sg_tag, auto last_tag, int k) __attribute__((always_inline)) {
        constexpr int WK = decltype(wk_tag)::value;  // 1: entries [ES, ES + EN); 2 / 3 / 4: FUSE part A / C / D of EN pixel fragments from ES
        constexpr int ES = decltype(es_tag)::value, EN = WK == 0 ? 0 : decltype(en_tag)::value, sg = decltype(sg_tag)::value;
        constexpr bool last = decltype(last_tag)::value != 0;
        [[maybe_unused]] constexpr int rs_c = 1 + (EN > 0 ? 2 : 0) + (sg == 0 ? 0 : 1);
        static_assert(EN <= 3, "three output registers");
        // VMEM instructions this step issues BEHIND its DMA: stores of the entries, or FUSE's x loads
        constexpr int VM_AFTER0 = WK == 1 ? EN : (WK == 2 ? (ES + 1 < NPF ? NF : 0) : (WK == 4 ? EN * NT : 0));
        constexpr int VM_AFTER = VM_AFTER0 + (NSEG == 3 && sg == 0 ? 8 : 0);  // (HALO_LATE, below: the image's pieces stay in flight for a step)
        RS_BEGIN();
        // x of a pixel fragment is requested in part A of the chunk before its gate step (covered by that chunk's closing vmcnt(0)); pixel
        // fragment 0's at the head of the tile's first step, AHEAD of the step's DMA: that step's closing wait covers it too
        if constexpr (WK == 2 && ES == 0) fuse_x(ic<0>{});
        if constexpr (WK == 3) x_landed(ic<ES>{});
        // ---- this step's DMA: in a chunk's first step the next chunk's halo image, and weight segment (k, sg) + 2 steps.
        //      NSEG = 2: the image first (its data comes from HBM and takes longest) and landed by the end of this step -- the compute role
        //      reads its first fragments in the chunk's SECOND step.  NSEG = 3 (HALO_LATE): those reads come in the third step, so the
        //      image goes out BEHIND the weights and this step's closing wait leaves its eight pieces in flight (the next step's covers
        //      them): a step more for the lines to come out of HBM.  For that count to be exact the pieces are issued on every path: where
        //      no tile follows they re-fetch the current tile's first image into the free slot. ----
        constexpr bool HALO_LATE = NSEG == 3;
        auto halo_image = [&]() __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t h_rsrc = halo_rsrc(last ? 0 : k + 1);
            // (opaque: hipcc would otherwise keep the eight piece addresses of a slot alive from one chunk to the next but one as
            // spilled SGPRs -- a v_writelane / v_readlane pair each, vector instructions the helper role is short of)
            uint32_t h_off = (uint32_t)(us ^ 1) * (uint32_t)A_SLOT;
            asm volatile("" : "+s"(h_off));
            char* const h_dst = smem + h_off;
            const bool rag = RAG && !last && k + 2 == nchunks;  // the tile's last chunk is the ragged one
            halo_piece(ic<0>{}, h_rsrc, h_dst, rag); halo_piece(ic<1>{}, h_rsrc, h_dst, rag); halo_piece(ic<2>{}, h_rsrc, h_dst, rag);
            halo_piece(ic<3>{}, h_rsrc, h_dst, rag); halo_piece(ic<4>{}, h_rsrc, h_dst, rag); halo_piece(ic<5>{}, h_rsrc, h_dst, rag);
            halo_piece(ic<6>{}, h_rsrc, h_dst, rag); halo_piece(ic<7>{}, h_rsrc, h_dst, rag);
        };
        if constexpr (sg == 0 && !HALO_LATE) {
            if (!last || okB) halo_image();
        }
        {
            constexpr bool same = sg + 2 < NSEG;
            constexpr int qs = same ? sg + 2 : sg + 2 - NSEG;
            const char* wsrc;
            bool w_ok = true;
            if constexpr (same) wsrc = wA + (size_t)k * chunk_bytes;
            else if constexpr (!last) wsrc = wA + (size_t)(k + 1) * chunk_bytes;
            else { wsrc = wB; w_ok = okB; }
            wsrc += (size_t)(2 * S::start(qs)) * 1024;
            const int w_pieces = w_ok ? wseg_count(S::pieces(qs)) : 0;
            char* const w_dst = smem + B_BASE + (hs >= 1 ? hs - 1 : 2) * B_SLOT;  // slot (hs + 2) % 3
            const uint32_t lo = (uint32_t)lane_now() * 16u;
            wseg_piece(ic<0>{}, wsrc, w_pieces, w_dst, lo); wseg_piece(ic<1>{}, wsrc, w_pieces, w_dst, lo); wseg_piece(ic<2>{}, wsrc, w_pieces, w_dst, lo);
            wseg_piece(ic<3>{}, wsrc, w_pieces, w_dst, lo); wseg_piece(ic<4>{}, wsrc, w_pieces, w_dst, lo);
            if constexpr (WP > 5) wseg_piece(ic<5>{}, wsrc, w_pieces, w_dst, lo);
            if constexpr (WP > 6) wseg_piece(ic<6>{}, wsrc, w_pieces, w_dst, lo);
            static_assert(WP <= 7, "seven weight pieces per wave and step");
        }
        if constexpr (sg == 0 && HALO_LATE) halo_image();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        RS_LAP(4 * rs_c);
        // ---- this step's share of the finished tile's epilogue, behind the DMA issue: the arithmetic runs while the loads are in
        //      flight, the stores go out behind the step's last DMA piece (vmcnt: see the head of this file) ----
        if constexpr (WK == 1) {
            u32x4 o0, o1, o2;
            if constexpr (EN > 0) entry_words(ic<ES>{}, o0);
            if constexpr (EN > 1) entry_words(ic<ES + 1>{}, o1);
            if constexpr (EN > 2) entry_words(ic<ES + 2>{}, o2);
            if constexpr (EN > 0) epi_store(ic<ES>{}, o0);
            if constexpr (EN > 1) epi_store(ic<ES + 1>{}, o1);
            if constexpr (EN > 2) epi_store(ic<ES + 2>{}, o2);
        } else if constexpr (WK == 2) {
            static_assert(WK != 2 || EN == 1, "one pixel fragment per chunk");
            fuse_a(ic<ES>{});
        } else if constexpr (WK == 3) {
            fuse_c(ic<ES>{});
        } else if constexpr (WK == 4) {
            // (the blend is part of the entries: entry_words())
            entry_whole(ic<3 * ES>{}); entry_whole(ic<3 * ES + 1>{}); entry_whole(ic<3 * ES + 2>{});
        }
        RS_FENCE();
        RS_LAP(4 * rs_c + 1);
        if constexpr (sg == NSEG - 1) {
            if constexpr (last) {
                // the epilogue is complete: clear the accumulators and prime the fragment stream for the next tile.  Unconditional
                // on every path into the compute role, so that the fragment registers are DEAD throughout the loader role.
                if constexpr (FUSE) zero_acc();  // (with the tile's first tap writing the accumulators instead, hipcc spills in this variant)
                prime(next3(hs), us ^ 1);
            } else {
                if (k + 2 == nchunks && okB) {  // the next step requests tB's first halo image
                    set_load_tile(eB);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        RS_LAP(sg == NSEG - 1 ? (last ? 26 : 25) : 4 * rs_c + 1);  // (diagnostic build: the tile hand-over work on its own counters)
        // the DMA has landed once at most this step's stores (issued behind it) are outstanding
        wait_vmcnt<VM_AFTER>();
        RS_LAP(4 * rs_c + 2);
        __builtin_amdgcn_s_barrier();
        RS_LAP(4 * rs_c + 3);
        RS_COUNT(20 + rs_c);
        hs = next3(hs);
        if constexpr (sg == NSEG - 1) us ^= 1;  // the chunk is complete
    };
    // chunk iteration k of the plain variants; E0 = its first entry, N0 / N1 = entries of its first / second step
    auto loader_chunk = [&](auto e0_tag, auto n0_tag, auto n1_tag, auto last_tag, int k) __attribute__((always_inline)) {
        // (clamped to the NE entries a tile has: 18, or 15 with the five-fragment geometry)
        constexpr int E0 = decltype(e0_tag)::value < NE ? decltype(e0_tag)::value : NE;
        constexpr int N0 = decltype(n0_tag)::value < NE - E0 ? decltype(n0_tag)::value : NE - E0;
        constexpr int N1 = decltype(n1_tag)::value < NE - E0 - N0 ? decltype(n1_tag)::value : NE - E0 - N0;
        if constexpr (NSEG == 2) {
            loader_step(ic<(N0 > 0)>{}, ic<E0>{}, ic<N0>{}, ic<0>{}, last_tag, k);
            loader_step(ic<(N1 > 0)>{}, ic<E0 + N0>{}, ic<N1>{}, ic<1>{}, last_tag, k);
        } else {  // three steps: the first (halo image) takes N0, the other two share N1
            constexpr int N1a = N1 / 2, N1b = N1 - N1a;
            loader_step(ic<(N0 > 0)>{}, ic<E0>{}, ic<N0>{}, ic<0>{}, last_tag, k);
            loader_step(ic<(N1a > 0)>{}, ic<E0 + N0>{}, ic<N1a>{}, ic<1>{}, last_tag, k);
            loader_step(ic<(N1b > 0)>{}, ic<E0 + N0 + N1a>{}, ic<N1b>{}, ic<2>{}, last_tag, k);
        }
    };
    // chunk iteration k of the fused variant: parts A, C, D of G pixel fragments from P0 in its three steps
    auto fuse_chunk = [&](auto p0_tag, auto g_tag, auto last_tag, int k) __attribute__((always_inline)) {
        static_assert(!FUSE || NSEG == 3, "the fused variant runs three steps per chunk");
        loader_step(ic<2>{}, p0_tag, g_tag, ic<0>{}, last_tag, k);
        loader_step(ic<3>{}, p0_tag, g_tag, ic<1>{}, last_tag, k);
        loader_step(ic<4>{}, p0_tag, g_tag, ic<NSEG - 1>{}, last_tag, k);
    };
    auto plain_chunks = [&](int k0) __attribute__((always_inline)) {  // chunks k0 .. nchunks - 1 without epilogue work
        for (int k = k0; k + 1 < nchunks; ++k) loader_chunk(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, k);
        loader_chunk(ic<0>{}, ic<0>{}, ic<0>{}, ic<1>{}, nchunks - 1);
    };
    auto loader_phase = [&](auto epi_tag) __attribute__((always_inline)) {
        constexpr bool DO_EPI = decltype(epi_tag)::value != 0;
        wA = wsrc_of(tile_nt(eA));
        okB = a_pos + step < cnt;
        wB = wsrc_of(tile_nt(okB ? eB : eA));
        // the two tiles behind eB: eA / eB of this team's NEXT helper phase (two phases on), requested now, taken over at the end of this phase
        const TileE eA2 = tile_at(a_pos + 2 * step), eB2 = tile_at(a_pos + 3 * step);
        if constexpr (DO_EPI) {
            RS_BEGIN();
            set_load_tile(eA);
            epi_setup();
            RS_FENCE();
            RS_LAP(27);
            if constexpr (RAG) {
                // two chunks, six steps, three entries each
                loader_chunk(ic<0>{}, ic<3>{}, ic<6>{}, ic<0>{}, 0);
                loader_chunk(ic<9>{}, ic<3>{}, ic<6>{}, ic<1>{}, 1);
            } else if constexpr (FUSE) {
                // one pixel fragment per chunk: the tile has at least six chunks (C = 96: Cin = 192; the host guards)
                fuse_chunk(ic<0>{}, ic<1>{}, ic<0>{}, 0);
                fuse_chunk(ic<1>{}, ic<1>{}, ic<0>{}, 1);
                fuse_chunk(ic<2>{}, ic<1>{}, ic<0>{}, 2);
                fuse_chunk(ic<3>{}, ic<1>{}, ic<0>{}, 3);
                fuse_chunk(ic<4>{}, ic<1>{}, ic<0>{}, 4);
                if (nchunks == 6) {
                    fuse_chunk(ic<5>{}, ic<1>{}, ic<1>{}, 5);
                } else {
                    fuse_chunk(ic<5>{}, ic<1>{}, ic<0>{}, 5);
                    plain_chunks(6);
                }
            } else if (nchunks >= 6) {
                // The 18 entries are spread over the tile's first six chunks where it has six (1 + 2 per chunk: the first step also
                // carries the halo image; with three steps per chunk 1 + 1 + 1), else over its first three.
                loader_chunk(ic<0>{}, ic<1>{}, ic<2>{}, ic<0>{}, 0);
                loader_chunk(ic<3>{}, ic<1>{}, ic<2>{}, ic<0>{}, 1);
                loader_chunk(ic<6>{}, ic<1>{}, ic<2>{}, ic<0>{}, 2);
                loader_chunk(ic<9>{}, ic<1>{}, ic<2>{}, ic<0>{}, 3);
                loader_chunk(ic<12>{}, ic<1>{}, ic<2>{}, ic<0>{}, 4);
                if (nchunks == 6) {
                    loader_chunk(ic<15>{}, ic<1>{}, ic<2>{}, ic<1>{}, 5);
                } else {
                    loader_chunk(ic<15>{}, ic<1>{}, ic<2>{}, ic<0>{}, 5);
                    plain_chunks(6);
                }
            } else {
                constexpr int F3 = NSEG == 3 ? 2 : 3;  // entries of a chunk's first step (three steps: 2 + 2 + 2)
                loader_chunk(ic<0>{}, ic<F3>{}, ic<6 - F3>{}, ic<0>{}, 0);
                loader_chunk(ic<6>{}, ic<F3>{}, ic<6 - F3>{}, ic<0>{}, 1);
                if (nchunks == 3) {
                    loader_chunk(ic<12>{}, ic<F3>{}, ic<6 - F3>{}, ic<1>{}, 2);
                } else {
                    loader_chunk(ic<12>{}, ic<F3>{}, ic<6 - F3>{}, ic<0>{}, 2);
                    plain_chunks(3);
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
        const uint32_t a_lane = lds_base + g * A_PLANE + ((2 * wq) * ROWW + c) * 16;
        Bases bs;
        bs.a_cur = a_lane + us * A_SLOT;
        bs.a_nxt = a_lane + (us ^ 1) * A_SLOT;
        const uint32_t a_str = (uint32_t)((c >= 8 ? ROWW - 8 : 32) * 16);  // GEO 1: the straddling fragment's displacement
        bs.a2_cur = bs.a_cur + a_str;
        bs.a2_nxt = bs.a_nxt + a_str;
        const uint32_t bl = lds_base + B_BASE + lane_ * 16;
        uint32_t b0 = bl + hs * B_SLOT, b1 = bl + next3(hs) * B_SLOT, b2 = bl + next3(next3(hs)) * B_SLOT;
        bs.b_cur = b0;
        bs.b_nxt = b1;
        auto step_tail = [&]() __attribute__((always_inline)) {  // barrier; rotate the weight slots: cur <- nxt <- nn <- cur
            __builtin_amdgcn_s_barrier();
            const uint32_t u_ = b0; b0 = b1; b1 = b2; b2 = u_;
            bs.b_cur = b0; bs.b_nxt = b1;
        };
        auto chunk = [&](auto xp_tag, bool first) __attribute__((always_inline)) {
            constexpr int XP = decltype(xp_tag)::value;
            groups<TT, NSEG, GEO, S::start(0), S::start(1), XP>(acc, f, bs, first);
            step_tail();
            groups<TT, NSEG, GEO, S::start(1), S::start(2), XP>(acc, f, bs, false);
            step_tail();
            if constexpr (NSEG == 3) {
                groups<TT, NSEG, GEO, S::start(2), S::start(3), XP>(acc, f, bs, false);
                step_tail();
            }
            const uint32_t v_ = bs.a_cur; bs.a_cur = bs.a_nxt; bs.a_nxt = v_;
            if constexpr (GEO == 1) { const uint32_t v2_ = bs.a2_cur; bs.a2_cur = bs.a2_nxt; bs.a2_nxt = v2_; }
        };
        // The last groups of a tile request fragments of a "next chunk" this wave will not compute: they are waited for behind
        // the loop (the registers are reused by the loader role).
        // The tap-0 parity XP alternates from chunk to chunk; the loop leaves after either body (an odd chunk count ends behind
        // the XP = 0 body).  Both exits see the accumulators in the registers the loop carries them in: no copies, no spills
        // (a peeled chunk in front of or behind a pair loop made hipcc rename them at the junction).
        RS_BEGIN();
        int kc = 0;
        for (;;) {
            chunk(ic<0>{}, !FUSE && kc == 0);  // (the fused variant clears its accumulators in the helper role: see there)
            if (++kc >= nchunks) break;
            chunk(ic<1>{}, false);
            if (++kc >= nchunks) break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.x[0][0]), "+v"(f.x[0][1]), "+v"(f.x[0][2]), "+v"(f.x[0][3]), "+v"(f.x[0][4]), "+v"(f.x[0][NPF - 1]),
                       "+v"(f.x[1][0]), "+v"(f.x[1][1]), "+v"(f.x[1][2]), "+v"(f.x[1][3]), "+v"(f.x[1][4]), "+v"(f.x[1][NPF - 1])
                     :
                     : "memory");
        asm volatile("" : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.w[2][0]), "+v"(f.w[2][1]));
        RS_LAP(0);
        RS_COUNT(1);
        // slot counters after NSEG * nchunks steps
        const int adv = (NSEG * nchunks) % 3;
        hs = hs + adv >= 3 ? hs + adv - 3 : hs + adv;
        us ^= nchunks & 1;
    };
    auto final_fuse_pf = [&](auto pf_tag) __attribute__((always_inline)) {
        constexpr int pf = decltype(pf_tag)::value;
        fuse_x(pf_tag);
        fuse_z(pf_tag);
        wait_vmcnt<0>();  // (no partner, no DMA: the plain sequence)
        x_landed(pf_tag);
        fuse_c(pf_tag);
        entry_whole(ic<3 * pf>{}); entry_whole(ic<3 * pf + 1>{}); entry_whole(ic<3 * pf + 2>{});
    };
    auto final_epilogue = [&]() __attribute__((always_inline)) {
        epi_setup();
        if constexpr (FUSE) {
            final_fuse_pf(ic<0>{}); final_fuse_pf(ic<1>{}); final_fuse_pf(ic<2>{}); final_fuse_pf(ic<3>{}); final_fuse_pf(ic<4>{}); final_fuse_pf(ic<5>{});
        } else {
            entry_whole(ic<0>{}); entry_whole(ic<1>{}); entry_whole(ic<2>{}); entry_whole(ic<3>{}); entry_whole(ic<4>{}); entry_whole(ic<5>{});
            entry_whole(ic<6>{}); entry_whole(ic<7>{}); entry_whole(ic<8>{}); entry_whole(ic<9>{}); entry_whole(ic<10>{}); entry_whole(ic<11>{});
            entry_whole(ic<12>{}); entry_whole(ic<13>{}); entry_whole(ic<14>{}); entry_whole(ic<15>{}); entry_whole(ic<16>{}); entry_whole(ic<17>{});
        }
    };

    // ------------------------------------------------------------------------------------------------
    if (team == 1) {
        // prologue: chunk 0 of the first tile (halo image + the first two weight segments), published by B_0
        eA = tile_at(a_pos);
        eB = tile_at(a_pos + step);
        eD = eA;  // (unused: the first helper phase has no epilogue)
        wA = wsrc_of(tile_nt(eA));
        set_load_tile(eA);
        {
            const __amdgpu_buffer_rsrc_t r0 = halo_rsrc(0);
            halo_piece(ic<0>{}, r0, smem); halo_piece(ic<1>{}, r0, smem); halo_piece(ic<2>{}, r0, smem); halo_piece(ic<3>{}, r0, smem);
            halo_piece(ic<4>{}, r0, smem); halo_piece(ic<5>{}, r0, smem); halo_piece(ic<6>{}, r0, smem); halo_piece(ic<7>{}, r0, smem);
            const char* s1 = wA + (size_t)(2 * S::start(1)) * 1024;
            const uint32_t lo = (uint32_t)lane_now() * 16u;
            char* d0 = smem + B_BASE;
            char* d1 = smem + B_BASE + B_SLOT;
            wseg_piece(ic<0>{}, wA, wseg_count(S::pieces(0)), d0, lo); wseg_piece(ic<1>{}, wA, wseg_count(S::pieces(0)), d0, lo); wseg_piece(ic<2>{}, wA, wseg_count(S::pieces(0)), d0, lo);
            wseg_piece(ic<3>{}, wA, wseg_count(S::pieces(0)), d0, lo); wseg_piece(ic<4>{}, wA, wseg_count(S::pieces(0)), d0, lo); wseg_piece(ic<5>{}, wA, wseg_count(S::pieces(0)), d0, lo);
            wseg_piece(ic<6>{}, wA, wseg_count(S::pieces(0)), d0, lo);
            wseg_piece(ic<0>{}, s1, wseg_count(S::pieces(1)), d1, lo); wseg_piece(ic<1>{}, s1, wseg_count(S::pieces(1)), d1, lo); wseg_piece(ic<2>{}, s1, wseg_count(S::pieces(1)), d1, lo);
            wseg_piece(ic<3>{}, s1, wseg_count(S::pieces(1)), d1, lo); wseg_piece(ic<4>{}, s1, wseg_count(S::pieces(1)), d1, lo); wseg_piece(ic<5>{}, s1, wseg_count(S::pieces(1)), d1, lo);
            wseg_piece(ic<6>{}, s1, wseg_count(S::pieces(1)), d1, lo);
        }
        if constexpr (FUSE) {  // the gate weights, resident for the whole launch: 36 pieces, 9 per wave
            const int lane_ = lane_now();
            const char* msrc = (const char*)a.wmix16 + (uint32_t)lane_ * 16u;
#pragma unroll
            for (int i = 0; i < MIX_PIECES / 4; ++i) glds16(msrc + (size_t)(wq + 4 * i) * 1024, smem + S::MIX_BASE + (wq + 4 * i) * 1024);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // B_0
        if constexpr (FUSE) zero_acc();
        loader_phase(ic<0>{});
        advance();
        if (a_pos >= cnt) { RS_DUMP(); return; }
    } else {
        // its first helper phase (after tile a_pos) works on the tiles behind it
        eD = tile_at(a_pos);
        eA = tile_at(a_pos + step);
        eB = tile_at(a_pos + 2 * step);
        asm volatile("" ::"s"(eD.yx), "s"(eD.bn), "s"(eA.yx), "s"(eA.bn), "s"(eB.yx), "s"(eB.bn));
        __builtin_amdgcn_s_barrier();  // B_0
        if constexpr (FUSE) zero_acc();
        prime(0, 0);
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
