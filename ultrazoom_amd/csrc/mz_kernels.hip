// gfx950 (MI355X / CDNA4) kernels of the MewZoom upscale path.
//
// Everything here is written for 64-wide wavefronts and the CDNA4 matrix cores:
//   v_mfma_f32_32x32x16_{bf16,f16}  (8 K-elements per lane)  for the 16-bit modes
//   v_mfma_f32_32x32x2_f32          (exact f32)              for the f32 verification mode
//
// Convolution = implicit GEMM computed TRANSPOSED: D[n][pixel] = sum_k W[n][k] * X[pixel][k].
// The weight fragment is the MFMA "A" operand and the activation fragment the "B" operand, so an
// accumulator register quad holds 4 CONSECUTIVE channels of one pixel (rows of a 32x32 tile are
// (reg&3) + 8*(reg>>2) + 4*(lane>>5), the column = lane&31 = pixel): NHWC packing in the epilogue
// needs no cross-lane traffic.
//
// Activation tensors in HBM are "plane-major": [B][P][H][W][16 bytes], a plane being 16 bytes of
// consecutive channels (8 bf16/f16 or 4 f32 channels); P = padded_channels * sizeof / 16.  A 3x3 halo
// row or a run of pixels of one plane is then CONTIGUOUS in memory, so a global_load_lds instruction
// touches a handful of cache lines instead of one per lane.
//
// LDS image of one K-stage (all sizes in bytes; a "chunk" is 32 bytes of channels per pixel, i.e.
// 16 bf16/f16 channels or 8 f32 channels, split in two 16-byte "planes" = the two lane halves):
//   A (activations)  CONV3: [plane 2][pixel 352 (10 rows x 34 cols halo, padded)][16]   = 11264
//                    GEMM1: [chunk S][plane 2][pixel 256][16]                           = S*8192
//   B (weights)      [chunk S][tap][nt][lane 64][16]  — already in fragment order in HBM, so a
//                    stage is ONE contiguous run of TAPS*S*NT KiB copied by global_load_lds.
// Both images are lane-linear, which is what global_load_lds (LDS-DMA) requires, and every
// ds_read_b128 of a fragment covers contiguous 512-byte runs per half-wave: bank-conflict free.
#include "mz_device.h"

namespace mz {


// ================================================================================================
// implicit-GEMM convolution
// ================================================================================================
// Staging goes through LDS-DMA (global_load_lds).  -DMZ_REG_STAGING builds the same kernels with plain
// global loads + ds_write instead (a debugging aid: both variants must produce identical bits).
#ifdef MZ_REG_STAGING
static constexpr bool kGlds = false;
#else
static constexpr bool kGlds = true;
#endif
#ifndef MZ_GEMM1_S
#define MZ_GEMM1_S 2  // K-chunks per stage of the 1x1 kernel: 44 KiB of LDS -> 3 workgroups per CU (measured best of 1..4)
#endif
template <int MODE> struct Geo;
template <> struct Geo<MODE_CONV3> {  // 4 waves, 8 x 32 pixels
    static constexpr int TAPS = 9;
    static constexpr int S = 1;            // chunks per stage
    static constexpr int ROWW = 34;        // halo row width
    static constexpr int A_ENT = 704;      // 16-byte entries per A image (2 planes x 352)
    static constexpr int PLANE = 352 * 16;
    static constexpr int MF_STRIDE = 34 * 16;  // second M fragment = next tile row
};
template <> struct Geo<MODE_GEMM1> {
    static constexpr int TAPS = 1;
    static constexpr int S = MZ_GEMM1_S;
    static constexpr int ROWW = 0;
    static constexpr int A_ENT = MZ_GEMM1_S * 512;
    static constexpr int PLANE = 256 * 16;
    static constexpr int MF_STRIDE = 32 * 16;
};
template <> struct Geo<MODE_C3W16> {  // 8 compute waves, 16 x 32 pixels: wave w owns rows 2w, 2w+1
    static constexpr int TAPS = 9;
    static constexpr int S = 1;
    static constexpr int TH = 16, TW = 32;
    static constexpr int ROWW = 34;
    static constexpr int NPIX = 18 * 34;   // 612 halo pixels
    static constexpr int PLANE_ENT = 640;  // padded so that 2 planes = a whole number of 64-entry DMA instructions
    static constexpr int A_ENT = 2 * PLANE_ENT;
    static constexpr int PLANE = PLANE_ENT * 16;
    static constexpr int MF_STRIDE = 34 * 16;
    static constexpr int ROW_PER_WAVE = 2;
};
template <> struct Geo<MODE_C3W8> {  // 8 compute waves, 8 x 64 pixels: wave w owns row w, fragments = its two halves
    static constexpr int TAPS = 9;
    static constexpr int S = 1;
    static constexpr int TH = 8, TW = 64;
    static constexpr int ROWW = 66;
    static constexpr int NPIX = 10 * 66;   // 660 halo pixels
    static constexpr int PLANE_ENT = 672;
    static constexpr int A_ENT = 2 * PLANE_ENT;
    static constexpr int PLANE = PLANE_ENT * 16;
    static constexpr int MF_STRIDE = 32 * 16;
    static constexpr int ROW_PER_WAVE = 1;
};

// One "item" = one (chunk-in-stage, filter tap) pair = one 32-byte K-chunk of matrix work:
// 2 + NT fragment reads (two pixel fragments, NT weight fragments) feeding 2 * NT MFMAs.
template <int NT, int MODE, int ITEM, int K> __device__ __forceinline__ void issue_read(Frags<NT>& f, uint32_t a_addr, uint32_t b_addr) {
    using G = Geo<MODE>;
    constexpr int s = ITEM / G::TAPS, tap = ITEM % G::TAPS;
    constexpr int aofs = (MODE != MODE_GEMM1) ? ((tap / 3) * G::ROWW + (tap % 3)) * 16 : s * 8192;
    if constexpr (K == 0) f.x0 = lds_read128<aofs>(a_addr);
    else if constexpr (K == 1) f.x1 = lds_read128<aofs + G::MF_STRIDE>(a_addr);
    else if constexpr (K < 2 + NT) f.w[K - 2] = lds_read128<(ITEM * NT + (K - 2)) * 1024>(b_addr);
}
template <int NT, int MODE, int ITEM> __device__ __forceinline__ void issue_reads(Frags<NT>& f, uint32_t a_addr, uint32_t b_addr) {
    issue_read<NT, MODE, ITEM, 0>(f, a_addr, b_addr);
    issue_read<NT, MODE, ITEM, 1>(f, a_addr, b_addr);
    issue_read<NT, MODE, ITEM, 2>(f, a_addr, b_addr);
    issue_read<NT, MODE, ITEM, 3>(f, a_addr, b_addr);
    issue_read<NT, MODE, ITEM, 4>(f, a_addr, b_addr);
    issue_read<NT, MODE, ITEM, 5>(f, a_addr, b_addr);
}
// MFMA step M of an item (M = 2*nt + mf), followed by two of the NEXT item's fragment reads: the reads issue in
// the shadow of the MFMA just issued (the matrix pipe accepts one 32x32x16 MFMA per 32 cycles), and all of them
// are in flight at least (2*NT - 3) MFMAs before the item's closing s_waitcnt.
template <class TT, int NT, int MODE, int ITEM, int NITEMS, int M>
__device__ __forceinline__ void mfma_steps(f32x16 (&acc)[2][NT], const Frags<NT>& cur, Frags<NT>& nxt, uint32_t a_addr,
                                           uint32_t b_addr) {
    if constexpr (M < 2 * NT) {
        mma<TT>(acc[M & 1][M >> 1], cur.w[M >> 1], (M & 1) ? cur.x1 : cur.x0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ITEM + 1 < NITEMS) {
            issue_read<NT, MODE, ITEM + 1, 2 * M>(nxt, a_addr, b_addr);
            issue_read<NT, MODE, ITEM + 1, 2 * M + 1>(nxt, a_addr, b_addr);
            if constexpr (M == 2 * NT - 1) {  // NT == 1: 3 reads, 2 MFMAs
                issue_read<NT, MODE, ITEM + 1, 2 * M + 2>(nxt, a_addr, b_addr);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_steps<TT, NT, MODE, ITEM, NITEMS, M + 1>(acc, cur, nxt, a_addr, b_addr);
    }
}
// cur holds the (already waited-for) fragments of ITEM; nxt receives those of ITEM+1 while ITEM's MFMAs run.
template <class TT, int NT, int MODE, int ITEM, int NITEMS>
__device__ __forceinline__ void run_items(f32x16 (&acc)[2][NT], Frags<NT>& cur, Frags<NT>& nxt, uint32_t a_addr,
                                          uint32_t b_addr) {
    if constexpr (ITEM < NITEMS) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps<TT, NT, MODE, ITEM, NITEMS, 0>(acc, cur, nxt, a_addr, b_addr);
        if constexpr (ITEM + 1 < NITEMS) wait_frags<NT>(nxt);
        run_items<TT, NT, MODE, ITEM + 1, NITEMS>(acc, nxt, cur, a_addr, b_addr);
    }
}


__device__ __forceinline__ bool map_tile(const ConvArgs& a, int& mtile, int& ntile) {
    const int nblk = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7, pos = bid >> 3;
    const int L = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + pos;
    const int gsz = a.gm * a.gn;
    const int group = fdiv(L, gsz, a.inv_gsz), within = L - group * gsz;
    const int gi_n = fdiv(group, a.groups_m, a.inv_groups_m), gi_m = group - gi_n * a.groups_m;
    const int mi = fdiv(within, a.gn, a.inv_gn), ni = within - mi * a.gn;
    mtile = gi_m * a.gm + mi;
    ntile = gi_n * a.gn + ni;
    return mtile < a.mtiles && ntile < a.ntiles;
}

// PixelShuffle(2) + bicubic skip + residual add (+ clamp) -> NCHW image (reference model.py:926-930, 156, 162, 177).
// U8: both images are uint8 (a compile-time switch: a per-load branch would serialise the 48 taps of every lane).
constexpr int kFinalWinBytes = 15 * 64 * 4;  // per wave: the bicubic window of a 32-pixel fragment (5 rows x 3 channels x 64 columns x 4 bytes)
template <class TT, int NT, bool U8>
__device__ __forceinline__ void final_epilogue(const ConvArgs& a, f32x16 (&acc)[2][NT], char* ep, char* win, int lane, int b,
                                               const int (&ey)[2], const int (&ex)[2]) {
    constexpr int SZ = TT::SZ;
    const int h = lane >> 5, r = lane & 31;
    constexpr int ROWF = 80;  // 16 floats + 16 bytes pad
    const long long plane_i = (long long)a.Hi * a.Wi;
    const long long plane_o = (long long)a.Hout * a.Wout;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        const int y = ey[mf];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mf][0][4 * q + j];
            *(float4*)(ep + r * ROWF + (8 * q + 4 * h) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_wave_barrier();
        const int px = lane >> 1, jj = lane & 1;
        const int x = ex[mf] + px;
        const int X = 2 * x + jj;
        // the conv results of this lane's six outputs, read BEFORE any image load is issued: hipcc drains vmcnt to 0 in
        // front of every LDS read of a kernel that uses LDS-DMA, which would serialise the 96 image loads below
        float zres[2][3];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int c = 0; c < 3; ++c) zres[i2][c] = *(const float*)(ep + px * ROWF + ((2 * i2 + jj) * 4 + c) * 4);
        if (y < a.H) {  // (wave-uniform)
            // The bicubic skip reads a 4 x 4 window of the input image per output pixel.  The 64 output columns x 2 output rows of this
            // fragment share ONE window of 4 (R = 4, 8: both rows fall into the same phase half of a source pixel) or 5 (R = 2) image
            // rows x at most 37 columns x 3 channels: the wave loads it once, lane l taking column cbase + l of every row and channel
            // (12 or 15 two-byte loads per lane), passes it through LDS, and every lane picks its 16 taps per channel from there.
            // Before, every lane loaded its own 96 taps: the kernel was bound by the number of load INSTRUCTIONS (a 64-lane load of
            // any width occupies the CU's address unit for 16 cycles; 95 % of the image head's time).
            const int R = a.R;
            int row0[2];
            float cy[2][4];
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                const int Y = 2 * y + i2;
                const int ky = Y / R, phy = Y - ky * R;
                const float sy = (phy + 0.5f) / (float)R - 0.5f;
                const int fy = sy < 0.0f ? -1 : 0;
                cubic_coeffs(sy - (float)fy, cy[i2]);
                row0[i2] = ky + fy - 1;  // first (unclamped) row of the 4-tap window
            }
            const int rbase = __builtin_amdgcn_readfirstlane(row0[0] < row0[1] ? row0[0] : row0[1]);
            const int d0 = __builtin_amdgcn_readfirstlane(row0[0] - rbase), d1 = __builtin_amdgcn_readfirstlane(row0[1] - rbase);  // 0 or 1
            const bool five = (d0 | d1) != 0;
            const int cbase = __builtin_amdgcn_readfirstlane((2 * ex[mf]) / R) - 2;  // column of window slot 0 (fx - 1 >= -2)
            const int wcol = min(max(cbase + lane, 0), a.Wi - 1);                    // (slots past the window hold clamped repeats)
            uint32_t* const wl = (uint32_t*)win;
            uint32_t wv[15];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long long ip = ((long long)b * 3 + c) * plane_i;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i == 4 && !five) { wv[c * 5 + i] = 0u; break; }
                    wv[c * 5 + i] = ld_img_raw<TT, U8>(a.img, ip + (long long)min(max(rbase + i, 0), a.Hi - 1) * a.Wi + wcol);
                }
            }
#pragma unroll
            for (int j = 0; j < 15; ++j) wl[j * 64 + lane] = wv[j];
            __builtin_amdgcn_wave_barrier();
            if (x < a.W) {
                // horizontal taps of this output column
                const int kx = X / R, phx = X - kx * R;
                const float sx = (phx + 0.5f) / (float)R - 0.5f;
                const int fx = sx < 0.0f ? -1 : 0;
                float cx[4];
                cubic_coeffs(sx - (float)fx, cx);
                // window slot of the first tap: slot s holds column clamp(cbase + s), so slots o .. o + 3 are exactly the clamped taps
                // clamp(kx + fx - 1 + k) of the per-lane version; 0 <= o, o + 3 <= 36 (R = 2)
                const int o = kx + fx - 1 - cbase;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // the rows' horizontal sums once, shared by both output rows (same operations in the same order as a per-row loop)
                    float rowv[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        if (i == 4 && !five) { rowv[i] = 0.f; break; }
                        const uint32_t* t = wl + (c * 5 + i) * 64 + o;
                        rowv[i] = img_cvt<TT, U8>(t[0]) * cx[0] + img_cvt<TT, U8>(t[1]) * cx[1] + img_cvt<TT, U8>(t[2]) * cx[2] +
                                  img_cvt<TT, U8>(t[3]) * cx[3];
                    }
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2) {
                        const int d = i2 == 0 ? d0 : d1;
                        float sres = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sres += (d ? rowv[i + 1] : rowv[i]) * cy[i2][i];
                        float v = sres + zres[i2][c];
                        if (a.clamp) v = fminf(fmaxf(v, 0.0f), 1.0f);
                        const int Y = 2 * y + i2;
                        st_img<TT, U8>(a.out, (((long long)b * 3 + c) * plane_o) + (long long)Y * a.Wout + X, v);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Store epilogues (STORE, D2S, MIX), specialised at compile time; `conv_epilogue` below dispatches.
template <class TT, int NT, bool IS_CONV, int EPI, bool SILU>
__device__ __forceinline__ void store_epilogue(const ConvArgs& a, f32x16 (&acc)[2][NT], int lane, int nbase, int b,
                                               const int (&ey)[2], const int (&ex)[2], const long long (&em)[2]) {
    constexpr int SZ = TT::SZ;
    const int h = lane >> 5, r = lane & 31;
    // Direct 16-byte stores, no LDS.  An accumulator quad = 4 consecutive channels of the lane's pixel.  f32: that is one
    // 16-byte plane entry.  16-bit types: lanes (0, r) and (1, r) hold the two halves of an entry, so two quads are
    // exchanged with v_permlane32_swap: afterwards lane (0, r) owns all 8 channels of the even quad's plane and lane
    // (1, r) those of the odd quad's plane.  r walks 32 consecutive pixels: one store instruction writes two
    // 512-byte runs.
    constexpr int PPU = SZ == 2 ? 8 : 4;  // channels per plane (= per 16-byte unit)
    constexpr int UNITS = SZ == 2 ? 2 : 4;  // store units this lane produces per 32-channel accumulator tile
    const long long hwo = (long long)a.Ho * a.Wo;
    const long long M = (long long)a.B * hwo;
    constexpr bool d2s = IS_CONV && EPI == EPI_D2S;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int bimg = -1;       // image index, -1 = pixel outside the tensor
        long long pix = 0;   // y * Wo + x inside the image
        int py = 0, pxx = 0;
        if (IS_CONV) {
            py = ey[mf];
            pxx = ex[mf] + r;
            if (py < a.H && pxx < a.W) {
                bimg = b;
                pix = (long long)py * a.W + pxx;
            }
        } else {
            const long long m = em[mf] + r;
            if (m < M) {
                bimg = (int)(m / hwo);
                pix = m - (long long)bimg * hwo;
            }
        }
        const long long plane_o = d2s ? (long long)a.Hout * a.Wout * 16 : hwo * 16;
        char* const obase = (char*)a.out + (long long)(bimg < 0 ? 0 : bimg) * a.p_out * plane_o;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                float v[PPU];
                int cu;  // 16-byte unit index inside this workgroup's BN channels
                if constexpr (SZ == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ea = acc[mf][nt][8 * u + j], eb = acc[mf][nt][8 * u + 4 + j];
                        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, ea),
                                                                         __builtin_bit_cast(uint32_t, eb), false, false);
                        const uint32_t s0 = sw[0], s1 = sw[1];
                        v[j] = __builtin_bit_cast(float, s0);
                        v[4 + j] = __builtin_bit_cast(float, s1);
                    }
                    cu = 4 * nt + 2 * u + h;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[mf][nt][4 * u + j];
                    cu = 8 * nt + 2 * u + h;
                }
                const int n = nbase + cu * PPU;
                if constexpr (SILU) {
#pragma unroll
                    for (int j = 0; j < PPU; ++j) v[j] = v[j] * sigmoidf_(v[j]);
                }
                if (bimg < 0) continue;
                char* dst;
                if constexpr (d2s) {
                    if (n >= 4 * a.cp_out) continue;
                    const int ij = n / a.cp_out;
                    const int c = n - ij * a.cp_out;
                    const int Y = 2 * py + (ij >> 1), X = 2 * pxx + (ij & 1);
                    dst = obase + (c / PPU) * plane_o + ((long long)Y * a.Wout + X) * 16;
                } else {
                    if (n >= a.cp_out) continue;
                    const int plane = n / PPU;
                    if constexpr (!IS_CONV && EPI == EPI_MIX) {  // only the 1x1 kernel runs the mix
                        float xv[PPU], zv[PPU];
                        ld_unit<TT>((const char*)a.in0 + (((long long)bimg * a.p0 + plane) * hwo + pix) * 16, xv);
                        ld_unit<TT>((const char*)a.in1 + (((long long)bimg * a.p1 + plane) * hwo + pix) * 16, zv);
#pragma unroll
                        for (int j = 0; j < PPU; ++j) v[j] = blend_(xv[j], zv[j], v[j], a.inv_mix_scale);
                    }
                    dst = obase + plane * plane_o + pix * 16;
                }
                st_unit<TT>(dst, v);
            }
        }
    }
}

// ================================================================================================
// epilogue, shared by every convolution kernel.  Wave-local (no workgroup barrier); only FINAL touches LDS (the
// wave's own region `ep`).  Pixel geometry of the wave's two M fragments:
//   IS_CONV: fragment mf covers pixels (ey[mf], ex[mf] + r) of image b;   else: linear pixels em[mf] + r.
// ================================================================================================
template <class TT, int NT, bool IS_CONV>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const int epi, const int silu, f32x16 (&acc)[2][NT], char* ep,
                                              char* win, int lane, int nbase, int b, const int (&ey)[2], const int (&ex)[2],
                                              const long long (&em)[2]) {
    if (epi == EPI_FINAL) {
        if (IS_CONV) {
            if (a.io_u8) final_epilogue<TT, NT, true>(a, acc, ep, win, lane, b, ey, ex);
            else final_epilogue<TT, NT, false>(a, acc, ep, win, lane, b, ey, ex);
        }
        return;
    }
    if constexpr (IS_CONV) {
        if (epi == EPI_D2S) store_epilogue<TT, NT, true, EPI_D2S, false>(a, acc, lane, nbase, b, ey, ex, em);
        else if (silu) store_epilogue<TT, NT, true, EPI_STORE, true>(a, acc, lane, nbase, b, ey, ex, em);
        else store_epilogue<TT, NT, true, EPI_STORE, false>(a, acc, lane, nbase, b, ey, ex, em);
    } else {
        if (epi == EPI_MIX) store_epilogue<TT, NT, false, EPI_MIX, false>(a, acc, lane, nbase, b, ey, ex, em);
        else if (silu) store_epilogue<TT, NT, false, EPI_STORE, true>(a, acc, lane, nbase, b, ey, ex, em);
        else store_epilogue<TT, NT, false, EPI_STORE, false>(a, acc, lane, nbase, b, ey, ex, em);
    }
}

// ================================================================================================
// 3x3 convolution, wide tile: 512 output pixels x BN channels per workgroup, 8 compute waves + 1 loader wave.
//   - the weight stage (9 * NT KiB per K-chunk) is fetched ONCE for 512 pixels, by a dedicated wave;
//   - each compute wave issues only its 2-3 activation DMA instructions per stage;
//   - 3-slot LDS ring [A0 B0 | A1 B1 | A2 B2], prefetch distance 2 stages, counted s_waitcnt vmcnt(N): the
//     DMA of stage t+2 stays in flight across the single barrier of stage t.  The ring is rotated so that the
//     LAST stage sits in slot 0: slots 1-2 are then one contiguous free region during the last stage(s).
//   - FUSE: the AdaptiveResidualMix that follows conv2 of a block (reference model.py:507-511, 826-839) runs in
//     the epilogue.  With BN == all channels every wave owns all channels of its 64 pixels, so the gate
//     beta = Wx.x + Wz.z is wave-local: z goes from the accumulators straight into the MFMA B operand (the
//     accumulator rows are the K index; the gate weights are packed in that row order), x fragments come
//     from HBM as plain 16-byte loads (plane-major layout), and the gate weights are prefetched by the loader
//     wave into ring slots 1-2 while the last K-stage is being computed.
// ================================================================================================

// z accumulators -> MFMA B-operand fragments, and back to the (rounded) values for the blend
template <class TT> struct ZFrag;
template <> struct ZFrag<TF32> {
    static constexpr int ZG = 4;  // fragments per 32-row accumulator tile
    static __device__ __forceinline__ u32x4 make(const f32x16& t, int g) {
        // (copy each element to a scalar first: __builtin_bit_cast applied to an ext-vector element reads element 0)
        const float e0 = t[4 * g + 0], e1 = t[4 * g + 1], e2 = t[4 * g + 2], e3 = t[4 * g + 3];
        u32x4 f;
        f[0] = __builtin_bit_cast(uint32_t, e0); f[1] = __builtin_bit_cast(uint32_t, e1);
        f[2] = __builtin_bit_cast(uint32_t, e2); f[3] = __builtin_bit_cast(uint32_t, e3);
        return f;
    }
    static __device__ __forceinline__ void quad(const u32x4 (&f)[4], int q, float v[4]) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 t = __builtin_bit_cast(f32x4, f[q]);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
};
template <> struct ZFrag<TBF16> {
    static constexpr int ZG = 2;
    static __device__ __forceinline__ u32x4 make(const f32x16& t, int g) {
        u32x4 f;
        f[0] = pack_bf16(t[8 * g + 0], t[8 * g + 1]); f[1] = pack_bf16(t[8 * g + 2], t[8 * g + 3]);
        f[2] = pack_bf16(t[8 * g + 4], t[8 * g + 5]); f[3] = pack_bf16(t[8 * g + 6], t[8 * g + 7]);
        return f;
    }
    static __device__ __forceinline__ void quad(const u32x4 (&f)[2], int q, float v[4]) {
        const u32x4 t = f[q >> 1];
        const uint32_t lo = (q & 1) ? t[2] : t[0], hi = (q & 1) ? t[3] : t[1];
        v[0] = __builtin_bit_cast(float, lo << 16); v[1] = __builtin_bit_cast(float, lo & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, hi << 16); v[3] = __builtin_bit_cast(float, hi & 0xffff0000u);
    }
};
template <> struct ZFrag<TF16> {
    static constexpr int ZG = 2;
    static __device__ __forceinline__ u32x4 make(const f32x16& t, int g) {
        u32x4 f;
        f[0] = pack_f16(t[8 * g + 0], t[8 * g + 1]); f[1] = pack_f16(t[8 * g + 2], t[8 * g + 3]);
        f[2] = pack_f16(t[8 * g + 4], t[8 * g + 5]); f[3] = pack_f16(t[8 * g + 6], t[8 * g + 7]);
        return f;
    }
    static __device__ __forceinline__ void quad(const u32x4 (&f)[2], int q, float v[4]) {
        const u32x4 t = f[q >> 1];
        const uint32_t lo = (q & 1) ? t[2] : t[0], hi = (q & 1) ? t[3] : t[1];
        v[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(lo & 0xffff)); v[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(lo >> 16));
        v[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(hi & 0xffff)); v[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(hi >> 16));
    }
};

template <class TT, int NT, int MODE, bool FUSE>
__global__ __launch_bounds__(576, 3) void conv3w_kernel(const ConvArgs a) {
    using G = Geo<MODE>;
    constexpr int SZ = TT::SZ;
    constexpr int BN = 32 * NT;
    constexpr int A_SLOT = G::A_ENT * 16;
    constexpr int A_INSTR = G::A_ENT / 64;
    constexpr int B_PIECES = 9 * NT;
    constexpr int B_SLOT = B_PIECES * 1024;
    constexpr int SLOT = A_SLOT + B_SLOT;
    static_assert(B_PIECES < 60, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 compute, 8 = weight loader
    const int h = lane >> 5;
    const int r = lane & 31;

    int mtile, ntile;
    if (!map_tile(a, mtile, ntile)) return;  // padding id of a partial tile group (whole workgroup, uniform)
    const int nbase = ntile * BN;
    const int nstages = a.nchunks;
    const int last = nstages - 1;
    const int s0 = (3 - last % 3) % 3;  // slot(st) = (st + s0) % 3, so slot(last) == 0
    char* const mixw = smem + SLOT;     // FUSE: gate weights live in slots 1-2 once those are free

    if (w == 8) {
        // ------------------------- weight loader wave -------------------------
        const char* wsrc = (const char*)a.wpk + (size_t)ntile * a.nchunks * (B_PIECES * 1024) + lane * 16;
        auto loadB = [&](int st, int slot) {
            const char* src = wsrc + (size_t)st * (B_PIECES * 1024);
            char* dst = smem + slot * SLOT + A_SLOT;
#pragma unroll
            for (int j = 0; j < B_PIECES; ++j) glds16(src + j * 1024, dst + j * 1024);
        };
        int sl = s0;
        loadB(0, sl);
        sl = sl == 2 ? 0 : sl + 1;
        if (nstages > 1) loadB(1, sl);
        sl = sl == 2 ? 0 : sl + 1;  // slot of stage st + 2
        const int mix1 = FUSE ? (a.mix_pieces < SLOT / 1024 ? a.mix_pieces : SLOT / 1024) : 0;  // pieces that fit slot 1
        for (int st = 0; st < nstages; ++st) {
            if (st + 1 < nstages) wait_vmcnt<B_PIECES>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (st + 2 < nstages) loadB(st + 2, sl);
            if (FUSE) {
                const char* msrc = (const char*)a.wmix + lane * 16;
                if (st == (last > 0 ? last - 1 : 0))  // slot 1 was last read in stage last-2: free after this barrier
                    for (int j = 0; j < mix1; ++j) glds16(msrc + j * 1024, mixw + j * 1024);
                if (st == last)                       // slot 2 was last read in stage last-1
                    for (int j = mix1; j < a.mix_pieces; ++j) glds16(msrc + j * 1024, mixw + j * 1024);
            }
            sl = sl == 2 ? 0 : sl + 1;
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (FUSE) __builtin_amdgcn_s_barrier();  // the compute waves' barrier between the gate GEMM and the stores
        return;
    }

    // ------------------------- compute waves -------------------------
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = fdiv(mtile, tpi, a.inv_tpi);
    const int trem = mtile - b * tpi;
    const int tyi = fdiv(trem, a.tiles_x, a.inv_tiles_x);
    const int y0 = tyi * G::TH;
    const int x0 = (trem - tyi * a.tiles_x) * G::TW;

    // activation DMA: instruction j covers entries [64 j, 64 j + 64) of the halo image; wave w issues j = w, w+8, w+16
    const long long plane_in = (long long)a.H * a.W * 16;
    long long aoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = 64 * (w + 8 * i) + lane;
        const int plane = e >= G::PLANE_ENT ? 1 : 0;
        const int p = e - plane * G::PLANE_ENT;
        const int py = p / G::ROWW, px = p - py * G::ROWW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = (e < G::A_ENT) && (p < G::NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        aoff[i] = ok ? ((((long long)b * a.p0 + plane) * a.H + gy) * a.W + gx) * 16 : -1;
    }
    const int nA = (A_INSTR - w + 7) / 8;  // 2 or 3 instructions per stage for this wave
    auto loadA = [&](int st, int slot) {
        const long long kbyte = 2LL * st * plane_in;
        char* dst = smem + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (w + 8 * i >= A_INSTR) break;
            const char* src = aoff[i] >= 0 ? (const char*)a.in0 + aoff[i] + kbyte : (const char*)a.zero;
            glds16(src, dst + (w + 8 * i) * 1024);
        }
    };
    int slot = s0;
    int sl2 = s0;
    loadA(0, sl2);
    sl2 = sl2 == 2 ? 0 : sl2 + 1;
    if (nstages > 1) loadA(1, sl2);
    sl2 = sl2 == 2 ? 0 : sl2 + 1;  // slot of stage st + 2

    f32x16 acc[2][NT];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nt][i] = 0.0f;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds_base + h * G::PLANE + ((G::ROW_PER_WAVE * w) * G::ROWW + r) * 16;
    const uint32_t b_lane = lds_base + A_SLOT + lane * 16;

    // pixel geometry of this wave's two M fragments
    int ey[2], ex[2];
    if (G::ROW_PER_WAVE == 2) {
        ey[0] = y0 + 2 * w; ey[1] = y0 + 2 * w + 1;
        ex[0] = x0; ex[1] = x0;
    } else {
        ey[0] = y0 + w; ey[1] = y0 + w;
        ex[0] = x0; ex[1] = x0 + 32;
    }
    // FUSE: the block input x as MFMA B fragments, fetched by LDS-DMA into this wave's corner of ring slots 1-2
    const int ncx = FUSE ? a.p1 / 2 : 0;  // K-chunks of x (two planes per chunk)
    const long long plane_x = (long long)a.H * a.W * 16;
    char* const xr = mixw + (FUSE ? a.mix_pieces * 1024 + w * (ncx * 1024) : 0);
    auto pix_ok = [&](int mf) { return ey[mf] < a.H && ex[mf] + r < a.W; };
    auto x_base = [&](int mf) {
        return (const char*)a.in1 + ((((long long)b * a.p1) * a.H + ey[mf]) * a.W + ex[mf] + r) * 16;
    };
    auto x_dma = [&](int mf) {  // entry (chunk c, lane (h, r)) = plane 2c + h of pixel r
        const bool ok = pix_ok(mf);
        const char* xb = x_base(mf);
        for (int c = 0; c < ncx; ++c) glds16(ok ? xb + (2LL * c + h) * plane_x : (const char*)a.zero, xr + c * 1024);
    };

    for (int st = 0; st < nstages; ++st) {
        // my own activation DMA of stage st has landed once at most the newer stage's instructions are pending
        if (st + 1 < nstages) {
            if (nA == 3) wait_vmcnt<3>(); else wait_vmcnt<2>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();  // stage st is complete in LDS; everyone is done reading stage st-1
        if (st + 2 < nstages) loadA(st + 2, sl2);
        if (FUSE && st == last && a.x_via_lds) x_dma(0);  // slots 1-2 are free from here on; lands under this stage's MFMAs

        const uint32_t a_addr = a_lane + slot * SLOT;
        const uint32_t b_addr = b_lane + slot * SLOT;
        Frags<NT> fa, fb;
        issue_reads<NT, MODE, 0>(fa, a_addr, b_addr);
        wait_frags<NT>(fa);
        run_items<TT, NT, MODE, 0, 9>(acc, fa, fb, a_addr, b_addr);
        slot = slot == 2 ? 0 : slot + 1;
        sl2 = sl2 == 2 ? 0 : sl2 + 1;
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // all fragment reads are done (and, FUSE, the gate weights have landed)

    constexpr int EPW = 32 * (BN * SZ + 16) > 32 * 80 ? 32 * (BN * SZ + 16) : 32 * 80;
    const long long em[2] = {0, 0};

    if (FUSE) {
        // ---- AdaptiveResidualMix in registers: acc = z (conv2 output), x = a.in1 (the block input) ----
        using Z = ZFrag<TT>;
        constexpr int ZG = Z::ZG;
        constexpr int PPU = SZ == 2 ? 8 : 4;
        // Register diet (the 9-wave workgroup caps a wave at 168 VGPRs): z of BOTH fragments is packed to the storage
        // type first (the unfused path rounds z the same way when it stores it), the accumulators die, and each
        // fragment's blended result is packed again until the store phase.
        u32x4 zf[2][NT][ZG];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < ZG; ++g) {
                    zf[mf][nt][g] = Z::make(acc[mf][nt], g);
                    // opaque: stops hipcc from forwarding pack -> unpack and keeping 96 unpacked floats alive
                    asm volatile("" : "+v"(zf[mf][nt][g]));
                }
        u32x4 res[2][NT][ZG];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const bool inside = pix_ok(mf);
            const char* xbase = x_base(mf);
            f32x16 beta[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) beta[nt][i] = 0.0f;
            // gate, z half: accumulator rows are the K index (weights were packed in that row order)
            const char* wz = mixw + ncx * NT * 1024 + lane * 16;
#pragma unroll
            for (int ntz = 0; ntz < NT; ++ntz)
#pragma unroll
                for (int g = 0; g < ZG; ++g) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const u32x4 wv = *(const u32x4*)(wz + ((ntz * ZG + g) * NT + nt) * 1024);
                        mma<TT>(beta[nt], wv, zf[mf][ntz][g]);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep hipcc from hoisting every weight fragment read up front
                }
            // gate, x half
            if (a.x_via_lds) {
                if (mf == 1) wait_vmcnt<0>();  // mf 1's fragments were requested after mf 0's blend (below)
                for (int c = 0; c < ncx; ++c) {
                    const u32x4 xf = *(const u32x4*)(xr + c * 1024 + lane * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const u32x4 wv = *(const u32x4*)(mixw + (c * NT + nt) * 1024 + lane * 16);
                        mma<TT>(beta[nt], wv, xf);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int c0 = 0; c0 < ncx; c0 += 2) {  // straight from HBM, two K-chunks in flight (register budget)
                    u32x4 xf[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        xf[i] = u32x4{0u, 0u, 0u, 0u};
                        if (inside && c0 + i < ncx) xf[i] = *(const u32x4*)(xbase + (2LL * (c0 + i) + h) * plane_x);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (c0 + i < ncx) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const u32x4 wv = *(const u32x4*)(mixw + ((c0 + i) * NT + nt) * 1024 + lane * 16);
                                mma<TT>(beta[nt], wv, xf[i]);
                            }
                        }
                    }
                }
            }
            // blend: out = x + sigmoid(alpha) * sigmoid(beta) * (z - x), in place, one quad at a time
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float xv[4] = {0.f, 0.f, 0.f, 0.f}, zv[4];
                    Z::quad(zf[mf][nt], q, zv);
                    const int n = 32 * nt + 8 * q + 4 * h;
                    if (a.x_via_lds) {
                        // channels n..n+3 of pixel r sit in chunk n / CK, plane (n / PPU) & 1 of the fragment image
                        const int plane = n / PPU, inner = (n - plane * PPU) * SZ;
                        if (plane < a.p1) ld4<TT>(xr + (plane >> 1) * 1024 + ((plane & 1) * 32 + r) * 16 + inner, xv);
                    } else if (inside && n < a.cp_out) {
                        const int plane = n / PPU, inner = (n - plane * PPU) * SZ;
                        ld4<TT>(xbase + plane * plane_x + inner, xv);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        beta[nt][4 * q + j] = blend_(xv[j], zv[j], beta[nt][4 * q + j], a.inv_mix_scale);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < ZG; ++g) {
                    res[mf][nt][g] = Z::make(beta[nt], g);
                    asm volatile("" : "+v"(res[mf][nt][g]));
                }
            }
            if (a.x_via_lds && mf == 0) {
                __builtin_amdgcn_wave_barrier();
                x_dma(1);  // overlaps mf 1's z-half MFMAs
            }
        }
        // unpack the (already rounded) results back into the accumulator registers for the common store path
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
                    Z::quad(res[mf][nt], q, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mf][nt][4 * q + j] = v[j];
                }
        __builtin_amdgcn_s_barrier();  // every wave is done with the gate weights: the ring can take epilogue data
        conv_epilogue<TT, NT, true>(a, EPI_STORE, 0, acc, smem + w * EPW, smem + 8 * EPW + w * kFinalWinBytes, lane, nbase, b, ey, ex, em);
    } else {
        conv_epilogue<TT, NT, true>(a, a.epi, a.silu, acc, smem + w * EPW, smem + 8 * EPW + w * kFinalWinBytes, lane, nbase, b, ey, ex, em);
    }
}

// ================================================================================================
// 3x3 convolution, wide tile, PERSISTENT: one workgroup per CU walks its XCD's share of the tile list, and the
// LDS ring simply keeps turning across tile boundaries.  Two loader waves issue every LDS-DMA (wave 8 the halo
// images, wave 9 the weight stages), two K-stages ahead of the compute waves -- also across a tile boundary, so
// the first two stages of the next tile land while this tile's last stages and its epilogue run.  The compute
// waves issue no loads at all: they never wait on vmcnt, so the epilogue's stores drain under the next tile's
// MFMAs instead of at the end of a workgroup's life.  (Store epilogues only: STORE / D2S need no LDS.)
// ================================================================================================
template <class TT, int NT, int MODE>
__global__ __launch_bounds__(640) void conv3p_kernel(const ConvArgs a) {
    using G = Geo<MODE>;
    constexpr int A_SLOT = G::A_ENT * 16;
    constexpr int A_INSTR = G::A_ENT / 64;
    constexpr int B_PIECES = 9 * NT;
    constexpr int B_SLOT = B_PIECES * 1024;
    constexpr int SLOT = A_SLOT + B_SLOT;
    constexpr int BN = 32 * NT;
    static_assert(2 * B_PIECES < 64 && 2 * A_INSTR < 64, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 compute, 8 = halo loader, 9 = weight loader
    const int nstages = a.nchunks;

    // this workgroup's tile list: logical ids base + pos, base + pos + step, ... inside its XCD's contiguous range
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3, step = gridDim.x >> 3;
    const int q = a.grid >> 3, rem = a.grid & 7;
    const int cnt = q + (xcd < rem ? 1 : 0);
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    // first valid (non-padding) tile at or after list position i; cnt when the list is exhausted
    auto seek = [&](int i, int& mtile, int& ntile) __attribute__((always_inline)) {
        while (i < cnt && !tile_of(a, base + i, mtile, ntile)) i += step;
        return i;
    };
    int mtile = 0, ntile = 0;
    int cur = seek(pos, mtile, ntile);
    if (cur >= cnt) return;  // uniform over the workgroup

    const int tpi = a.tiles_x * a.tiles_y;
    auto tile_origin = [&](int mt, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        b = fdiv(mt, tpi, a.inv_tpi);
        const int trem = mt - b * tpi;
        const int tyi = fdiv(trem, a.tiles_x, a.inv_tiles_x);
        y0 = tyi * G::TH;
        x0 = (trem - tyi * a.tiles_x) * G::TW;
    };

    if (w >= 8) {
        // ------------------------- loader waves -------------------------
        // how many stages this workgroup will run in total (the compute waves meet us at one barrier per stage)
        int ntl = 0;
        {
            int mt_, nt_;
            for (int i = cur; i < cnt; i = seek(i + step, mt_, nt_)) ++ntl;
        }
        const int total = ntl * nstages;
        const long long plane_in = (long long)a.H * a.W * 16;
        int l_pos = cur, l_st = 0, l_slot = 0, pending = 0;
        bool l_ok = true;
        if (w == 9) {
            const char* wsrc = (const char*)a.wpk + (size_t)ntile * nstages * B_SLOT + lane * 16;
            auto issue = [&]() __attribute__((always_inline)) {
                if (!l_ok) return;
                const char* src = wsrc + (size_t)l_st * B_SLOT;
                char* dst = smem + l_slot * SLOT + A_SLOT;
#pragma unroll
                for (int j = 0; j < B_PIECES; ++j) glds16(src + j * 1024, dst + j * 1024);
                ++pending;
                l_slot = l_slot == 2 ? 0 : l_slot + 1;
                if (++l_st == nstages) {
                    l_st = 0;
                    int mt_, nt_ = 0;
                    l_pos = seek(l_pos + step, mt_, nt_);
                    l_ok = l_pos < cnt;
                    wsrc = (const char*)a.wpk + (size_t)nt_ * nstages * B_SLOT + lane * 16;
                }
            };
            issue();
            issue();
            for (int g = 0; g < total; ++g) {
                if (pending >= 2) wait_vmcnt<B_PIECES>(); else wait_vmcnt<0>();
                --pending;
                __builtin_amdgcn_s_barrier();
                issue();
            }
        } else {
            // halo image: instruction j covers entries [64 j, 64 j + 64); per-lane byte offsets inside image b
            uint32_t aoff[A_INSTR];
            const char* img = nullptr;
            auto set_tile = [&](int mt) __attribute__((always_inline)) {
                int b, y0, x0;
                tile_origin(mt, b, y0, x0);
                img = (const char*)a.in0 + (long long)b * a.p0 * plane_in;
#pragma unroll
                for (int j = 0; j < A_INSTR; ++j) {
                    const int e = 64 * j + lane;
                    const int plane = e >= G::PLANE_ENT ? 1 : 0;
                    const int p = e - plane * G::PLANE_ENT;
                    const int py = p / G::ROWW, px = p - py * G::ROWW;
                    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                    const bool ok = (p < G::NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    aoff[j] = ok ? (((uint32_t)plane * (uint32_t)a.H + (uint32_t)gy) * (uint32_t)a.W + (uint32_t)gx) * 16u : 0xffffffffu;  // host: planes * H * W * 16 < 2^32
                }
            };
            set_tile(mtile);
            auto issue = [&]() __attribute__((always_inline)) {
                if (!l_ok) return;
                const char* src = img + 2LL * l_st * plane_in;
                char* dst = smem + l_slot * SLOT;
#pragma unroll
                for (int j = 0; j < A_INSTR; ++j)
                    glds16(aoff[j] != 0xffffffffu ? src + aoff[j] : (const char*)a.zero, dst + j * 1024);
                ++pending;
                l_slot = l_slot == 2 ? 0 : l_slot + 1;
                if (++l_st == nstages) {
                    l_st = 0;
                    int mt_ = 0, nt_;
                    l_pos = seek(l_pos + step, mt_, nt_);
                    l_ok = l_pos < cnt;
                    if (l_ok) set_tile(mt_);
                }
            };
            issue();
            issue();
            for (int g = 0; g < total; ++g) {
                if (pending >= 2) wait_vmcnt<A_INSTR>(); else wait_vmcnt<0>();
                --pending;
                __builtin_amdgcn_s_barrier();
                issue();
            }
        }
        return;
    }

    // ------------------------- compute waves -------------------------
    const int h = lane >> 5, r = lane & 31;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds_base + h * G::PLANE + ((G::ROW_PER_WAVE * w) * G::ROWW + r) * 16;
    const uint32_t b_lane = lds_base + A_SLOT + lane * 16;
    const long long em[2] = {0, 0};
    int slot = 0;
    while (cur < cnt) {
        int b, y0, x0;
        tile_origin(mtile, b, y0, x0);
        const int nbase = ntile * BN;
        int ey[2], ex[2];
        if (G::ROW_PER_WAVE == 2) {
            ey[0] = y0 + 2 * w; ey[1] = y0 + 2 * w + 1;
            ex[0] = x0; ex[1] = x0;
        } else {
            ey[0] = y0 + w; ey[1] = y0 + w;
            ex[0] = x0; ex[1] = x0 + 32;
        }
        f32x16 acc[2][NT];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mf][nt][i] = 0.0f;

        for (int st = 0; st < nstages; ++st) {
            __builtin_amdgcn_s_barrier();  // stage landed (the loaders waited for it); everyone is done with the slot two back
            const uint32_t a_addr = a_lane + slot * SLOT;
            const uint32_t b_addr = b_lane + slot * SLOT;
            Frags<NT> fa, fb;
            issue_reads<NT, MODE, 0>(fa, a_addr, b_addr);
            wait_frags<NT>(fa);
            run_items<TT, NT, MODE, 0, 9>(acc, fa, fb, a_addr, b_addr);
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (a.epi == EPI_D2S) store_epilogue<TT, NT, true, EPI_D2S, false>(a, acc, lane, nbase, b, ey, ex, em);
        else if (a.silu) store_epilogue<TT, NT, true, EPI_STORE, true>(a, acc, lane, nbase, b, ey, ex, em);
        else store_epilogue<TT, NT, true, EPI_STORE, false>(a, acc, lane, nbase, b, ey, ex, em);
        cur = seek(cur + step, mtile, ntile);
    }
}

// ================================================================================================
// 3x3 convolution on v_mfma_f32_16x16x32_{bf16,f16}, persistent (16-bit types only).
// Why a second MFMA shape: the chip is power-limited in this loop and holds a visibly higher clock
// on the 16x16x32 shape than on 32x32x16 at identical FLOPs, LDS bytes and staging traffic.
//   * K-step of one MFMA = 32 channels = FOUR 16-byte planes: lane (g, c) = (lane >> 4, lane & 15) supplies plane g
//     of pixel c (B operand) / of output channel c (A operand); it receives channels 4g..4g+3 of pixel c.
//   * wave tile as before: 64 pixels x BN channels = 4 pixel fragments x 2*NT channel fragments (96 accumulator regs).
//   * a 32-channel K-stage with all 9 taps would need 94 KB per ring slot, so the two operands turn on separate
//     rings: the halo image (4 planes, 40 KB) is double-buffered per 32-channel chunk, the weights stream in
//     tap-ROW sub-stages (3 taps x 2*NT fragments = 18 KB) through 3 slots; one barrier per sub-stage.  With three
//     sub-stages per chunk the weight slot of a sub-stage is simply its tap row.
//   * loaders / persistence / tile walk exactly as conv3p_kernel.
//   * inside a sub-stage the fragments are software-pipelined per GROUP of 8 MFMAs (two channel fragments x four
//     pixel fragments): the next group's 2 weight fragments and a share of the next tap's 4 pixel fragments are
//     requested in the shadow of the group's first MFMAs.
// ================================================================================================
struct Frag16 {
    u32x4 x[2][4];  // [tap parity][pixel fragment]
    u32x4 w[3][2];  // [group % 3][channel fragment of the group]: requested TWO groups ahead
};
// LDS reads return in order, so lgkmcnt(N) = "everything but the N youngest reads has landed" (no scalar load is in
// flight inside the K loop: its straight-line code uses no kernel argument).  The registers named "+v" are the
// ones the following MFMAs may use; the N youngest stay untouched until a later wait.
template <int N> __device__ __forceinline__ void wait_w16(u32x4& w0, u32x4& w1) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w0), "+v"(w1) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_wx16(u32x4& w0, u32x4& w1, u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(w0), "+v"(w1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "n"(N) : "memory");
}
// byte offset of pixel fragment pf of tap (dy, dx) inside one plane of the halo image, relative to the wave's first row
template <int MODE, int TAP, int PF> constexpr int s16_a_off() {
    using G = Geo<MODE>;
    constexpr int DY = TAP / 3, DX = TAP % 3;
    return G::ROW_PER_WAVE == 2 ? ((DY + (PF >> 1)) * G::ROWW + DX + 16 * (PF & 1)) * 16 : (DY * G::ROWW + DX + 16 * PF) * 16;
}
// A 32-channel chunk = 9 * NT GROUPS; group G = (tap G / NT, channel-fragment pair G % NT) = 8 MFMAs, its two weight
// fragments are pieces 2G, 2G+1 of the chunk's packed weights.  The chunk's weights arrive in two halves (groups
// [0, G0) and [G0, NG), one LDS slot each), so there are two barriers per chunk -- the cadence of the 32x32x16
// kernel's two 16-channel stages.
// While group G runs it requests the weight pair of group G + 2 (inside the same half) and its share of the NEXT
// tap's pixel fragments (NT = 3: two each in the tap's groups 0 and 1; NT = 2: all four in group 0; NT = 1: all
// four, one group ahead).  Tap t uses pixel buffer t & 1; pixel fragments are prefetched across the mid-chunk barrier
// (the halo image does not change there), weight fragments are not (the second half has just landed).
template <int NT> struct S16Geo {
    static constexpr int NG = 9 * NT;
    static constexpr int G0 = (NG + 1) / 2;
    static constexpr int B_SLOT = 2 * G0 * 1024;
    // LDS byte offset of weight piece k of group G, relative to the weight area
    static constexpr int w_off(int G, int k) { return G < G0 ? (2 * G + k) * 1024 : B_SLOT + (2 * (G - G0) + k) * 1024; }
};
template <int NT, int G, int GE> struct S16Plan {  // group G of the segment ending at GE
    static constexpr int t = G / NT, n = G % NT;
    static constexpr bool w_issue = G + 2 < GE;
    // (requesting all four in the tap's first group keeps them live a group longer: 30 spilled VGPRs, +12 % time)
    static constexpr int x_count = t + 1 < 9 ? (NT == 3 ? (n < 2 ? 2 : 0) : (n == 0 ? 4 : 0)) : 0;
    static constexpr int x_first = NT == 3 ? 2 * n : 0;
    static constexpr int issued = (w_issue ? 2 : 0) + x_count;  // reads requested during this group
    // everything requested BEFORE this group has landed once at most `issued` reads are outstanding; NT = 1 needs the
    // pixel fragments it has just requested right away
    static constexpr int allow = NT == 1 ? (w_issue ? 2 : 0) : issued;
};
template <class TT, int NT, int MODE, int G, int GE, int M>
__device__ __forceinline__ void s16_mfmas(f32x4 (&acc)[4][2 * NT], Frag16& f, uint32_t a_addr, uint32_t b_addr) {
    if constexpr (M < 8) {
        using P = S16Plan<NT, G, GE>;
        constexpr int t = P::t, n = P::n, xp = t & 1, wp = G % 3;
        // pixel-fragment major, the channel pair in serpentine order, odd pairs of a tap walk the pixel fragments backwards: one operand
        // changes per MFMA (conv3r_kernel's order, DESIGN.md 5.2c: the same sums at a higher clock)
        constexpr int pf = (n & 1) ? 3 - (M >> 1) : (M >> 1), k = ((M >> 1) & 1) ? 1 - (M & 1) : (M & 1);
        mma16<TT>(acc[pf][2 * n + k], f.w[wp][k], f.x[xp][pf]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P::w_issue && M < 2) {
            f.w[(G + 2) % 3][M] = lds_read128<S16Geo<NT>::w_off(G + 2, M)>(b_addr);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (M >= 2 && M - 2 < P::x_count) {
            constexpr int pfn = P::x_first + M - 2;
            f.x[xp ^ 1][pfn] = lds_read128<s16_a_off<MODE, t + 1, pfn>()>(a_addr);
            __builtin_amdgcn_sched_barrier(0);
        }
        s16_mfmas<TT, NT, MODE, G, GE, M + 1>(acc, f, a_addr, b_addr);
    }
}
template <class TT, int NT, int MODE, int G, int GE>
__device__ __forceinline__ void s16_groups(f32x4 (&acc)[4][2 * NT], Frag16& f, uint32_t a_addr, uint32_t b_addr) {
    if constexpr (G < GE) {
        using P = S16Plan<NT, G, GE>;
        __builtin_amdgcn_sched_barrier(0);
        s16_mfmas<TT, NT, MODE, G, GE, 0>(acc, f, a_addr, b_addr);
        if constexpr (G + 1 < GE) {
            constexpr int wn = (G + 1) % 3, xn = ((G + 1) / NT) & 1;
            if constexpr ((G + 1) % NT == 0)  // the next group starts a new tap: its pixel fragments must be in
                wait_wx16<P::allow>(f.w[wn][0], f.w[wn][1], f.x[xn][0], f.x[xn][1], f.x[xn][2], f.x[xn][3]);
            else
                wait_w16<P::allow>(f.w[wn][0], f.w[wn][1]);
        }
        s16_groups<TT, NT, MODE, G + 1, GE>(acc, f, a_addr, b_addr);
    }
}
// first half of a chunk: a new halo image and the first weight half have just been published by the barrier
template <class TT, int NT, int MODE>
__device__ __forceinline__ void s16_front(f32x4 (&acc)[4][2 * NT], Frag16& f, uint32_t a_addr, uint32_t b_addr) {
    using S = S16Geo<NT>;
    f.x[0][0] = lds_read128<s16_a_off<MODE, 0, 0>()>(a_addr);
    f.x[0][1] = lds_read128<s16_a_off<MODE, 0, 1>()>(a_addr);
    f.x[0][2] = lds_read128<s16_a_off<MODE, 0, 2>()>(a_addr);
    f.x[0][3] = lds_read128<s16_a_off<MODE, 0, 3>()>(a_addr);
    f.w[0][0] = lds_read128<S::w_off(0, 0)>(b_addr);
    f.w[0][1] = lds_read128<S::w_off(0, 1)>(b_addr);
    f.w[1][0] = lds_read128<S::w_off(1, 0)>(b_addr);
    f.w[1][1] = lds_read128<S::w_off(1, 1)>(b_addr);
    wait_wx16<2>(f.w[0][0], f.w[0][1], f.x[0][0], f.x[0][1], f.x[0][2], f.x[0][3]);
    s16_groups<TT, NT, MODE, 0, S::G0>(acc, f, a_addr, b_addr);
}
// second half: only the weights are new; pixel fragments requested before the barrier are simply older in the queue
template <class TT, int NT, int MODE>
__device__ __forceinline__ void s16_back(f32x4 (&acc)[4][2 * NT], Frag16& f, uint32_t a_addr, uint32_t b_addr) {
    using S = S16Geo<NT>;
    constexpr int G0 = S::G0, w0 = G0 % 3, w1 = (G0 + 1) % 3, xn = (G0 / NT) & 1;
    f.w[w0][0] = lds_read128<S::w_off(G0, 0)>(b_addr);
    f.w[w0][1] = lds_read128<S::w_off(G0, 1)>(b_addr);
    f.w[w1][0] = lds_read128<S::w_off(G0 + 1, 0)>(b_addr);
    f.w[w1][1] = lds_read128<S::w_off(G0 + 1, 1)>(b_addr);
    wait_wx16<2>(f.w[w0][0], f.w[w0][1], f.x[xn][0], f.x[xn][1], f.x[xn][2], f.x[xn][3]);
    s16_groups<TT, NT, MODE, G0, S::NG>(acc, f, a_addr, b_addr);
}

// accumulators -> plane-major tensor.  Lane (g, c) holds channels 4g..4g+3 of pixel c of each 16-channel fragment;
// v_permlane16_swap between the two fragments of a group leaves lane g with one full 16-byte plane entry:
// fragment (g & 1) of the pair, plane (g >> 1) of that fragment.
// FILM (NO reference counterpart in the snapshot): a per-image, per-channel affine gamma * y + beta on the
// convolution result, ahead of the optional SiLU -- the shape of a FiLM / control-module modulation.
template <class TT, int NT, int MODE, int EPI, bool SILU, bool FILM = false>
__device__ __forceinline__ void store_frag16(const ConvArgs& a, f32x4 (&accpf)[2 * NT], const int pf, int lane, int w, int nbase,
                                             int b, int y0, int x0) {
    using G = Geo<MODE>;
    constexpr bool d2s = EPI == EPI_D2S;
    const int g = lane >> 4, c = lane & 15;
    const long long plane_o = d2s ? (long long)a.Hout * a.Wout * 16 : (long long)a.H * a.W * 16;
    char* const obase = (char*)a.out + (long long)b * a.p_out * plane_o;
    const int py = G::ROW_PER_WAVE == 2 ? y0 + 2 * w + (pf >> 1) : y0 + w;
    const int px = G::ROW_PER_WAVE == 2 ? x0 + 16 * (pf & 1) + c : x0 + 16 * pf + c;
    const bool inside = py < a.H && px < a.W;
    const int lane_cu = 2 * (g & 1) + (g >> 1);  // 16-byte unit of the lane inside a channel-fragment pair's 4 planes
    if constexpr (!d2s && !FILM) {
        // one 64-bit base per pixel fragment, then a uniform stride of four planes per pair (entry16(): mz_device.h)
        char* dst = obase + (long long)((nbase >> 3) + lane_cu) * plane_o + ((long long)py * a.W + px) * 16;
        const long long stride = 4 * plane_o;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const u32x4 o = entry16<TT, SILU>(accpf[2 * n], accpf[2 * n + 1]);
            const int nch = nbase + (4 * n + lane_cu) * 8;
            if (inside && nch < a.cp_out) *(u32x4*)dst = o;
            dst += stride;
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ea = accpf[2 * n][j], eb = accpf[2 * n + 1][j];
            const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ea), __builtin_bit_cast(uint32_t, eb),
                                                             false, false);
            const uint32_t s0 = sw[0], s1 = sw[1];
            v[j] = __builtin_bit_cast(float, s0);
            v[4 + j] = __builtin_bit_cast(float, s1);
        }
        const int nch = nbase + (4 * n + lane_cu) * 8;
        if constexpr (SILU && !FILM) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * sigmoidf_(v[j]);
        }
        if (!inside) continue;
        char* dst;
        if constexpr (d2s) {
            if (nch >= 4 * a.cp_out) continue;
            const int ij = nch / a.cp_out;
            const int ch = nch - ij * a.cp_out;
            const int Y = 2 * py + (ij >> 1), X = 2 * px + (ij & 1);
            dst = obase + (ch >> 3) * plane_o + ((long long)Y * a.Wout + X) * 16;
        } else {
            if (nch >= a.cp_out) continue;
            dst = obase + (nch >> 3) * plane_o + ((long long)py * a.W + px) * 16;
        }
        if constexpr (FILM) {  // gamma / beta: float [B][cp_out], pad channels zero (the host pads them)
            const float4* gp = (const float4*)(a.film_gamma + (long long)b * a.cp_out + nch);
            const float4* bp = (const float4*)(a.film_beta + (long long)b * a.cp_out + nch);
            const float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = gg[j] * v[j] + bb[j];
                if constexpr (SILU) v[j] = v[j] * sigmoidf_(v[j]);
            }
        }
        st_unit<TT>(dst, v);
    }
}
template <class TT, int NT, int MODE, int EPI, bool SILU, bool FILM = false>
__device__ __forceinline__ void store_epilogue16(const ConvArgs& a, f32x4 (&acc)[4][2 * NT], int lane, int w, int nbase, int b,
                                                 int y0, int x0) {
#pragma unroll
    for (int pf = 0; pf < 4; ++pf) store_frag16<TT, NT, MODE, EPI, SILU, FILM>(a, acc[pf], pf, lane, w, nbase, b, y0, x0);
}

// gate GEMM of the fused mix on the 16x16 layout: 2 NT K-steps of NF = 2 NT weight fragments each, walked in HALF
// steps of NT fragments: the next half step's fragments are requested before the current one's MFMAs are issued
// (LDS reads return in order: lgkmcnt(NT) = "everything but the NT reads just requested has landed").
template <int NT, int H, int I> __device__ __forceinline__ void gate_reads(u32x4 (&wv)[NT], uint32_t addr) {
    if constexpr (I < NT) {
        constexpr int ks = H >> 1, part = H & 1;
        wv[I] = lds_read128<(ks * 2 * NT + part * NT + I) * 1024>(addr);
        gate_reads<NT, H, I + 1>(wv, addr);
    }
}
template <int NT, int N> __device__ __forceinline__ void gate_wait(u32x4 (&wv)[NT]) {
    if constexpr (NT == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wv[0]) : "n"(N) : "memory");
    else if constexpr (NT == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wv[0]), "+v"(wv[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]) : "n"(N) : "memory");
}
template <class TT, int NT, int H>
__device__ __forceinline__ void gate_halves(f32x4 (&beta)[2 * NT], const u32x4 (&xf)[NT], const u32x4 (&zf)[NT], u32x4 (&wa)[NT],
                                            u32x4 (&wb)[NT], uint32_t addr) {
    if constexpr (H < 4 * NT) {
        constexpr int ks = H >> 1, part = H & 1;
        constexpr bool more = H + 1 < 4 * NT;
        if constexpr (more) gate_reads<NT, H + 1, 0>(wb, addr);  // wa = this half step's fragments, wb = the next one's
        gate_wait<NT, (more ? NT : 0)>(wa);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if constexpr (ks < NT) mma16<TT>(beta[part * NT + i], wa[i], xf[ks]);
            else mma16<TT>(beta[part * NT + i], wa[i], zf[ks - NT]);
        }
        __builtin_amdgcn_sched_barrier(0);
        gate_halves<TT, NT, H + 1>(beta, xf, zf, wb, wa, addr);
    }
}

// FUSE: conv2 + AdaptiveResidualMix (model.py:826-839) in one pass, as in conv3w_kernel<.., FUSE> but on the 16x16
// accumulator layout: after the K loop the wave packs z into MFMA B operands (two 16-channel accumulator fragments =
// one 32-wide K step; the gate weights were packed in that order, SRC_MIXF + frag16), two extra barriers let the
// weight loader drop the 4 NT^2 KB of gate weights into the second weight slot (+ the LDS behind it) once every wave
// has left the K loop, x arrives as plain 16-byte loads (the plane-major layout IS the B-operand layout), and the
// blend x + sigmoid(alpha) sigmoid(beta) (z - x) runs in the accumulator registers before the common store.
template <class TT> __device__ __forceinline__ void unpack2(uint32_t v, float& lo, float& hi) {
    if constexpr (TT::IS_BF16) {
        lo = __builtin_bit_cast(float, v << 16);
        hi = __builtin_bit_cast(float, v & 0xffff0000u);
    } else {
        lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(v & 0xffff));
        hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16));
    }
}
template <class TT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (TT::IS_BF16) return pack_bf16(lo, hi);
    else return pack_f16(lo, hi);
}

template <class TT, int NT, int MODE, bool FUSE>
__global__ __launch_bounds__(640) void conv3s_kernel(const ConvArgs a) {
    using G = Geo<MODE>;
    constexpr int NF = 2 * NT;
    constexpr int MIX_PIECES = 4 * NT * NT;  // FUSE: gate weights = 2 NT K-steps x NF fragments of 1 KiB
    constexpr int BN = 32 * NT;
    constexpr int A_PLANE = G::PLANE;
    constexpr int A_SLOT = 4 * A_PLANE;
    constexpr int A_INSTR = 4 * G::PLANE_ENT / 64;
    using SG = S16Geo<NT>;
    constexpr int P0 = 2 * SG::G0, P1 = 2 * (SG::NG - SG::G0);  // DMA pieces of the two weight halves of a chunk
    constexpr int B_SLOT = SG::B_SLOT;
    constexpr int B_BASE = 2 * A_SLOT;  // LDS: [halo 0][halo 1][weights: first half][weights: second half]
    static_assert(P0 < 64 && A_INSTR < 64, "vmcnt is a 6-bit counter");
    static_assert((4 * G::PLANE_ENT) % 64 == 0, "halo image = whole DMA instructions");
    static_assert(2 * B_SLOT < 65536, "ds offset is 16 bits");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 compute, 8 = halo loader, 9 = weight loader
    const int nchunks = a.nchunks16;                          // 32-channel chunks

    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3, step = gridDim.x >> 3;
    const int q = a.grid >> 3, rem = a.grid & 7;
    const int cnt = q + (xcd < rem ? 1 : 0);
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    auto seek = [&](int i, int& mtile, int& ntile) __attribute__((always_inline)) {
        while (i < cnt && !tile_of(a, base + i, mtile, ntile)) i += step;
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
        y0 = tyi * G::TH;
        x0 = txi * G::TW;
    };

    if (w >= 8) {
        int ntl = 0;
        {
            int mt_, nt_;
            for (int i = cur; i < cnt; i = seek(i + step, mt_, nt_)) ++ntl;
        }
        int l_pos = cur;
        bool l_ok = true;
        if (w == 9) {
            // ---- weight loader: the two halves of each chunk, one half ahead (two slots: half u + 1 goes where half
            //      u - 1 was, once everyone has passed barrier u) ----
            const size_t chunk_bytes = (size_t)(P0 + P1) * 1024;
            const char* wsrc = (const char*)a.wpk16 + (size_t)ntile * nchunks * chunk_bytes + lane * 16;
            int l_kc = 0, l_half = 0;
            auto issue = [&]() __attribute__((always_inline)) {
                if (!l_ok) return;
                const char* src = wsrc + (size_t)l_kc * chunk_bytes;
                if (l_half == 0) {
#pragma unroll
                    for (int j = 0; j < P0; ++j) glds16(src + j * 1024, smem + B_BASE + j * 1024);
                    l_half = 1;
                } else {
#pragma unroll
                    for (int j = 0; j < P1; ++j) glds16(src + (P0 + j) * 1024, smem + B_BASE + B_SLOT + j * 1024);
                    l_half = 0;
                    if (++l_kc == nchunks) {
                        l_kc = 0;
                        int mt_, nt_ = 0;
                        l_pos = seek(l_pos + step, mt_, nt_);
                        l_ok = l_pos < cnt;
                        wsrc = (const char*)a.wpk16 + (size_t)nt_ * nchunks * chunk_bytes + lane * 16;
                    }
                }
            };
            issue();
            int u = 0;
            for (int t = 0; t < ntl; ++t) {
                for (int hh = 0; hh < 2 * nchunks; ++hh, ++u) {
                    wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    issue();
                }
                if constexpr (FUSE) {
                    __builtin_amdgcn_s_barrier();  // E1: every wave has left the K loop: the second weight slot is free
                    const char* msrc = (const char*)a.wmix16 + lane * 16;
                    char* mdst = smem + B_BASE + B_SLOT;
#pragma unroll
                    for (int j = 0; j < MIX_PIECES; ++j) glds16(msrc + j * 1024, mdst + j * 1024);
                    wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();  // E2: the gate weights have landed
                }
            }
        } else {
            // ---- halo loader: one 4-plane image per 32-channel chunk, one chunk ahead.  Buffer-addressed LDS-DMA: the
            //      descriptor covers exactly the planes of this chunk that exist (2 or 4), so halo pixels outside the
            //      image (offset 0xffffffff) and the missing planes of a half chunk read as zeros by the hardware's
            //      range check -- no zero page, no per-lane 64-bit address arithmetic in the issue loop ----
            const long long plane_in = (long long)a.H * a.W * 16;
            uint32_t aoff[A_INSTR];
            const char* img = nullptr;
            auto set_tile = [&](int mt) __attribute__((always_inline)) {
                int b, y0, x0;
                tile_origin(mt, b, y0, x0);
                img = (const char*)a.in0 + (long long)b * a.p0 * plane_in;
#pragma unroll
                for (int j = 0; j < A_INSTR; ++j) {
                    const int e = 64 * j + lane;
                    const int plane = e / G::PLANE_ENT;
                    const int p = e - plane * G::PLANE_ENT;
                    const int py = p / G::ROWW, px = p - py * G::ROWW;
                    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                    const bool ok = (p < G::NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    aoff[j] = ok ? (((uint32_t)plane * (uint32_t)a.H + (uint32_t)gy) * (uint32_t)a.W + (uint32_t)gx) * 16u : 0xffffffffu;  // host: planes * H * W * 16 < 2^32
                }
            };
            set_tile(mtile);
            int l_kc = 0, l_slot = 0;
            auto issue = [&]() __attribute__((always_inline)) {
                if (!l_ok) return;
                const int planes = a.p0 - 4 * l_kc < 4 ? a.p0 - 4 * l_kc : 4;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(img + 4LL * l_kc * plane_in), 0, (int)(uint32_t)(planes * plane_in), 0x00020000);
                char* dst = smem + l_slot * A_SLOT;
#pragma unroll
                for (int j = 0; j < A_INSTR; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16,
                                                             (int)aoff[j], 0, 0, 0);
                l_slot ^= 1;
                if (++l_kc == nchunks) {
                    l_kc = 0;
                    int mt_ = 0, nt_;
                    l_pos = seek(l_pos + step, mt_, nt_);
                    l_ok = l_pos < cnt;
                    if (l_ok) set_tile(mt_);
                }
            };
            issue();
            int u = 0;
            for (int t = 0; t < ntl; ++t) {
                for (int hh = 0; hh < 2 * nchunks; ++hh, ++u) {
                    if ((hh & 1) == 0) wait_vmcnt<0>();  // a chunk's first barrier publishes its halo image
                    __builtin_amdgcn_s_barrier();
                    if ((hh & 1) == 0) issue();          // chunk c + 1 -> the slot chunk c - 1 was read from
                }
                if constexpr (FUSE) {
                    __builtin_amdgcn_s_barrier();  // E1
                    __builtin_amdgcn_s_barrier();  // E2
                }
            }
        }
        return;
    }

    // ------------------------- compute waves -------------------------
    const int g = lane >> 4, c = lane & 15;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_lane = lds_base + g * A_PLANE + ((G::ROW_PER_WAVE * w) * G::ROWW + c) * 16;
    const uint32_t b_lane = lds_base + B_BASE + lane * 16;
    uint32_t a_cur = a_lane, a_oth = a_lane + A_SLOT;  // this lane's address in the current / the other halo image
    Frag16 f;
    while (cur < cnt) {
        int b, y0, x0;
        tile_origin(mtile, b, y0, x0);
        f32x4 acc[4][NF];
#pragma unroll
        for (int pf = 0; pf < 4; ++pf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nchunks; ++kc) {
            __builtin_amdgcn_s_barrier();
            s16_front<TT, NT, MODE>(acc, f, a_cur, b_lane);
            __builtin_amdgcn_s_barrier();
            s16_back<TT, NT, MODE>(acc, f, a_cur, b_lane);
            const uint32_t tmp = a_cur; a_cur = a_oth; a_oth = tmp;
        }
        const int nbase = ntile * BN;
        if constexpr (FUSE) {
            // ---- AdaptiveResidualMix in registers: acc = z (conv2 output), x = a.in1 (the block input) ----
            u32x4 zb[4][NT];  // z as B operands: K step m = accumulator fragments 2m, 2m+1 (rounded to the storage type)
#pragma unroll
            for (int pf = 0; pf < 4; ++pf)
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    const f32x4 za = acc[pf][2 * m], zc = acc[pf][2 * m + 1];
                    u32x4 t;
                    t[0] = pack2<TT>(za[0], za[1]); t[1] = pack2<TT>(za[2], za[3]);
                    t[2] = pack2<TT>(zc[0], zc[1]); t[3] = pack2<TT>(zc[2], zc[3]);
                    asm volatile("" : "+v"(t));  // opaque: no pack -> unpack forwarding that would keep 96 floats alive
                    zb[pf][m] = t;
                }
            const long long hw = (long long)a.H * a.W;
            const char* const xim = (const char*)a.in1 + (long long)b * a.p1 * hw * 16;
            const uint32_t mix_lane = lds_base + B_BASE + B_SLOT + lane * 16;
            // x as B operands (plane 4 kc + g of the lane's pixel): requested one pixel fragment ahead of its use
            auto x_ptr = [&](int pf, bool& inside) __attribute__((always_inline)) {
                const int py = G::ROW_PER_WAVE == 2 ? y0 + 2 * w + (pf >> 1) : y0 + w;
                const int px = G::ROW_PER_WAVE == 2 ? x0 + 16 * (pf & 1) + c : x0 + 16 * pf + c;
                inside = py < a.H && px < a.W;
                return xim + ((long long)py * a.W + px) * 16;
            };
            auto load_xf = [&](int pf, u32x4 (&xf)[NT]) __attribute__((always_inline)) {
                bool inside;
                const char* const xp = x_ptr(pf, inside);
#pragma unroll
                for (int kc = 0; kc < NT; ++kc) {
                    xf[kc] = u32x4{0u, 0u, 0u, 0u};
                    if (inside && 4 * kc + g < a.p1) xf[kc] = *(const u32x4*)(xp + (long long)(4 * kc + g) * hw * 16);
                }
            };
            u32x4 xfa[NT], xfb[NT];
            load_xf(0, xfa);
            __builtin_amdgcn_s_barrier();  // E1
            __builtin_amdgcn_s_barrier();  // E2: gate weights are in LDS
#pragma unroll
            for (int pf = 0; pf < 4; ++pf) {
                __builtin_amdgcn_sched_barrier(0);
                u32x4 (&xf)[NT] = (pf & 1) ? xfb : xfa;
                if (pf + 1 < 4) load_xf(pf + 1, (pf & 1) ? xfa : xfb);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
                // gate: beta = Wx . x + Wz . z   (K steps 0..NT-1 = x, NT..2NT-1 = z)
                u32x4 wa[NT], wb[NT];
                gate_reads<NT, 0, 0>(wa, mix_lane);
                gate_halves<TT, NT, 0>(acc[pf], xf, zb[pf], wa, wb, mix_lane);
                // x again, in accumulator layout (channels 16 nf + 4 g .. + 3): the lines were just fetched above
                bool inside;
                const char* const xp = x_ptr(pf, inside);
                uint2 xq[NF];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    xq[nf] = make_uint2(0u, 0u);
                    if (inside && 2 * nf + (g >> 1) < a.p1) xq[nf] = *(const uint2*)(xp + (long long)(2 * nf + (g >> 1)) * hw * 16 + (g & 1) * 8);
                }
                // blend, in place: out = x + sigmoid(alpha) * sigmoid(beta) * (z - x)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    float zv[4], xv[4];
                    unpack2<TT>(zb[pf][nf >> 1][(nf & 1) * 2], zv[0], zv[1]);
                    unpack2<TT>(zb[pf][nf >> 1][(nf & 1) * 2 + 1], zv[2], zv[3]);
                    unpack2<TT>(xq[nf].x, xv[0], xv[1]);
                    unpack2<TT>(xq[nf].y, xv[2], xv[3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[pf][nf][j] = blend_(xv[j], zv[j], acc[pf][nf][j], a.inv_mix_scale);
                }
                store_frag16<TT, NT, MODE, EPI_STORE, false>(a, acc[pf], pf, lane, w, nbase, b, y0, x0);
            }
        } else
        if (a.epi == EPI_D2S) store_epilogue16<TT, NT, MODE, EPI_D2S, false>(a, acc, lane, w, nbase, b, y0, x0);
        else if (a.film_gamma) {
            if (a.silu) store_epilogue16<TT, NT, MODE, EPI_STORE, true, true>(a, acc, lane, w, nbase, b, y0, x0);
            else store_epilogue16<TT, NT, MODE, EPI_STORE, false, true>(a, acc, lane, w, nbase, b, y0, x0);
        }
        else if (a.silu) store_epilogue16<TT, NT, MODE, EPI_STORE, true>(a, acc, lane, w, nbase, b, y0, x0);
        else store_epilogue16<TT, NT, MODE, EPI_STORE, false>(a, acc, lane, w, nbase, b, y0, x0);
        cur = seek(cur + step, mtile, ntile);
    }
}

// ================================================================================================
// mix16_kernel: AdaptiveResidualMix (model.py:826-839) for C = k * 192 channels on the 16x16x32 MFMA (16-bit types).
//   out = x + sigmoid(alpha) * sigmoid(W [x ; z]) * (z - x),   W: [C, 2C]
// The 1x1 gate GEMM has no tap reuse, so it lives on activation traffic: the general 1x1 kernel stages x and z through
// LDS once per 96-channel N tile.  Here (a) an N tile is 192 channels (half the passes over x and z), and (b) x and z never
// touch LDS: in the plane-major layout a lane's 16 bytes of plane 4 ks + g of pixel c ARE its B-operand fragment of
// K step ks, so they are plain global loads, requested three K steps ahead.  Only the weights (12 KB per K step, shared
// by the 8 compute waves) go through LDS: a loader wave streams stages of 4 K steps into two slots.
// Workgroup = 256 pixels x 192 channels: wave w owns pixels 32 w .. 32 w + 31 (two 16-pixel fragments) x 12 channel
// fragments = 96 accumulator registers; weight pairs are read two 4-MFMA groups ahead with counted lgkmcnt.
// ================================================================================================
// blend_() of TWO values as one inline-asm block of two interleaved scalar-f32 chains (blend_()'s operations in blend_()'s order: identical
// bits).  Left to hipcc, the SLP vectoriser pairs the subtractions and fmas of neighbouring values into v_pk_add_f32 / v_pk_fma_f32, and
// packed-f32 instructions take ~40 cycles beside the MFMA stream of the SIMD's other wave instead of ~9 (tools/microbench/mb_coissue.hip,
// DESIGN.md 5.0) -- in mix16b_kernel a wave's blend runs beside its partner's K loop most of the time.
__device__ __forceinline__ void mix_blend_pair(float& o0, float& o1, const float b0, const float b1, const float x0, const float x1,
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

struct MixFrag {
    u32x4 w[3][2];
};
template <class TT, int G>  // group G of a stage: K step G / 6, channel-fragment pair G % 6
__device__ __forceinline__ void mix16_group(f32x4 (&acc)[2][12], MixFrag& f, const u32x4 (&xb)[2], uint32_t b_addr) {
    constexpr int n = G % 6, wp = G % 3;
    // request the pair of group G + 2 (same stage), then this group's four MFMAs
    if constexpr (G + 2 < 24) {
        f.w[(G + 2) % 3][0] = lds_read128<(((G + 2) / 6) * 12 + 2 * ((G + 2) % 6)) * 1024>(b_addr);
        f.w[(G + 2) % 3][1] = lds_read128<(((G + 2) / 6) * 12 + 2 * ((G + 2) % 6) + 1) * 1024>(b_addr);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int p0 = G & 1, p1 = p0 ^ 1;   // serpentine: one operand changes per MFMA; odd groups start on the other pixel fragment,
    mma16<TT>(acc[p0][2 * n], f.w[wp][0], xb[p0]);           // so the B operand also stays put across a group boundary
    mma16<TT>(acc[p0][2 * n + 1], f.w[wp][1], xb[p0]);
    mma16<TT>(acc[p1][2 * n + 1], f.w[wp][1], xb[p1]);
    mma16<TT>(acc[p1][2 * n], f.w[wp][0], xb[p1]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G + 1 < 24) wait_w16<(G + 2 < 24 ? 2 : 0)>(f.w[(G + 1) % 3][0], f.w[(G + 1) % 3][1]);
}
template <class TT, int KS> __device__ __forceinline__ void mix16_kstep(f32x4 (&acc)[2][12], MixFrag& f, const u32x4 (&xb)[2], uint32_t b_addr) {
    mix16_group<TT, KS * 6 + 0>(acc, f, xb, b_addr);
    mix16_group<TT, KS * 6 + 1>(acc, f, xb, b_addr);
    mix16_group<TT, KS * 6 + 2>(acc, f, xb, b_addr);
    mix16_group<TT, KS * 6 + 3>(acc, f, xb, b_addr);
    mix16_group<TT, KS * 6 + 4>(acc, f, xb, b_addr);
    mix16_group<TT, KS * 6 + 5>(acc, f, xb, b_addr);
}

template <class TT>
__global__ __launch_bounds__(576) void mix16_kernel(const ConvArgs a) {
    constexpr int STAGE = 4 * 12 * 1024;  // 4 K steps x 12 fragments
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7 compute, 8 = weight loader
    int mtile, ntile;
    if (!map_tile(a, mtile, ntile)) return;
    const int nsteps = a.nchunks16;  // K steps of 32 channels over [x ; z]; a multiple of 4
    const int nstages = nsteps >> 2;

    if (w == 8) {
        const char* src = (const char*)a.wpk16 + (size_t)ntile * nsteps * (12 * 1024) + lane * 16;
        auto issue = [&](int st) __attribute__((always_inline)) {
            char* dst = smem + (st & 1) * STAGE;
#pragma unroll
            for (int j = 0; j < 48; ++j) glds16(src + (size_t)st * STAGE + j * 1024, dst + j * 1024);
        };
        issue(0);
        for (int st = 0; st < nstages; ++st) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();            // stage st landed; everyone has finished stage st - 1
            if (st + 1 < nstages) issue(st + 1);
        }
        return;
    }

    const int g = lane >> 4, c = lane & 15;
    const long long hw = (long long)a.Ho * a.Wo;
    const long long M = (long long)a.B * hw;
    const int half_steps = nsteps >> 1;  // K steps of x (= of z)
    // Buffer-addressed loads: one descriptor per tensor (the host guarantees < 4 GiB), this lane's pixel as a 32-bit byte
    // offset of its plane g (0xffffffff = beyond the tensor: the range check returns zeros), the K step as a scalar offset.
    const uint32_t tensor_bytes = (uint32_t)((long long)a.B * a.p0 * hw * 16);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, 0, (int)tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)a.in1, 0, (int)tensor_bytes, 0x00020000);
    uint32_t voff[2];   // (image, plane g, pixel) -> bytes
    uint32_t vpix[2];   // (image, plane 0, pixel) -> bytes, for the epilogue
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) {
        const long long m = (long long)mtile * 256 + 32 * w + 16 * pf + c;
        const bool in = m < M;
        const long long mm = in ? m : 0;
        const int bimg = (int)(mm / hw);
        const long long pix = mm - (long long)bimg * hw;
        vpix[pf] = in ? (uint32_t)((((long long)bimg * a.p0) * hw + pix) * 16) : 0xffffffffu;
        voff[pf] = in ? (uint32_t)((((long long)bimg * a.p0 + g) * hw + pix) * 16) : 0xffffffffu;
    }
    const uint32_t step_bytes = (uint32_t)(4 * hw * 16);  // four planes per K step
    auto load_b = [&](int ks, u32x4 (&xb)[2]) __attribute__((always_inline)) {  // B operands of K step ks (zeros past the end)
        const bool isz = ks >= half_steps;
        const int kk = ks >= nsteps ? 0 : (isz ? ks - half_steps : ks);
        const int so = __builtin_amdgcn_readfirstlane((int)(kk * step_bytes));
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
            xb[pf] = isz ? __builtin_amdgcn_raw_buffer_load_b128(zr, (int)voff[pf], so, 0)
                         : __builtin_amdgcn_raw_buffer_load_b128(xr, (int)voff[pf], so, 0);
    };
    f32x4 acc[2][12];
#pragma unroll
    for (int pf = 0; pf < 2; ++pf)
#pragma unroll
        for (int nf = 0; nf < 12; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t b_lane = lds_base + lane * 16;
    // B operands two K steps ahead in three rotating buffers; the stage loop is unrolled three times so that the
    // rotation (12 K steps = 4 turns) is static -- the K-step count is a multiple of 12 whenever C is one of 192
    u32x4 xb0[2], xb1[2], xb2[2];
    load_b(0, xb0);
    load_b(1, xb1);
    MixFrag f;
    auto stage_head = [&](uint32_t b_addr) __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();
        f.w[0][0] = lds_read128<0>(b_addr);
        f.w[0][1] = lds_read128<1024>(b_addr);
        f.w[1][0] = lds_read128<2048>(b_addr);
        f.w[1][1] = lds_read128<3072>(b_addr);
        wait_w16<2>(f.w[0][0], f.w[0][1]);
    };
    for (int st = 0; st < nstages; st += 3) {
        const int ks = 4 * st;
        uint32_t b_addr = b_lane + (st & 1) * STAGE;
        stage_head(b_addr);
        load_b(ks + 2, xb2);  mix16_kstep<TT, 0>(acc, f, xb0, b_addr);
        load_b(ks + 3, xb0);  mix16_kstep<TT, 1>(acc, f, xb1, b_addr);
        load_b(ks + 4, xb1);  mix16_kstep<TT, 2>(acc, f, xb2, b_addr);
        load_b(ks + 5, xb2);  mix16_kstep<TT, 3>(acc, f, xb0, b_addr);
        b_addr = b_lane + ((st + 1) & 1) * STAGE;
        stage_head(b_addr);
        load_b(ks + 6, xb0);  mix16_kstep<TT, 0>(acc, f, xb1, b_addr);
        load_b(ks + 7, xb1);  mix16_kstep<TT, 1>(acc, f, xb2, b_addr);
        load_b(ks + 8, xb2);  mix16_kstep<TT, 2>(acc, f, xb0, b_addr);
        load_b(ks + 9, xb0);  mix16_kstep<TT, 3>(acc, f, xb1, b_addr);
        b_addr = b_lane + (st & 1) * STAGE;
        stage_head(b_addr);
        load_b(ks + 10, xb1); mix16_kstep<TT, 0>(acc, f, xb2, b_addr);
        load_b(ks + 11, xb2); mix16_kstep<TT, 1>(acc, f, xb0, b_addr);
        load_b(ks + 12, xb0); mix16_kstep<TT, 2>(acc, f, xb1, b_addr);
        load_b(ks + 13, xb1); mix16_kstep<TT, 3>(acc, f, xb2, b_addr);
    }
    // ---- blend and store, one pixel fragment and one channel-fragment pair at a time.  x and z come back in ACCUMULATOR layout
    //      (8 bytes per lane and fragment; L2 hits: the K loop has just read these lines); the loads of group i + 1 are requested
    //      before group i is blended -- left to itself hipcc requests them right before their use, twelve exposed L2 round trips
    //      per tile with nothing else in flight on the CU ----
    const int nbase = ntile * 192;
    const uint32_t plane_bytes = (uint32_t)(hw * 16);
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    struct XZ { u32x2_ x[2], z[2]; };
    auto request = [&](int i, XZ& q) __attribute__((always_inline)) {
        const int pf = i / 6, n = i - 6 * pf;
        const bool in = vpix[pf] != 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nf = 2 * n + k;
            const int plane = (nbase >> 3) + 2 * nf + (g >> 1);  // channels nbase + 16 nf + 4 g ..
            const uint32_t off = in ? vpix[pf] + (uint32_t)plane * plane_bytes + (g & 1) * 8 : 0xffffffffu;
            q.x[k] = __builtin_amdgcn_raw_buffer_load_b64(xr, (int)off, 0, 0);
            q.z[k] = __builtin_amdgcn_raw_buffer_load_b64(zr, (int)off, 0, 0);
        }
    };
    auto finish = [&](int i, const XZ& q) __attribute__((always_inline)) {
        const int pf = i / 6, n = i - 6 * pf;
        const bool in = vpix[pf] != 0xffffffffu;
        float v[8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nf = 2 * n + k;
            float xv[4], zv[4];
            unpack2<TT>(q.x[k][0], xv[0], xv[1]); unpack2<TT>(q.x[k][1], xv[2], xv[3]);
            unpack2<TT>(q.z[k][0], zv[0], zv[1]); unpack2<TT>(q.z[k][1], zv[2], zv[3]);
            float o0, o1, o2, o3;
            mix_blend_pair(o0, o1, acc[pf][nf][0], acc[pf][nf][1], xv[0], xv[1], zv[0], zv[1], a.inv_mix_scale);
            mix_blend_pair(o2, o3, acc[pf][nf][2], acc[pf][nf][3], xv[2], xv[3], zv[2], zv[3], a.inv_mix_scale);
            acc[pf][nf] = f32x4{o0, o1, o2, o3};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ea = acc[pf][2 * n][j], eb = acc[pf][2 * n + 1][j];
            const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ea), __builtin_bit_cast(uint32_t, eb), false, false);
            const uint32_t s0 = sw[0], s1 = sw[1];
            v[j] = __builtin_bit_cast(float, s0);
            v[4 + j] = __builtin_bit_cast(float, s1);
        }
        const int cu = 2 * (2 * n + (g & 1)) + (g >> 1);
        if (in) st_unit<TT>((char*)a.out + vpix[pf] + (long long)((nbase >> 3) + cu) * plane_bytes, v);
    };
    XZ qa, qb;
    request(0, qa);
#pragma unroll
    for (int i = 0; i < 12; i += 2) {
        request(i + 1, qb);
        __builtin_amdgcn_sched_barrier(0);
        finish(i, qa);
        if (i + 2 < 12) request(i + 2, qa);
        __builtin_amdgcn_sched_barrier(0);
        finish(i + 1, qb);
    }
}

hipError_t launch_mix16(int dtype, const ConvArgs& a, hipStream_t s) {
    if (a.mtiles <= 0 || a.ntiles <= 0 || a.gm <= 0 || a.gn <= 0 || a.grid <= 0 || a.grid >= (1 << 24)) return hipErrorInvalidValue;
    if (a.nchunks16 <= 0 || a.nchunks16 % 12) return hipErrorInvalidValue;  // three stages of four K steps per loop turn
    const size_t lds = 2 * 4 * 12 * 1024;
    switch (dtype) {
        case DT_BF16: hipLaunchKernelGGL(mix16_kernel<TBF16>, dim3(a.grid), dim3(576), lds, s, a); break;
        case DT_F16: hipLaunchKernelGGL(mix16_kernel<TF16>, dim3(a.grid), dim3(576), lds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ================================================================================================
// mix16b_kernel (round 3, second session): AdaptiveResidualMix for C = 192 without the second read of x and z, persistent.
// mix16_kernel blends in accumulator layout and therefore fetches x and z a second time (8-byte loads, L2 hit rate 0.38 at
// C = 192: counted traffic 2.79 GB against 1.79 GB algorithmic -- DESIGN 5.3), and its workgroup -- the only one its CU has room
// for -- alternates between a read-only K loop and a write-only epilogue.  Here
//   * the gate weights are packed (PackArgs::frag16 = 3) so that accumulator row 4 g + j of channel fragment 2 m + h is channel
//     32 m + 8 g + 4 h + j: lane (g, c) then owns, as accumulators, exactly the eight channels of pixel c that it loaded as the B
//     operand of K step m -- x, z and beta of one 16-byte plane entry sit in ONE lane: the 24 B operands of a unit (96 registers)
//     are kept until the blend, no second read, no v_permlane16_swap, one 16-byte store per entry;
//   * the whole gate matrix (144 KB) stays in LDS for the life of the workgroup (one per CU, eight waves of 256 registers, no loader
//     wave): after the first barrier there is no barrier and no LDS-DMA at all; every WAVE walks its own 32-pixel units;
//   * the loads of a wave's NEXT unit are issued between the stores of the current one, entry by entry into the registers the blend
//     has just released: reads and writes of a CU overlap, and the next K loop finds its first operands on the way.
// C = 192 only (one N tile whose twelve K steps are all its own): for C > 192 the other K steps have to stream through rotating
// buffers next to the 96 kept registers and hipcc spills in that loop, so C = 384 / 768 stay on mix16_kernel.
// ================================================================================================
template <class TT>
__global__ __launch_bounds__(512) void mix16b_kernel(const ConvArgs a) {
    constexpr int STAGE = 4 * 12 * 1024;  // 4 K steps x 12 fragments
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
    {   // the gate matrix: 144 pieces of 1 KB, 18 per wave
        const char* wsrc = (const char*)a.wpk16 + lane * 16 + w * (18 * 1024);
        char* dst = smem + w * (18 * 1024);
#pragma unroll
        for (int j = 0; j < 18; ++j) glds16(wsrc + j * 1024, dst + j * 1024);
    }
    const int g = lane >> 4, c = lane & 15;
    const long long hw = (long long)a.Ho * a.Wo;
    const long long M = (long long)a.B * hw;
    const uint32_t tensor_bytes = (uint32_t)((long long)a.B * a.p0 * hw * 16);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, 0, (int)tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)a.in1, 0, (int)tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)tensor_bytes, 0x00020000);
    const uint32_t step_bytes = (uint32_t)(4 * hw * 16);  // four planes per K step
    const float inv_hw = 1.0f / (float)hw;
    // (image, plane g, pixel) -> bytes for the lane's pixel of fragment pf of unit u; 0xffffffff beyond the tensor: the range check
    // returns zeros for such loads and drops such stores
    const int hwi = (int)hw, Mi = (int)M;   // the host guarantees B * p0 * hw * 16 < 2^32, so B * hw < 2^24
    auto offsets = [&](int u, uint32_t (&vo)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int m = u * 32 + 16 * pf + c;
            const bool in = m < Mi;
            const int mm = in ? m : 0;
            int bimg = (int)((float)mm * inv_hw);  // estimate (exact float of mm < 2^24), then corrected: no integer division per unit
            int pix = mm - bimg * hwi;
            if (pix < 0) { bimg -= 1; pix += hwi; }
            if (pix >= hwi) { bimg += 1; pix -= hwi; }
            vo[pf] = in ? ((uint32_t)(bimg * a.p0 + g) * (uint32_t)hwi + (uint32_t)pix) * 16u : 0xffffffffu;
        }
    };
    const int nunits = (Mi + 31) / 32;
    const int stride = (int)gridDim.x * 8;
    int u = (int)blockIdx.x * 8 + w;
    uint32_t voff[2];
    offsets(u, voff);
    u32x4 R[12][2];  // x K steps 0..5, z K steps 0..5 of the current unit
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int so = __builtin_amdgcn_readfirstlane((int)(i * step_bytes));
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) R[i][pf] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)voff[pf], so, 0);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int so = __builtin_amdgcn_readfirstlane((int)(i * step_bytes));
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) R[6 + i][pf] = __builtin_amdgcn_raw_buffer_load_b128(zr, (int)voff[pf], so, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<0>();                // this wave's 18 pieces of the gate matrix (once per workgroup: the first unit's loads may as well land)
    __builtin_amdgcn_s_barrier();   // the only barrier of the kernel: every wave reaches it, also one without a unit
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t b_lane = lds_base + lane * 16;
    MixFrag f;
    auto stage_head = [&](uint32_t b_addr) __attribute__((always_inline)) {
        f.w[0][0] = lds_read128<0>(b_addr);
        f.w[0][1] = lds_read128<1024>(b_addr);
        f.w[1][0] = lds_read128<2048>(b_addr);
        f.w[1][1] = lds_read128<3072>(b_addr);
        wait_w16<2>(f.w[0][0], f.w[0][1]);
    };
    while (u < nunits) {
        // the next unit's offsets first: here the accumulators are dead and their registers hold the temporaries
        const int un = u + stride;
        uint32_t vnext[2];
        offsets(un, vnext);   // beyond the last unit: 0xffffffff, the loads return zeros and nobody uses them
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][12];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int nf = 0; nf < 12; ++nf) acc[pf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        stage_head(b_lane);
        mix16_kstep<TT, 0>(acc, f, R[0], b_lane);
        mix16_kstep<TT, 1>(acc, f, R[1], b_lane);
        mix16_kstep<TT, 2>(acc, f, R[2], b_lane);
        mix16_kstep<TT, 3>(acc, f, R[3], b_lane);
        stage_head(b_lane + STAGE);
        mix16_kstep<TT, 0>(acc, f, R[4], b_lane + STAGE);
        mix16_kstep<TT, 1>(acc, f, R[5], b_lane + STAGE);
        mix16_kstep<TT, 2>(acc, f, R[6], b_lane + STAGE);
        mix16_kstep<TT, 3>(acc, f, R[7], b_lane + STAGE);
        stage_head(b_lane + 2 * STAGE);
        mix16_kstep<TT, 0>(acc, f, R[8], b_lane + 2 * STAGE);
        mix16_kstep<TT, 1>(acc, f, R[9], b_lane + 2 * STAGE);
        mix16_kstep<TT, 2>(acc, f, R[10], b_lane + 2 * STAGE);
        mix16_kstep<TT, 3>(acc, f, R[11], b_lane + 2 * STAGE);
        // ---- blend and store entry (K step m, pixel fragment pf) = plane 4 m + g of the lane's pixel; behind it, the same entry's
        //      x and z of the wave's next unit go into the registers just released ----
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int so = __builtin_amdgcn_readfirstlane((int)(m * step_bytes));
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float xl, xh, zl, zh;
                    unpack2<TT>(R[m][pf][q], xl, xh);
                    unpack2<TT>(R[6 + m][pf][q], zl, zh);
                    mix_blend_pair(v[2 * q], v[2 * q + 1], acc[pf][2 * m + (q >> 1)][2 * (q & 1)], acc[pf][2 * m + (q >> 1)][2 * (q & 1) + 1], xl, xh, zl, zh,
                                   a.inv_mix_scale);
                }
                u32x4 t;
                if constexpr (TT::IS_BF16) {
                    t[0] = pack_bf16(v[0], v[1]); t[1] = pack_bf16(v[2], v[3]); t[2] = pack_bf16(v[4], v[5]); t[3] = pack_bf16(v[6], v[7]);
                } else {
                    t[0] = pack_f16(v[0], v[1]); t[1] = pack_f16(v[2], v[3]); t[2] = pack_f16(v[4], v[5]); t[3] = pack_f16(v[6], v[7]);
                }
                // (an SGPR-offset store: hipcc puts no wait state behind it and had placed the next entry's first v_mul into v[data + 2]
                // right there -- garbage in a third of the runs; store16_soff() pins the wait states, mz_device.h)
                store16_soff(t, orr, (int)voff[pf], so);
            }
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                R[m][pf] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)vnext[pf], so, 0);
                R[6 + m][pf] = __builtin_amdgcn_raw_buffer_load_b128(zr, (int)vnext[pf], so, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        u = un;
        voff[0] = vnext[0];
        voff[1] = vnext[1];
    }
}

hipError_t launch_mix16b(int dtype, const ConvArgs& a, hipStream_t s, int workgroups) {
    if (a.mtiles <= 0 || a.ntiles != 1 || workgroups <= 0) return hipErrorInvalidValue;
    if (a.nchunks16 != 12) return hipErrorInvalidValue;  // C = 192: twelve K steps over [x ; z], all of them the N tile's own
    const size_t lds = 3 * 4 * 12 * 1024;  // the whole gate matrix
    const int grid = a.mtiles < workgroups ? a.mtiles : workgroups;   // a.mtiles = 256-pixel tiles = 8 units each
    switch (dtype) {
        case DT_BF16: hipLaunchKernelGGL(mix16b_kernel<TBF16>, dim3(grid), dim3(512), lds, s, a); break;
        case DT_F16: hipLaunchKernelGGL(mix16b_kernel<TF16>, dim3(grid), dim3(512), lds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <class TT, int NT, int MODE>
__global__ __launch_bounds__(256, 2) void conv_kernel(const ConvArgs a) {
    using G = Geo<MODE>;
    constexpr int SZ = TT::SZ;
    constexpr int TAPS = G::TAPS;
    constexpr int S = G::S;
    constexpr int BN = 32 * NT;
    constexpr int A_BYTES = G::A_ENT * 16;
    constexpr int B_PIECES = TAPS * S * NT;
    constexpr int STAGE = A_BYTES + B_PIECES * 1024;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int r = lane & 31;

    // ---- workgroup -> (pixel tile, N tile); consecutive logical ids share an XCD (and its L2) ----
    int mtile, ntile;
    if (!map_tile(a, mtile, ntile)) return;  // padding id of a partial tile group (whole workgroup, uniform)
    const int nbase = ntile * BN;
    const char* wtile = (const char*)a.wpk + (size_t)ntile * a.nchunks * (TAPS * NT * 1024);

    // ---- tile geometry ----
    int b = 0, y0 = 0, x0 = 0;   // CONV3
    long long m0 = 0;            // GEMM1
    const long long M = (long long)a.B * a.Ho * a.Wo;
    if (MODE == MODE_CONV3) {
        const int tpi = a.tiles_x * a.tiles_y;
        b = fdiv(mtile, tpi, a.inv_tpi);
        const int rem = mtile - b * tpi;
        const int ty = fdiv(rem, a.tiles_x, a.inv_tiles_x);
        y0 = ty * 8;
        x0 = (rem - ty * a.tiles_x) * 32;
    } else {
        m0 = (long long)mtile * 256;
    }

    // ---- per-thread staging sources (fixed for the whole K loop) ----
    // CONV3: entries e = tid + 256*i of the halo image; GEMM1: pixel m0 + tid of both sources.
    long long aoff[3];
    if (MODE == MODE_CONV3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + 256 * i;
            const int plane = e >= 352 ? 1 : 0;
            const int p = e - plane * 352;
            const int py = p / 34, px = p - py * 34;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool ok = (e < 704) && (p < 340) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            aoff[i] = ok ? ((((long long)b * a.p0 + plane) * a.H + gy) * a.W + gx) * 16 : -1;
        }
    } else {
        const long long m = m0 + tid;
        aoff[0] = aoff[1] = aoff[2] = -1;
        if (m < M) {
            const int hwo = a.Ho * a.Wo;
            const int bb = (int)(m / hwo);
            const int pix = (int)(m - (long long)bb * hwo);
            if (a.src == SRC_CRUSH) {
                const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
                aoff[0] = ((long long)bb * a.p0 * a.H * a.W + (long long)(2 * oy) * a.W + 2 * ox) * 16;
            } else {
                aoff[0] = ((long long)bb * a.p0 * hwo + pix) * 16;
                aoff[1] = ((long long)bb * a.p1 * hwo + pix) * 16;
            }
        }
    }
    const long long plane_in = (long long)a.H * a.W * 16;  // bytes between two planes of an input tensor

    auto stage_load = [&](int st, int buf) {
        char* Abuf = smem + buf * STAGE;
        char* Bbuf = Abuf + A_BYTES;
        // ---- weights: contiguous run of pieces, one KiB per wave-instruction ----
        const int kc0 = st * S;
        const int npieces = B_PIECES;  // GEMM1: nchunks is padded to a multiple of S with zero weights
        const char* wsrc = wtile + (size_t)kc0 * (TAPS * NT * 1024);
        for (int j = w; j < npieces; j += 4) {
            if (kGlds) {
                glds16(wsrc + j * 1024 + lane * 16, Bbuf + j * 1024);
            } else {
                *(uint4*)(Bbuf + j * 1024 + lane * 16) = *(const uint4*)(wsrc + j * 1024 + lane * 16);
            }
        }
        // ---- activations ----
        if (MODE == MODE_CONV3) {
            const long long kbyte = 2LL * kc0 * plane_in;  // a K-chunk = two planes
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i == 2 && w == 3) break;  // entries 704.. do not exist
                const char* src = aoff[i] >= 0 ? (const char*)a.in0 + aoff[i] + kbyte : (const char*)a.zero;
                if (kGlds) {
                    glds16(src, Abuf + (64 * w + 256 * i) * 16);
                } else {
                    *(uint4*)(Abuf + (tid + 256 * i) * 16) = *(const uint4*)src;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int kc = kc0 + s;
                const char* base;
                if (kc >= a.nchunks_real) {
                    base = nullptr;  // K padding: zero activations against zero weights
                } else if (a.src == SRC_CRUSH) {
                    const int tap = kc / a.nchunks0;
                    const int cc = kc - tap * a.nchunks0;
                    const long long toff = ((long long)(tap >> 1) * a.W + (tap & 1)) * 16 + 2LL * cc * plane_in;
                    base = aoff[0] >= 0 ? (const char*)a.in0 + aoff[0] + toff : nullptr;
                } else if (kc < a.nchunks0) {
                    base = aoff[0] >= 0 ? (const char*)a.in0 + aoff[0] + 2LL * kc * plane_in : nullptr;
                } else {
                    base = aoff[1] >= 0 ? (const char*)a.in1 + aoff[1] + 2LL * (kc - a.nchunks0) * plane_in : nullptr;
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const char* src = base ? base + hh * plane_in : (const char*)a.zero;
                    char* dstw = Abuf + s * 8192 + hh * 4096 + (64 * w) * 16;
                    if (kGlds) {
                        glds16(src, dstw);
                    } else {
                        *(uint4*)(dstw + lane * 16) = *(const uint4*)src;
                    }
                }
            }
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mf][nt][i] = 0.0f;

    const int a_lane = (MODE == MODE_CONV3) ? h * G::PLANE + ((2 * w) * 34 + r) * 16
                                            : h * G::PLANE + (64 * w + r) * 16;

    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int nstages = (a.nchunks + S - 1) / S;
    stage_load(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int st = 0; st < nstages; ++st) {
        const int cur = st & 1;
        if (st + 1 < nstages) stage_load(st + 1, cur ^ 1);

        const uint32_t a_addr = lds_base + cur * STAGE + a_lane;
        const uint32_t b_addr = lds_base + cur * STAGE + A_BYTES + lane * 16;
        Frags<NT> fa, fb;
        issue_reads<NT, MODE, 0>(fa, a_addr, b_addr);
        wait_frags<NT>(fa);
        run_items<TT, NT, MODE, 0, TAPS * S>(acc, fa, fb, a_addr, b_addr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ============================== epilogue ==============================
    // The staging buffers are free now (every wave is past the last barrier, no DMA in flight).
    {
        constexpr int EPW = 32 * (BN * SZ + 16) > 32 * 80 ? 32 * (BN * SZ + 16) : 32 * 80;
        const int ey[2] = {y0 + 2 * w, y0 + 2 * w + 1};
        const int ex[2] = {x0, x0};
        const long long em[2] = {m0 + 64 * w, m0 + 64 * w + 32};
        conv_epilogue<TT, NT, MODE == MODE_CONV3>(a, a.epi, a.silu, acc, smem + w * EPW, smem + 4 * EPW + w * kFinalWinBytes, lane, nbase, b, ey, ex, em);
    }
}

// the 16x16x32 kernel: two 4-plane halo images + the two halves of a chunk's weights
size_t conv16_lds_bytes(int mode, int nt, bool fuse) {
    const size_t a_slot = 4 * (size_t)(mode == MODE_C3W16 ? 640 : 672) * 16;
    const size_t b_slot = 2 * (size_t)((9 * nt + 1) / 2) * 1024;
    const size_t gate = fuse ? (size_t)4 * nt * nt * 1024 : 0;  // lives in the second weight slot and the LDS behind it
    return 2 * a_slot + b_slot + (gate > b_slot ? gate : b_slot);
}

size_t conv_lds_bytes(int mode, int nt) {
    if (mode == MODE_C3W16 || mode == MODE_C3W8) {
        const int a_slot = (mode == MODE_C3W16 ? 2 * 640 : 2 * 672) * 16;
        const size_t ring = 3 * (size_t)(a_slot + 9 * nt * 1024);
        // epilogue scratch of the 8 compute waves + (EPI_FINAL) their bicubic windows behind it: smem + 8 * EPW + w * kFinalWinBytes
        const size_t epi = 8 * 32 * (size_t)(32 * nt * 4 + 16) + 8 * (size_t)kFinalWinBytes;
        return ring > epi ? ring : epi;
    }
    const int taps = mode == MODE_CONV3 ? 9 : 1;
    const int S = mode == MODE_CONV3 ? 1 : MZ_GEMM1_S;
    const int a_bytes = (mode == MODE_CONV3 ? 704 : MZ_GEMM1_S * 512) * 16;
    const size_t staging = 2 * (size_t)(a_bytes + taps * S * nt * 1024);
    const size_t epi = 4 * 32 * (size_t)(32 * nt * 4 + 16) + 4 * (size_t)kFinalWinBytes;  // (+ the EPI_FINAL windows, as above)
    return staging > epi ? staging : epi;
}

int gemm1_chunks_per_stage() { return MZ_GEMM1_S; }

int choose_nt(int n_padded) {
    // smallest padded N wins; ties prefer 3, 2, 4, 1 (4 needs 94 KiB of LDS: one workgroup per CU)
    const int order[4] = {3, 2, 4, 1};
    int best = 1, best_n = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        const int bn = 32 * order[i];
        const int padded = (n_padded + bn - 1) / bn * bn;
        if (padded < best_n) {
            best_n = padded;
            best = order[i];
        }
    }
    return best;
}

template <class TT, int NT, int MODE> static hipError_t launch_one(const ConvArgs& a, hipStream_t s) {
    const size_t lds = conv_lds_bytes(MODE, NT);
    if constexpr (MODE == MODE_C3W16 || MODE == MODE_C3W8) {
        if constexpr (NT <= 3) {
            if (a.epi == EPI_FUSEDMIX && a.persist > 0 && a.s16) {
                if constexpr (TT::SZ == 2)
                    hipLaunchKernelGGL((conv3s_kernel<TT, NT, MODE, true>), dim3(a.persist), dim3(640), conv16_lds_bytes(MODE, NT, true), s, a);
                else
                    return hipErrorInvalidValue;
            } else if (a.epi == EPI_FUSEDMIX)
                hipLaunchKernelGGL((conv3w_kernel<TT, NT, MODE, true>), dim3(a.grid), dim3(576), lds, s, a);
            else if (a.persist > 0 && a.s16 && (a.epi == EPI_STORE || a.epi == EPI_D2S)) {
                if constexpr (TT::SZ == 2)
                    hipLaunchKernelGGL((conv3s_kernel<TT, NT, MODE, false>), dim3(a.persist), dim3(640), conv16_lds_bytes(MODE, NT, false), s, a);
                else
                    return hipErrorInvalidValue;
            } else if (a.persist > 0 && (a.epi == EPI_STORE || a.epi == EPI_D2S))
                hipLaunchKernelGGL((conv3p_kernel<TT, NT, MODE>), dim3(a.persist), dim3(640), lds, s, a);
            else
                hipLaunchKernelGGL((conv3w_kernel<TT, NT, MODE, false>), dim3(a.grid), dim3(576), lds, s, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        hipLaunchKernelGGL((conv_kernel<TT, NT, MODE>), dim3(a.grid), dim3(256), lds, s, a);
    }
    return hipGetLastError();
}
template <class TT, int MODE> static hipError_t launch_nt(int nt, const ConvArgs& a, hipStream_t s) {
    switch (nt) {
        case 1: return launch_one<TT, 1, MODE>(a, s);
        case 2: return launch_one<TT, 2, MODE>(a, s);
        case 3: return launch_one<TT, 3, MODE>(a, s);
        case 4: return launch_one<TT, 4, MODE>(a, s);
    }
    return hipErrorInvalidValue;
}
template <class TT> static hipError_t launch_mode(int mode, int nt, const ConvArgs& a, hipStream_t s) {
    switch (mode) {
        case MODE_CONV3: return launch_nt<TT, MODE_CONV3>(nt, a, s);
        case MODE_GEMM1: return launch_nt<TT, MODE_GEMM1>(nt, a, s);
        case MODE_C3W16: return launch_nt<TT, MODE_C3W16>(nt, a, s);
        case MODE_C3W8: return launch_nt<TT, MODE_C3W8>(nt, a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_conv(int dtype, int mode, int nt, const ConvArgs& a, hipStream_t s) {
    if (a.mtiles <= 0 || a.ntiles <= 0 || a.gm <= 0 || a.gn <= 0 || a.grid <= 0) return hipErrorInvalidValue;
    if (a.grid >= (1 << 24)) return hipErrorInvalidValue;  // fdiv() needs dividends below 2^24
    switch (dtype) {
        case DT_F32: return launch_mode<TF32>(mode, nt, a, s);
        case DT_BF16: return launch_mode<TBF16>(mode, nt, a, s);
        case DT_F16: return launch_mode<TF16>(mode, nt, a, s);
    }
    return hipErrorInvalidValue;
}

template <class TT, int NT, int MODE> static hipError_t set_lds_one() {
    const int bytes = (int)conv_lds_bytes(MODE, NT);
    if constexpr (MODE == MODE_C3W16 || MODE == MODE_C3W8) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3w_kernel<TT, NT, MODE, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv3p_kernel<TT, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        if constexpr (TT::SZ == 2) {
            e = hipFuncSetAttribute((const void*)conv3s_kernel<TT, NT, MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)conv16_lds_bytes(MODE, NT, false));
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)conv3s_kernel<TT, NT, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)conv16_lds_bytes(MODE, NT, true));
            if (e != hipSuccess) return e;
        }
        return hipFuncSetAttribute((const void*)conv3w_kernel<TT, NT, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   bytes);
    } else {
        return hipFuncSetAttribute((const void*)conv_kernel<TT, NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
}
template <class TT> static hipError_t set_lds_all() {
    hipError_t e;
#define MZ_SET(NT, MODE) \
    if ((e = set_lds_one<TT, NT, MODE>()) != hipSuccess) return e;
    MZ_SET(1, MODE_CONV3) MZ_SET(2, MODE_CONV3) MZ_SET(3, MODE_CONV3) MZ_SET(4, MODE_CONV3)
    MZ_SET(1, MODE_GEMM1) MZ_SET(2, MODE_GEMM1) MZ_SET(3, MODE_GEMM1) MZ_SET(4, MODE_GEMM1)
    MZ_SET(1, MODE_C3W16) MZ_SET(2, MODE_C3W16) MZ_SET(3, MODE_C3W16)
    MZ_SET(1, MODE_C3W8) MZ_SET(2, MODE_C3W8) MZ_SET(3, MODE_C3W8)
#undef MZ_SET
    return hipSuccess;
}
hipError_t init_kernels() {
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)mix16_kernel<TBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 12 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)mix16_kernel<TF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 12 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)mix16b_kernel<TBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * 12 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)mix16b_kernel<TF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * 12 * 1024)) != hipSuccess) return e;
    if ((e = set_lds_all<TF32>()) != hipSuccess) return e;
    if ((e = set_lds_all<TBF16>()) != hipSuccess) return e;
    if ((e = set_lds_all<TF16>()) != hipSuccess) return e;
    return hipSuccess;
}

// ================================================================================================
// weight packing: OIHW float32 -> [ntile][kchunk][tap][nt][lane][16 bytes] in the compute dtype
// ================================================================================================
template <class TT> __global__ void pack_kernel(const PackArgs a, long long total) {
    constexpr int SZ = TT::SZ;
    constexpr int CK = TT::CK;
    constexpr int EPL = 16 / SZ;  // elements per lane
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    long long t = idx;
    const int e = (int)(t % EPL); t /= EPL;
    const int lane = (int)(t % 64); t /= 64;
    const int nfr = a.frag16 ? (a.nfr ? a.nfr : 2 * a.nt) : a.nt;  // fragments per tap: 16-channel (16x16x32 MFMA) or 32-channel ones
    const int nt = (int)(t % nfr); t /= nfr;
    const int tap = (int)(t % a.taps); t /= a.taps;
    const int kc = (int)(t % a.nchunks); t /= a.nchunks;
    const int nb = (int)t;
    int n = a.frag16 ? (nb * nfr + nt) * 16 + (lane & 15) : (nb * a.nt + nt) * 32 + (lane & 31);
    if (a.frag16 == 3) {
        // mix16b_kernel: accumulator rows in B-OPERAND order.  Row r = 4 g + j of fragment 2 m + h stands for channel
        // 32 m + 8 g + 4 h + j of the N tile, so that lane (g, c) of the accumulators owns exactly the eight channels whose x and z it
        // loaded as the B operand of K step m: the blend needs no second read of x and z, and its result is a whole 16-byte entry
        const int r = lane & 15;
        n = nb * (nfr * 16) + 32 * (nt >> 1) + 8 * (r >> 2) + 4 * (nt & 1) + (r & 3);
    }
    const int hh = lane >> 5;
    const int kin = a.frag16 ? (lane >> 4) * 8 + e : hh * (CK / 2) + e;  // channel within the chunk
    const int ckk = a.frag16 ? 32 : CK;                                  // channels per chunk

    // output channel
    int o = -1;
    if (a.out_map == OUT_PLAIN) {
        o = n < a.cout ? n : -1;
    } else if (a.out_map == OUT_D2S) {
        const int ij = n / a.cq_p, c = n - ij * a.cq_p;
        o = (ij < 4 && c < a.cq) ? c * 4 + ij : -1;  // PixelShuffle(2): in-channel = c*4 + 2i + j
    } else {
        const int ij = n >> 2, c = n & 3;
        o = (n < 16 && c < 3) ? c * 4 + ij : -1;
    }
    // input channel and filter tap
    int ci = -1, ty = 0, tx = 0;
    if (a.in_map == SRC_PLAIN) {
        const int k = kc * ckk + kin;
        ci = k < a.c0 ? k : -1;
        ty = tap / a.kw;
        tx = tap - ty * a.kw;
    } else if (a.in_map == SRC_CONCAT) {
        int ks = kc;
        if (a.frag16 == 3) {
            // ... and the K steps of an N tile start with its OWN x and z channels (six steps each, kept in registers for the blend);
            // the rest follows in natural order (mix16b_step() in the kernel is the same map)
            const int hs = a.nchunks >> 1, t6 = 6 * nb;
            if (kc < 6) ks = t6 + kc;
            else if (kc < 12) ks = hs + t6 + (kc - 6);
            else {
                ks = kc - 12;
                if (ks >= t6) ks += 6;
                if (ks >= hs + t6) ks += 6;
            }
        }
        const int k = ks * ckk + kin;
        if (k < a.cp0) ci = k < a.c0 ? k : -1;
        else ci = (k - a.cp0) < a.c1 ? a.c0 + (k - a.cp0) : -1;
    } else if (a.in_map == SRC_MIXF && a.frag16 == 4) {
        // conv3t_kernel's gate (C <= 48: three 16-channel fragments of x, three of z): K step kc = fragments 2 kc and 2 kc + 1 of
        // [x0 x1 x2 z0 z1 z2], the K elements of lane group g in accumulator-row order: e < 4 -> the first fragment's channels 4 g + e,
        // e >= 4 -> the second's
        const int g = lane >> 4;
        const int fr = 2 * kc + (e < 4 ? 0 : 1);
        const int ch = 16 * (fr % 3) + 4 * g + (e & 3);
        if (fr < 3) ci = ch < a.c0 ? ch : -1;
        else ci = ch < a.c1 ? a.c0 + ch : -1;
    } else if (a.in_map == SRC_MIXF && a.frag16) {
        // fused gate for the 16x16x32 kernel: K-steps [0, ncx) = x channels in natural order (32 per step); then one
        // K-step per PAIR of 16-channel accumulator fragments of z, K elements in the order the accumulator quads of
        // lane group g = lane >> 4 supply them: e < 4 -> fragment 2m, channel 4g + e; e >= 4 -> fragment 2m + 1
        const int ncx = (a.cp0 + 31) / 32;
        if (kc < ncx && a.frag16 == 2) {
            // conv3r_kernel's fused variant: the x half in accumulator-row order too (x is fetched in accumulator layout, so
            // that a pair of its channel fragments is a B operand as it stands)
            const int g = lane >> 4;
            const int xch = 32 * kc + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
            ci = xch < a.c0 ? xch : -1;
        } else if (kc < ncx) {
            const int k = kc * 32 + kin;
            ci = k < a.c0 ? k : -1;
        } else {
            const int m = kc - ncx, g = lane >> 4;
            const int zch = 32 * m + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
            ci = zch < a.c1 ? a.c0 + zch : -1;
        }
    } else if (a.in_map == SRC_MIXF) {
        // fused AdaptiveResidualMix gate: chunks [0, ncx) = x channels in natural order; then one chunk per
        // (32-row accumulator tile, fragment g) of z, K-elements in ACCUMULATOR ROW order (ZFrag<TT>::make)
        const int ncx = a.cp0 / CK;
        if (kc < ncx) {
            const int k = kc * CK + kin;
            ci = k < a.c0 ? k : -1;
        } else {
            constexpr int ZG = SZ == 2 ? 2 : 4;
            const int gz = kc - ncx, ntz = gz / ZG, g = gz - ntz * ZG;
            const int zrow = 32 * ntz + (SZ == 2 ? 16 * g + 8 * (e >> 2) + 4 * hh + (e & 3) : 8 * g + 4 * hh + e);
            ci = zrow < a.c1 ? a.c0 + zrow : -1;
        }
    } else {  // CRUSH: K axis = [tap][padded channel]
        const int cpt = a.cp0 / CK;  // chunks per tap
        const int st = kc / cpt;
        const int k = (kc - st * cpt) * CK + kin;
        ci = (st < 4 && k < a.c0) ? k : -1;  // st >= 4: K padding
        ty = st >> 1;
        tx = st & 1;
    }
    float v = 0.0f;
    if (o >= 0 && ci >= 0) v = a.w[(((long long)o * a.cin + ci) * a.kh + ty) * a.kw + tx];
    st1<TT>((char*)a.dst + idx * SZ, v);
}

size_t packed_bytes(int taps, int nt, int ntiles, int nchunks) {
    return (size_t)ntiles * nchunks * taps * nt * 1024;
}

hipError_t launch_pack(const PackArgs& a, hipStream_t s) {
    const int sz = dtype_size(a.dtype);
    const long long total = (long long)packed_bytes(a.taps, a.frag16 ? (a.nfr ? a.nfr : 2 * a.nt) : a.nt, a.ntiles, a.nchunks) / sz;
    const int blocks = (int)((total + 255) / 256);
    switch (a.dtype) {
        case DT_F32: hipLaunchKernelGGL(pack_kernel<TF32>, dim3(blocks), dim3(256), 0, s, a, total); break;
        case DT_BF16: hipLaunchKernelGGL(pack_kernel<TBF16>, dim3(blocks), dim3(256), 0, s, a, total); break;
        case DT_F16: hipLaunchKernelGGL(pack_kernel<TF16>, dim3(blocks), dim3(256), 0, s, a, total); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ================================================================================================
// small kernels
// ================================================================================================
__global__ void pack_stem_kernel(const float* w, const float* b, float* dst, int c, int cp) {
    // dst: float4 per channel {w0, w1, w2, bias}; zero-initialised by the caller; either of w / b may be null
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cp) return;
    float4 v = ((float4*)dst)[i];
    if (i < c) {
        if (w) { v.x = w[i * 3 + 0]; v.y = w[i * 3 + 1]; v.z = w[i * 3 + 2]; }
        if (b) v.w = b[i];
    } else {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ((float4*)dst)[i] = v;
}
hipError_t launch_pack_stem(const float* w, const float* b, float* dst, int c, int cp, hipStream_t s) {
    hipLaunchKernelGGL(pack_stem_kernel, dim3((cp + 63) / 64), dim3(64), 0, s, w, b, dst, c, cp);
    return hipGetLastError();
}

// FanOutProjection (reference model.py:239-242): per-pixel 3 -> C affine, NCHW image -> plane-major features.
template <class TT, bool U8> __global__ void stem_kernel(const void* x, const float4* w4, void* out, long long total,
                                                          long long HW, int groups) {
    constexpr int SZ = TT::SZ;
    constexpr int NPL = SZ / 2;  // planes per group of 8 channels
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long p = idx % HW;  // pixel fastest: loads and stores are contiguous along p
    const long long t = idx / HW;
    const int g = (int)(t % groups);
    const long long b = t / groups;
    const long long xi = b * 3 * HW + p;
    const float r0 = ld_img<TT, U8>(x, xi), r1 = ld_img<TT, U8>(x, xi + HW), r2 = ld_img<TT, U8>(x, xi + 2 * HW);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 wv = w4[g * 8 + j];
        v[j] = wv.w + wv.x * r0 + wv.y * r1 + wv.z * r2;
    }
    char* op = (char*)out + ((b * groups * NPL + (long long)g * NPL) * HW + p) * 16;
    if (SZ == 2) {
        st4<TT>(op, v);
        st4<TT>(op + 4 * SZ, v + 4);
    } else {
        st4<TT>(op, v);
        st4<TT>(op + HW * 16, v + 4);
    }
}
hipError_t launch_stem(int dtype, const void* x, const float* w4, void* out, int B, int H, int W, int cp, hipStream_t s,
                       int u8) {
    const long long HW = (long long)H * W;
    const int groups = cp / 8;
    const long long total = HW * B * groups;
    const int blocks = (int)((total + 255) / 256);
    switch (dtype) {
        case DT_F32: if (u8) hipLaunchKernelGGL((stem_kernel<TF32, true>), dim3(blocks), dim3(256), 0, s, x, (const float4*)w4, out, total, HW, groups); else hipLaunchKernelGGL((stem_kernel<TF32, false>), dim3(blocks), dim3(256), 0, s, x, (const float4*)w4, out, total, HW, groups); break;
        case DT_BF16: if (u8) hipLaunchKernelGGL((stem_kernel<TBF16, true>), dim3(blocks), dim3(256), 0, s, x, (const float4*)w4, out, total, HW, groups); else hipLaunchKernelGGL((stem_kernel<TBF16, false>), dim3(blocks), dim3(256), 0, s, x, (const float4*)w4, out, total, HW, groups); break;
        case DT_F16: if (u8) hipLaunchKernelGGL((stem_kernel<TF16, true>), dim3(blocks), dim3(256), 0, s, x, (const float4*)w4, out, total, HW, groups); else hipLaunchKernelGGL((stem_kernel<TF16, false>), dim3(blocks), dim3(256), 0, s, x, (const float4*)w4, out, total, HW, groups); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Decoder.crop_feature_maps zero padding (reference model.py:667-673, 681-687): bottom rows / right columns.
__global__ void zero_border_kernel(char* t, int B, int Hout, int Wout, int planes, int Hv, int Wv) {
    // one thread per border pixel of one plane; border = rows >= Hv (all columns) and columns >= Wv (rows < Hv)
    const long long nb_rows = (long long)(Hout - Hv) * Wout;
    const long long nb_cols = (long long)Hv * (Wout - Wv);
    const long long per_plane = nb_rows + nb_cols;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = per_plane * B * planes;
    if (idx >= total) return;
    long long k = idx % per_plane;
    const long long bp = idx / per_plane;  // b * planes + plane
    int y, x;
    if (k < nb_rows) {
        y = Hv + (int)(k / Wout);
        x = (int)(k % Wout);
    } else {
        k -= nb_rows;
        const int wc = Wout - Wv;
        y = (int)(k / wc);
        x = Wv + (int)(k % wc);
    }
    *(uint4*)(t + ((bp * Hout + y) * Wout + x) * 16) = make_uint4(0, 0, 0, 0);
}
hipError_t launch_zero_border(int dtype, void* t, int B, int Hout, int Wout, int cp, int Hv, int Wv, hipStream_t s) {
    if (Hv >= Hout && Wv >= Wout) return hipSuccess;
    const int planes = cp * dtype_size(dtype) / 16;
    const long long per_plane = (long long)(Hout - Hv) * Wout + (long long)Hv * (Wout - Wv);
    const long long total = per_plane * B * planes;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(zero_border_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (char*)t, B, Hout, Wout,
                       planes, Hv, Wv);
    return hipGetLastError();
}

// QualityAssessor pooling (reference model.py:1028-1030): spatial mean of the conv output, plus the conv bias.
template <class TT> __global__ void qa_reduce_kernel(const void* feat, const float* bias, float* qa, int P, int cp, int F) {
    constexpr int SZ = TT::SZ;
    __shared__ float red[256];
    constexpr int PPU = 16 / SZ;  // channels per plane
    const int b = blockIdx.x, f = blockIdx.y;
    const int planes = cp / PPU;
    const char* base = (const char*)feat + (((long long)b * planes + f / PPU) * P) * 16 + (f % PPU) * SZ;
    float sum = 0.0f;
    for (int p = threadIdx.x; p < P; p += 256) sum += ld1<TT>(base + (long long)p * 16);
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) qa[b * F + f] = red[0] / (float)P + bias[f];
}
hipError_t launch_qa_reduce(int dtype, const void* feat, const float* bias, float* qa, int B, int P, int cp, int F,
                            hipStream_t s) {
    switch (dtype) {
        case DT_F32: hipLaunchKernelGGL(qa_reduce_kernel<TF32>, dim3(B, F), dim3(256), 0, s, feat, bias, qa, P, cp, F); break;
        case DT_BF16: hipLaunchKernelGGL(qa_reduce_kernel<TBF16>, dim3(B, F), dim3(256), 0, s, feat, bias, qa, P, cp, F); break;
        case DT_F16: hipLaunchKernelGGL(qa_reduce_kernel<TF16>, dim3(B, F), dim3(256), 0, s, feat, bias, qa, P, cp, F); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); }

}  // namespace mz
