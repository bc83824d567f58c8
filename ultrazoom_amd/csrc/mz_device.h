// Device-side building blocks shared by the kernel translation units (mz_kernels.hip, mz_conv3r.hip, mz_conv3t.hip): element types,
// conversions, MFMA wrappers, LDS-DMA / LDS fragment-read primitives, the XCD-aware tile walk.
#pragma once
#include "mz_kernels.h"

namespace mz {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct TF32 {
    static constexpr int SZ = 4;
    static constexpr int CK = 8;
    static constexpr bool IS_BF16 = false;
};
struct TBF16 {
    static constexpr int SZ = 2;
    static constexpr int CK = 16;
    static constexpr bool IS_BF16 = true;
};
struct TF16 {
    static constexpr int SZ = 2;
    static constexpr int CK = 16;
    static constexpr bool IS_BF16 = false;
};

// ---- scalar conversions -----------------------------------------------------------------------
template <class TT> __device__ __forceinline__ float ld1(const void* p);
template <> __device__ __forceinline__ float ld1<TF32>(const void* p) { return *(const float*)p; }
template <> __device__ __forceinline__ float ld1<TBF16>(const void* p) {
    return __builtin_bit_cast(float, (uint32_t)(*(const uint16_t*)p) << 16);
}
template <> __device__ __forceinline__ float ld1<TF16>(const void* p) { return (float)(*(const _Float16*)p); }

template <class TT> __device__ __forceinline__ void st1(void* p, float v);
template <> __device__ __forceinline__ void st1<TF32>(void* p, float v) { *(float*)p = v; }
template <> __device__ __forceinline__ void st1<TBF16>(void* p, float v) { *(__bf16*)p = (__bf16)v; }
template <> __device__ __forceinline__ void st1<TF16>(void* p, float v) { *(_Float16*)p = (_Float16)v; }

// image pixels at the two ends of the path: the module dtype, or uint8 with the scaling every caller of the reference
// applies around upscale() (torchvision ToDtype(scale=True) before, save_image's mul(255).add(0.5).clamp().byte() after;
// reference README.md:72-83, test_compare.py:53-57,89)
template <class TT, bool U8> __device__ __forceinline__ float ld_img(const void* base, long long idx) {
    if constexpr (U8) return (float)((const uint8_t*)base)[idx] / 255.0f;  // a true division, as ToDtype(scale=True) does
    else return ld1<TT>((const char*)base + idx * TT::SZ);
}
// raw element load / conversion split, so that a batch of loads can be issued before the first conversion
template <class TT, bool U8> __device__ __forceinline__ uint32_t ld_img_raw(const void* base, long long idx) {
    if constexpr (U8) return ((const uint8_t*)base)[idx];
    else if constexpr (TT::SZ == 4) return ((const uint32_t*)base)[idx];
    else return ((const uint16_t*)base)[idx];
}
template <class TT, bool U8> __device__ __forceinline__ float img_cvt(uint32_t raw) {
    if constexpr (U8) return (float)raw / 255.0f;
    else if constexpr (TT::SZ == 4) return __builtin_bit_cast(float, raw);
    else if constexpr (TT::IS_BF16) return __builtin_bit_cast(float, raw << 16);
    else return (float)__builtin_bit_cast(_Float16, (uint16_t)raw);
}
template <class TT, bool U8> __device__ __forceinline__ void st_img(void* base, long long idx, float v) {
    if constexpr (U8) ((uint8_t*)base)[idx] = (uint8_t)fminf(fmaxf(v * 255.0f + 0.5f, 0.0f), 255.0f);
    else st1<TT>((char*)base + idx * TT::SZ, v);
}

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    uint16_t lo = __builtin_bit_cast(uint16_t, (__bf16)a);
    uint16_t hi = __builtin_bit_cast(uint16_t, (__bf16)b);
    return (uint32_t)lo | ((uint32_t)hi << 16);
}
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    // Opaque: the value is rounded to f32 first in EVERY kernel.  Left alone, hipcc folds a preceding multiply into a
    // v_fma_mix* form (one rounding, straight to f16) in some instantiations and not in others, and two kernels
    // that must agree bit for bit (per-tile vs persistent, any batch size) then differ in the last f16 bit.
    asm("" : "+v"(a), "+v"(b));
    // gfx950: ONE v_cvt_pk_f16_f32 (round to nearest even, as the two scalar conversions + a pack it replaces)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{a, b}, f16x2_));
}

// 4 consecutive channels <-> memory
template <class TT> __device__ __forceinline__ void st4(void* p, const float v[4]);
template <> __device__ __forceinline__ void st4<TF32>(void* p, const float v[4]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<TBF16>(void* p, const float v[4]) {
    *(uint2*)p = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
}
template <> __device__ __forceinline__ void st4<TF16>(void* p, const float v[4]) {
    *(uint2*)p = make_uint2(pack_f16(v[0], v[1]), pack_f16(v[2], v[3]));
}
template <class TT> __device__ __forceinline__ void ld4(const void* p, float v[4]);
template <> __device__ __forceinline__ void ld4<TF32>(const void* p, float v[4]) {
    float4 t = *(const float4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<TBF16>(const void* p, float v[4]) {
    uint2 t = *(const uint2*)p;
    v[0] = __builtin_bit_cast(float, t.x << 16);
    v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, t.y << 16);
    v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld4<TF16>(const void* p, float v[4]) {
    uint2 t = *(const uint2*)p;
    v[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(t.x & 0xffff));
    v[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(t.x >> 16));
    v[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(t.y & 0xffff));
    v[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(t.y >> 16));
}

// sigmoid with the hardware transcendental units: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an IEEE division
__device__ __forceinline__ float sigmoidf_(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

// AdaptiveResidualMix blend (model.py:833-837): x + sigmoid(alpha) sigmoid(beta) (z - x) with the scale folded into the reciprocal:
// sigmoid(alpha) sigmoid(beta) = 1 / ((1 + e^-beta) / sigmoid(alpha)) = rcp(fma(e^-beta, inv_s, inv_s)), inv_s = 1 + e^-alpha.
// Six instructions per value (v_mul, v_exp, v_fma, v_rcp, v_sub, v_fma) instead of eight: the blends run beside MFMA streams, where
// the vector ALU is the scarce resource (tools/microbench/mb_coissue.hip).
__device__ __forceinline__ float blend_(float x, float z, float beta, float inv_s) {
    const float e = __builtin_amdgcn_exp2f(beta * -1.4426950408889634f);
    const float w = __builtin_amdgcn_rcpf(__builtin_fmaf(e, inv_s, inv_s));
    return __builtin_fmaf(w, z - x, x);
}

// ---- one K-chunk of matrix work: acc[n][pixel] += W[n][k] * X[pixel][k] ------------------------
// one 16-byte plane entry (8 channels of a 16-bit type, 4 of f32) <-> floats
template <class TT> __device__ __forceinline__ void ld_unit(const void* p, float* v) {
    if constexpr (TT::SZ == 4) {
        ld4<TT>(p, v);
    } else {
        const uint4 t = *(const uint4*)p;
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (TT::IS_BF16) {
                v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
                v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
            } else {
                v[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] & 0xffff));
                v[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] >> 16));
            }
        }
    }
}
template <class TT> __device__ __forceinline__ void st_unit(void* p, const float* v) {
    if constexpr (TT::SZ == 4) {
        st4<TT>(p, v);
    } else {
        uint4 t;
        if constexpr (TT::IS_BF16) {
            t.x = pack_bf16(v[0], v[1]); t.y = pack_bf16(v[2], v[3]); t.z = pack_bf16(v[4], v[5]); t.w = pack_bf16(v[6], v[7]);
        } else {
            t.x = pack_f16(v[0], v[1]); t.y = pack_f16(v[2], v[3]); t.z = pack_f16(v[4], v[5]); t.w = pack_f16(v[6], v[7]);
        }
        *(uint4*)p = t;
    }
}

template <class TT> __device__ __forceinline__ void mma(f32x16& acc, const u32x4& w, const u32x4& x);
template <> __device__ __forceinline__ void mma<TBF16>(f32x16& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x),
                                                  acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<TF16>(f32x16& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x),
                                                 acc, 0, 0, 0);
}
// f32: lane half h holds channels 4h..4h+3 of the 8-channel chunk; step e contracts {e, 4+e}.
template <> __device__ __forceinline__ void mma<TF32>(f32x16& acc, const u32x4& w, const u32x4& x) {
    // (bit_cast of the whole vector: __builtin_bit_cast on a single ext-vector element reads element 0)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[0], xf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[1], xf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[2], xf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[3], xf[3], acc, 0, 0, 0);
}

// LDS-DMA: 64 lanes x 16 bytes, per-lane global source, wave-uniform LDS destination.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// PyTorch's cubic convolution coefficients (A = -0.75), reference model.py:71 -> aten::upsample_bicubic2d
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    float x = t + 1.0f;
    c[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    c[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    c[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}

// ---- LDS fragment reads hidden from hipcc's waitcnt pass ----------------------------------------
// hipcc makes every ds_read it can see wait for ALL pending LDS-DMA (s_waitcnt vmcnt(0)), which would
// serialise the prefetch of the next K-stage behind the current stage's first fragment read.  The
// fragment reads of the main loop are therefore inline asm: the DMA -> read ordering is enforced by
// hand (s_waitcnt vmcnt(0) + barrier at the end of each stage) and the read -> MFMA ordering by the
// `wait_frags` statement, which names every destination as "+v" so no MFMA can be scheduled above it.
// Only lgkmcnt(0) is used, so compiler-generated scalar loads in flight cannot confuse the count.
template <int OFF> __device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16 bits");
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <int NT> struct Frags {
    u32x4 x0, x1;
    u32x4 w[NT];
};
template <int NT> __device__ __forceinline__ void wait_frags(Frags<NT>& f) {
    if constexpr (NT == 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.x0), "+v"(f.x1), "+v"(f.w[0])::"memory");
    else if constexpr (NT == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.x0), "+v"(f.x1), "+v"(f.w[0]), "+v"(f.w[1])::"memory");
    else if constexpr (NT == 3)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.x0), "+v"(f.x1), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2])::"memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.x0), "+v"(f.x1), "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
}

// ================================================================================================
// workgroup -> (pixel tile, N tile).  Consecutive logical ids land on one XCD (bijective remap of the round-robin
// dispatch), and within an XCD's contiguous id range tiles are walked in gm x gn groups: the ~32 workgroups that are
// resident on an XCD together then share gm activation tiles and gn weight tiles through that XCD's L2 instead of
// re-streaming one operand per tile of the other.  Returns false for the padding ids of a partial group.
// ================================================================================================
// x / d for 0 <= x < 2^24 with a host-provided 1.0f / d: a handful of instructions instead of the ~30 of a runtime
// integer division (the tile bookkeeping below ran five of them per workgroup)
// Every caller divides wave-uniform tile ids: the quotient is returned through readfirstlane so that it (and the
// control flow that depends on it) stays in scalar registers although the float conversion runs on the VALU.
__device__ __forceinline__ int fdiv(int x, int d, float inv) {
    int q = (int)((float)x * inv);
    const int r = x - q * d;
    q += (r >= d) - (r < 0);
    return __builtin_amdgcn_readfirstlane(q);
}

// logical id -> tile (the group walk)
__device__ __forceinline__ bool tile_of(const ConvArgs& a, int L, int& mtile, int& ntile) {
    const int gsz = a.gm * a.gn;
    const int group = fdiv(L, gsz, a.inv_gsz), within = L - group * gsz;
    const int gi_n = fdiv(group, a.groups_m, a.inv_groups_m), gi_m = group - gi_n * a.groups_m;
    const int mi = fdiv(within, a.gn, a.inv_gn), ni = within - mi * a.gn;
    mtile = gi_m * a.gm + mi;
    ntile = gi_n * a.gn + ni;
    return mtile < a.mtiles && ntile < a.ntiles;
}

// ---- the same tile arithmetic on the SCALAR unit (conv3r_kernel: its VALU issue is what the epilogue under the partner's MFMA
// stream is short of; SALU instructions issue at full rate beside MFMAs, tools/microbench/mb_coissue.hip) ----
// x / d for wave-uniform 0 <= x < 2^31, magic = floor(2^32 / d) (0xffffffff for d = 1): the estimate is at most one too small
__device__ __forceinline__ int sdiv(int x, int d, uint32_t magic) {
    const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane(x);
    uint32_t q = (uint32_t)(((unsigned long long)xs * (unsigned long long)magic) >> 32);  // s_mul_hi_u32
    q += (xs - q * (uint32_t)d >= (uint32_t)d) ? 1u : 0u;
    return (int)q;
}
__device__ __forceinline__ bool tile_of_s(const ConvArgs& a, int L, int& mtile, int& ntile) {
    const int gsz = a.gm * a.gn;
    const int group = sdiv(L, gsz, a.mg_gsz), within = L - group * gsz;
    const int gi_n = sdiv(group, a.groups_m, a.mg_groups_m), gi_m = group - gi_n * a.groups_m;
    const int mi = sdiv(within, a.gn, a.mg_gn), ni = within - mi * a.gn;
    mtile = gi_m * a.gm + mi;
    ntile = gi_n * a.gn + ni;
    return mtile < a.mtiles && ntile < a.ntiles;
}
__device__ __forceinline__ void tile_rc_s(const ConvArgs& a, int trem, int& tyi, int& txi) {
    if (!a.blk4) {
        tyi = sdiv(trem, a.tiles_x, a.mg_tiles_x);
        txi = trem - tyi * a.tiles_x;
        return;
    }
    const int bsz = 4 * a.tiles_x;
    const int br = sdiv(trem, bsz, a.mg_bsz);
    const int rem = trem - br * bsz;
    int rows = a.tiles_y - 4 * br;  // tile rows of this block row: 4, or 1..3 for the last one
    rows = rows < 4 ? rows : 4;
    txi = rows == 4 ? rem >> 2 : (rows == 3 ? (int)(((unsigned)rem * 43691u) >> 17) : (rows == 2 ? rem >> 1 : rem));
    tyi = 4 * br + rem - txi * rows;
}

// ---- 16-byte buffer store whose offset travels in an SGPR, with the wait states gfx950 needs behind it PINNED -----------------------
// A 12- / 16-byte store reads its data registers late: a vector instruction that overwrites one of them right behind the store
// corrupts what reaches memory (measured in isolation: csrc/mz_probe.hip, tests/test_store_hazard_gpu.py,
// profiles/r04_store_hazard_probe.json).  For stores with soffset 0 / global stores hipcc's hazard recogniser inserts a wait state
// itself; for THIS form (soffset in a register) it inserts none.  Every such store in the kernels goes through this helper: the
// store, then `s_nop 1` (two wait states: one more than the probe found necessary), fenced so that the scheduler cannot move
// anything between them.  tools/asm_store_hazard.py re-checks the listings of every build (tests/test_kernel_resources.py).
__device__ __forceinline__ void store16_soff(const u32x4& v, const __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, 0);
#ifndef MZ_HAZARD_DEMO  // (-DMZ_HAZARD_DEMO: the round-3 bug on purpose, for tools/debug/mix192_probe.py's "before" record)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
#endif
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class TT> __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& w, const u32x4& x);
template <> __device__ __forceinline__ void mma16<TBF16>(f32x4& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<TF16>(f32x4& acc, const u32x4& w, const u32x4& x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), acc, 0, 0, 0);
}

// the same MFMA WRITING the accumulator (C = 0 as an inline constant): the first K step of a sum needs no cleared registers
template <class TT> __device__ __forceinline__ void mma16_first(f32x4& acc, const u32x4& w, const u32x4& x) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (TT::IS_BF16) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), zero, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, x), zero, 0, 0, 0);
}

// ---- store epilogue of the 16x16x32 kernels: one 16-byte plane entry (8 channels) from a pair of accumulator fragments ----
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// SiLU of two values: the scale, the + 1 and the final product are packed-f32 instructions, the two transcendentals per value
// stay scalar.  The same operations in the same order as v * sigmoidf_(v): identical bits.
__device__ __forceinline__ f32x2 silu2(const f32x2 v) {
    const f32x2 t = v * f32x2{-1.4426950408889634f, -1.4426950408889634f};
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = f32x2{1.0f, 1.0f} + e;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return v * r;
}
template <class TT> __device__ __forceinline__ uint32_t pack_pair(const f32x2 v) {
    if constexpr (TT::IS_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));  // v_cvt_pk_bf16_f32
    else return pack_f16(v[0], v[1]);
}
// fa, fb: the two 16-channel accumulator fragments of a pair, lane (g, c) holding channels 4g..4g+3 of pixel c.  v_permlane16_swap
// between them leaves lane row g with one whole plane entry of pixel c: fragment (g & 1), plane (g >> 1) of that fragment; fa then
// holds channels 0..3 of the entry, fb channels 4..7.  IN PLACE (the accumulators are dead behind the epilogue: no register copies).
template <class TT, bool SILU> __device__ __forceinline__ u32x4 entry16(f32x4& fa, f32x4& fb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, (float)fa[j]), __builtin_bit_cast(uint32_t, (float)fb[j]), false, false);
        const uint32_t s0 = sw[0], s1 = sw[1];
        fa[j] = __builtin_bit_cast(float, s0);
        fb[j] = __builtin_bit_cast(float, s1);
    }
    f32x2 p0 = {fa[0], fa[1]}, p1 = {fa[2], fa[3]}, p2 = {fb[0], fb[1]}, p3 = {fb[2], fb[3]};
    if constexpr (SILU) { p0 = silu2(p0); p1 = silu2(p1); p2 = silu2(p2); p3 = silu2(p3); }
    u32x4 o;
    o[0] = pack_pair<TT>(p0); o[1] = pack_pair<TT>(p1); o[2] = pack_pair<TT>(p2); o[3] = pack_pair<TT>(p3);
    return o;
}

// Tile index inside an image -> (tile row, tile column).  a.blk4 = 0: row-major.  a.blk4 = 1: block rows of FOUR tile rows walked
// column by column, so that the ~32 workgroups that run on one XCD at a time (consecutive tile ids) cover a 4 x 8 block of tiles
// instead of a 1 x 32 strip: the halo rows between vertically adjacent tiles are then shared through that XCD's L2 instead of
// being fetched again a whole tile row later (input fetch 10/8 -> 34/32 of the compulsory bytes for 8-row tiles).
__device__ __forceinline__ void tile_rc(const ConvArgs& a, int trem, int& tyi, int& txi) {
    if (!a.blk4) {
        tyi = fdiv(trem, a.tiles_x, a.inv_tiles_x);
        txi = trem - tyi * a.tiles_x;
        return;
    }
    const int bsz = 4 * a.tiles_x;
    const int br = fdiv(trem, bsz, a.inv_bsz);
    const int rem = trem - br * bsz;
    int rows = a.tiles_y - 4 * br;  // tile rows of this block row: 4, or 1..3 for the last one
    rows = rows < 4 ? rows : 4;
    // rem / rows for rows in 1..4 (rem < 2^16: at most 4 * tiles_x tiles per block row)
    const int qx = rows == 4 ? rem >> 2 : (rows == 3 ? (int)(((unsigned)rem * 43691u) >> 17) : (rows == 2 ? rem >> 1 : rem));
    txi = __builtin_amdgcn_readfirstlane(qx);
    tyi = 4 * br + rem - txi * rows;
}

}  // namespace mz
