// Hardware probes behind the C ABI's mz_debug_* entry points (test infrastructure of the kernels, not part of the forward path).
//
// store_hazard_kernel: does a VALU write to a data register of a 16-byte buffer store, issued right behind that store, corrupt
// the stored data on gfx950?  Round 3 met garbage in the upper half of some of mix16b_kernel's output entries and put it down to
//     buffer_store_dwordx4 v[48:51], v183, s[16:19], s2 offen
//     v_mul_f32 v50, ...
// -- a store with an SGPR offset followed by a write to its third data register, a pair LLVM's hazard recogniser leaves alone (it
// inserts the VMEM-store-data wait state only when soffset is NOT a register: GCNHazardRecognizer::createsVALUHazard).  The
// diagnosis was an inference from an A/B of store flavours inside that kernel.  This kernel issues exactly that pair IN ISOLATION
// (hard registers, inline asm, nothing for the compiler to schedule) in every combination of
//     the store's form (buffer_store with soffset in an SGPR / soffset = 0, global_store with a 64-bit vaddr / with saddr), 0..2 wait states between the two, overwritten dword 0..3, and the follower being
//     v_mov_b32 / v_mul_f32 / v_cvt_pk_bf16_f32 / v_exp_f32 / v_pk_mul_f32 (64-bit write) / v_mfma (accumulator write)
// with eight stores back to back per loop trip (a full VMEM queue), on every CU, and a second kernel counts the 16-byte entries
// that do not hold what the registers held when the store was issued.  tests/test_store_hazard_gpu.py records the table.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mzp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t expect_word(uint32_t gw, uint32_t it, uint32_t lane, uint32_t k) {
    // never the overwrite values below (0xdeadbeef and small floats), never the buffer's pre-fill (0x11111111)
    return 0x40000000u | ((gw * 2654435761u) ^ (it * 40503u) ^ (lane << 9) ^ (k * 0x9e3779b9u)) >> 2;
}

// follower kinds
enum { F_MOV = 0, F_MUL = 1, F_CVT = 2, F_EXP = 3, F_PKMUL = 4, F_MFMA = 5 };

// One store + follower, data in v[40:43] (loaded from the compiler's registers first, then two idle wait states so that the moves
// themselves are not the hazard under test).  The follower writes v[40 + DW] (F_PKMUL: the aligned pair that contains it; F_MFMA:
// all four, as the accumulator of a 16x16x32 MFMA).
template <int FOLLOW, int FORM, int NOPS, int DW>
__device__ __forceinline__ void store_and_clobber(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t voff, u32x4 rsrc,
                                                  uint32_t soff, float fa, float fb, unsigned long long gaddr, unsigned long long sbase) {
#define MZP_MOVES "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\ts_nop 4\n\t"
#define MZP_STORE_S "buffer_store_dwordx4 v[40:43], %4, %5, %6 offen\n\t"
#define MZP_STORE_0 "buffer_store_dwordx4 v[40:43], %4, %5, 0 offen\n\t"
#define MZP_STORE_G "global_store_dwordx4 %9, v[40:43], off\n\t"
#define MZP_STORE_GS "global_store_dwordx4 %4, v[40:43], %10\n\t"
#define MZP_OPS : : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rsrc), "s"(soff), "v"(fa), "v"(fb), "v"(gaddr), "s"(sbase) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory"
    // (string pieces must be literals: one asm statement per combination, selected at compile time)
#define MZP_EMIT(STORE, NOPSTR, FOLLOWER) asm volatile(MZP_MOVES STORE NOPSTR FOLLOWER MZP_OPS)
#define MZP_FOLLOW_CASES(STORE, NOPSTR)                                                                                         \
    if constexpr (FOLLOW == F_MOV) {                                                                                            \
        if constexpr (DW == 0) MZP_EMIT(STORE, NOPSTR, "v_mov_b32 v40, 0xdeadbeef");                                            \
        else if constexpr (DW == 1) MZP_EMIT(STORE, NOPSTR, "v_mov_b32 v41, 0xdeadbeef");                                       \
        else if constexpr (DW == 2) MZP_EMIT(STORE, NOPSTR, "v_mov_b32 v42, 0xdeadbeef");                                       \
        else MZP_EMIT(STORE, NOPSTR, "v_mov_b32 v43, 0xdeadbeef");                                                              \
    } else if constexpr (FOLLOW == F_MUL) {                                                                                     \
        if constexpr (DW == 0) MZP_EMIT(STORE, NOPSTR, "v_mul_f32 v40, %7, %8");                                                \
        else if constexpr (DW == 1) MZP_EMIT(STORE, NOPSTR, "v_mul_f32 v41, %7, %8");                                           \
        else if constexpr (DW == 2) MZP_EMIT(STORE, NOPSTR, "v_mul_f32 v42, %7, %8");                                           \
        else MZP_EMIT(STORE, NOPSTR, "v_mul_f32 v43, %7, %8");                                                                  \
    } else if constexpr (FOLLOW == F_CVT) {                                                                                     \
        if constexpr (DW == 0) MZP_EMIT(STORE, NOPSTR, "v_cvt_pk_bf16_f32 v40, %7, %8");                                        \
        else if constexpr (DW == 1) MZP_EMIT(STORE, NOPSTR, "v_cvt_pk_bf16_f32 v41, %7, %8");                                   \
        else if constexpr (DW == 2) MZP_EMIT(STORE, NOPSTR, "v_cvt_pk_bf16_f32 v42, %7, %8");                                   \
        else MZP_EMIT(STORE, NOPSTR, "v_cvt_pk_bf16_f32 v43, %7, %8");                                                          \
    } else if constexpr (FOLLOW == F_EXP) {                                                                                     \
        if constexpr (DW == 0) MZP_EMIT(STORE, NOPSTR, "v_exp_f32 v40, %7");                                                    \
        else if constexpr (DW == 1) MZP_EMIT(STORE, NOPSTR, "v_exp_f32 v41, %7");                                               \
        else if constexpr (DW == 2) MZP_EMIT(STORE, NOPSTR, "v_exp_f32 v42, %7");                                               \
        else MZP_EMIT(STORE, NOPSTR, "v_exp_f32 v43, %7");                                                                      \
    } else if constexpr (FOLLOW == F_PKMUL) {                                                                                   \
        if constexpr (DW < 2) MZP_EMIT(STORE, NOPSTR, "v_pk_mul_f32 v[40:41], v[44:45], v[46:47]");                             \
        else MZP_EMIT(STORE, NOPSTR, "v_pk_mul_f32 v[42:43], v[44:45], v[46:47]");                                              \
    } else {                                                                                                                    \
        MZP_EMIT(STORE, NOPSTR, "v_mfma_f32_16x16x32_bf16 v[40:43], v[44:47], v[48:51], 0");                                    \
    }
#define MZP_NOP_CASES(STORE)                                                  \
    if constexpr (NOPS == 0) { MZP_FOLLOW_CASES(STORE, "") }                  \
    else if constexpr (NOPS == 1) { MZP_FOLLOW_CASES(STORE, "s_nop 0\n\t") }  \
    else { MZP_FOLLOW_CASES(STORE, "s_nop 1\n\t") }
    if constexpr (FORM == 1) { MZP_NOP_CASES(MZP_STORE_S) }
    else if constexpr (FORM == 0) { MZP_NOP_CASES(MZP_STORE_0) }
    else if constexpr (FORM == 2) { MZP_NOP_CASES(MZP_STORE_G) }
    else { MZP_NOP_CASES(MZP_STORE_GS) }
#undef MZP_NOP_CASES
#undef MZP_STORE_G
#undef MZP_STORE_GS
#undef MZP_MOVES
#undef MZP_STORE_S
#undef MZP_STORE_0
#undef MZP_OPS
#undef MZP_EMIT
#undef MZP_FOLLOW_CASES
}

constexpr int kWavesPerBlock = 4;
constexpr int kUnroll = 8;

// layout of the probe buffer: [global wave][iteration][lane] 16-byte entries
template <int FOLLOW, int FORM, int NOPS, int DW>
__global__ __launch_bounds__(256) void store_hazard_kernel(uint32_t* buf, int iters, uint32_t total_bytes) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t gw = blockIdx.x * kWavesPerBlock + w;
    const unsigned long long base = (unsigned long long)(uintptr_t)buf;
    u32x4 rsrc;
    rsrc[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffffu));
    rsrc[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)total_bytes);
    rsrc[3] = 0x00020000u;
    const uint32_t wave_bytes = (uint32_t)iters * 1024u;
    const unsigned long long sbase = ((unsigned long long)rsrc[0]) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32);
    const float fa = 1.5f + (float)lane, fb = 0.25f;
    asm volatile("v_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %1\n\tv_mov_b32 v47, %1\n\t"
                 "v_mov_b32 v48, %0\n\tv_mov_b32 v49, %1\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %1"
                 : : "v"(fa), "v"(fb) : "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
    for (int it0 = 0; it0 < iters; it0 += kUnroll) {
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const uint32_t it = (uint32_t)(it0 + j);
            const uint32_t in_wave = it * 1024u + lane * 16u;
            // SGPR variant: the wave's base travels in soffset (as the plane / K-step base does in the kernels); else in voffset
            const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gw * wave_bytes));
            const uint32_t voff = FORM == 1 ? in_wave : gw * wave_bytes + in_wave;
            store_and_clobber<FOLLOW, FORM, NOPS, DW>(expect_word(gw, it, lane, 0), expect_word(gw, it, lane, 1), expect_word(gw, it, lane, 2),
                                                       expect_word(gw, it, lane, 3), voff, rsrc, soff, fa, fb, base + gw * wave_bytes + in_wave, sbase);
        }
    }
}

__global__ void store_hazard_check_kernel(const uint32_t* buf, int iters, unsigned int* counts) {
    // counts[0] = entries that differ from what was stored, counts[1 + k] = ... whose dword k differs
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it) {
        const uint32_t* p = buf + ((size_t)gw * iters + it) * 256 + lane * 4;
        bool bad = false;
        for (uint32_t k = 0; k < 4; ++k) {
            if (p[k] != expect_word(gw, (uint32_t)it, lane, k)) {
                bad = true;
                atomicAdd(&counts[1 + k], 1u);
            }
        }
        if (bad) atomicAdd(&counts[0], 1u);
    }
}

template <int FOLLOW, int FORM, int NOPS>
static hipError_t run_dw(int dw, uint32_t* buf, int iters, uint32_t total, int blocks, hipStream_t s) {
    switch (dw) {
        case 0: hipLaunchKernelGGL((store_hazard_kernel<FOLLOW, FORM, NOPS, 0>), dim3(blocks), dim3(256), 0, s, buf, iters, total); break;
        case 1: hipLaunchKernelGGL((store_hazard_kernel<FOLLOW, FORM, NOPS, 1>), dim3(blocks), dim3(256), 0, s, buf, iters, total); break;
        case 2: hipLaunchKernelGGL((store_hazard_kernel<FOLLOW, FORM, NOPS, 2>), dim3(blocks), dim3(256), 0, s, buf, iters, total); break;
        case 3: hipLaunchKernelGGL((store_hazard_kernel<FOLLOW, FORM, NOPS, 3>), dim3(blocks), dim3(256), 0, s, buf, iters, total); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <int FOLLOW, int FORM>
static hipError_t run_nops(int nops, int dw, uint32_t* buf, int iters, uint32_t total, int blocks, hipStream_t s) {
    switch (nops) {
        case 0: return run_dw<FOLLOW, FORM, 0>(dw, buf, iters, total, blocks, s);
        case 1: return run_dw<FOLLOW, FORM, 1>(dw, buf, iters, total, blocks, s);
        case 2: return run_dw<FOLLOW, FORM, 2>(dw, buf, iters, total, blocks, s);
    }
    return hipErrorInvalidValue;
}
template <int FOLLOW>
static hipError_t run_off(int form, int nops, int dw, uint32_t* buf, int iters, uint32_t total, int blocks, hipStream_t s) {
    switch (form) {
        case 0: return run_nops<FOLLOW, 0>(nops, dw, buf, iters, total, blocks, s);
        case 1: return run_nops<FOLLOW, 1>(nops, dw, buf, iters, total, blocks, s);
        case 2: return run_nops<FOLLOW, 2>(nops, dw, buf, iters, total, blocks, s);
        case 3: return run_nops<FOLLOW, 3>(nops, dw, buf, iters, total, blocks, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mzp

// follower: 0 v_mov_b32, 1 v_mul_f32, 2 v_cvt_pk_bf16_f32, 3 v_exp_f32, 4 v_pk_mul_f32, 5 v_mfma_f32_16x16x32_bf16
// form: 0 = buffer_store, soffset 0; 1 = buffer_store, soffset in an SGPR; 2 = global_store, 64-bit vaddr, off; 3 = global_store, saddr;  wait_states: 0, 1 (s_nop 0), 2 (s_nop 1);  dword: 0..3 = the data register written
// counts_out[5]: entries stored wrongly, then per dword.  Returns 0, or a negative number for bad arguments / HIP errors.
extern "C" int mz_debug_store_hazard(int follower, int form, int wait_states, int dword, int iters, int blocks, unsigned int* counts_out) {
    using namespace mzp;
    if (!counts_out || iters <= 0 || iters % kUnroll != 0 || blocks <= 0 || dword < 0 || dword > 3 || form < 0 || form > 3 || wait_states < 0 || wait_states > 2) return -1;
    const size_t total = (size_t)blocks * kWavesPerBlock * iters * 1024;
    if (total >= ((size_t)1 << 32)) return -1;
    uint32_t* buf = nullptr;
    unsigned int* counts = nullptr;
    if (hipMalloc((void**)&buf, total) != hipSuccess) return -6;
    if (hipMalloc((void**)&counts, 5 * sizeof(unsigned int)) != hipSuccess) { (void)hipFree(buf); return -6; }
    int rc = 0;
    hipError_t e = hipMemset(buf, 0x11, total);
    if (e == hipSuccess) e = hipMemset(counts, 0, 5 * sizeof(unsigned int));
    if (e == hipSuccess) {
        switch (follower) {
            case F_MOV: e = run_off<F_MOV>(form, wait_states, dword, buf, iters, (uint32_t)total, blocks, 0); break;
            case F_MUL: e = run_off<F_MUL>(form, wait_states, dword, buf, iters, (uint32_t)total, blocks, 0); break;
            case F_CVT: e = run_off<F_CVT>(form, wait_states, dword, buf, iters, (uint32_t)total, blocks, 0); break;
            case F_EXP: e = run_off<F_EXP>(form, wait_states, dword, buf, iters, (uint32_t)total, blocks, 0); break;
            case F_PKMUL: e = run_off<F_PKMUL>(form, wait_states, dword, buf, iters, (uint32_t)total, blocks, 0); break;
            case F_MFMA: e = run_off<F_MFMA>(form, wait_states, dword, buf, iters, (uint32_t)total, blocks, 0); break;
            default: rc = -1;
        }
    }
    if (rc == 0 && e == hipSuccess) {
        hipLaunchKernelGGL(store_hazard_check_kernel, dim3(blocks), dim3(256), 0, 0, buf, iters, counts);
        e = hipGetLastError();
    }
    if (rc == 0 && e == hipSuccess) e = hipMemcpy(counts_out, counts, 5 * sizeof(unsigned int), hipMemcpyDeviceToHost);
    if (e != hipSuccess && rc == 0) rc = -6;
    (void)hipFree(buf);
    (void)hipFree(counts);
    return rc;
}
