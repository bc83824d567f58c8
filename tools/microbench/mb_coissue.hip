// Micro-benchmark: what can the SECOND wave of a SIMD issue while the first runs a back-to-back MFMA stream?
// (design input of conv3r_kernel: a tile's epilogue runs on wave w + 4 under the K loop of wave w.)
// One workgroup of 512 threads per CU.  Waves 0-3: a register-resident v_mfma_f32_16x16x32_bf16 loop on random operands
// (16 independent accumulators).  Waves 4-7: `reps` repetitions of a block of 32 instructions of one kind (variant), started
// behind the same barrier; they stop long before the MFMA waves do.  Reported per variant: cycles per instruction of the
// VALU wave with the partner's MFMAs running (and alone, mfma_iters = 0), and cycles per MFMA of the MFMA wave while the
// partner was active (its first `probe` iterations) against its undisturbed rate.
// Build: hipcc --offload-arch=gfx950 -O3 mb_coissue.hip -o mb_coissue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define R8(x) x x x x x x x x
#define R4(x) x x x x
template <int V> __device__ __forceinline__ void block32(float (&r)[16], u32x4& sv, uint32_t* gp, uint32_t lds, uint32_t lds0) {
    // 32 instructions per call
    if constexpr (V == 1) {        // independent v_exp_f32
        R4(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
    } else if constexpr (V == 2) { // dependent v_exp_f32 chain
        R4(asm volatile(R8("v_exp_f32 %0, %0\n") : "+v"(r[0]));)
    } else if constexpr (V == 3) { // independent v_pk_mul_f32
        R4(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                        "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                        : "+v"(*(f32x2*)&r[0]), "+v"(*(f32x2*)&r[2]), "+v"(*(f32x2*)&r[4]), "+v"(*(f32x2*)&r[6]) : "v"(*(f32x2*)&r[8]));)
    } else if constexpr (V == 4) { // independent v_mul_f32
        R4(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]));)
    } else if constexpr (V == 5) { // dependent v_mul_f32 chain
        R4(asm volatile(R8("v_mul_f32 %0, %0, %1\n") : "+v"(r[0]) : "v"(r[8]));)
    } else if constexpr (V == 6) { // v_permlane16_swap
        R4(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                        "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
    } else if constexpr (V == 7) { // independent v_rcp_f32
        R4(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
    } else if constexpr (V == 8) { // v_cvt_pk_bf16_f32
        R4(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7\n"
                        "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
    } else if constexpr (V == 9) { // s_nop-free SALU block: s_add
        R4(asm volatile(R8("s_add_u32 s20, s20, 1\n") ::: "s20", "scc");)
    } else if constexpr (V == 10) { // 16-byte stores (4 per block, 28 v_mov between: counted as 32 instructions)
        R4(asm volatile("global_store_dwordx4 %0, %1, off\n" R4("v_mov_b32 %2, %2\n") "v_mov_b32 %2, %2\n v_mov_b32 %2, %2\n v_mov_b32 %2, %2"
                        : : "v"(gp), "v"(sv), "v"(r[0]) : "memory");)
    } else if constexpr (V == 11) { // ds_read_b128 x 8 (LDS fragment reads)
        R4(asm volatile(R8("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(sv) : "v"(lds) : "memory");)
    } else if constexpr (V == 13) { // v_mov_b64
        R4(asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4\n v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4"
                        : "+v"(*(f32x2*)&r[0]), "+v"(*(f32x2*)&r[2]), "+v"(*(f32x2*)&r[4]), "+v"(*(f32x2*)&r[6]) : "v"(*(f32x2*)&r[8]));)
    } else if constexpr (V == 14) { // v_pk_fma_f32
        R4(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                        "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                        : "+v"(*(f32x2*)&r[0]), "+v"(*(f32x2*)&r[2]), "+v"(*(f32x2*)&r[4]), "+v"(*(f32x2*)&r[6]) : "v"(*(f32x2*)&r[8]));)
    } else if constexpr (V == 15) { // v_fma_f32
        R4(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]));)
    } else if constexpr (V == 16) { // v_pk_mul_f16
        R4(asm volatile("v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]));)
    } else if constexpr (V == 17) { // v_add_u32 / v_cndmask mix
        R4(asm volatile("v_add_u32 %0, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_add_u32 %2, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_lshlrev_b32 %4, 1, %4\n v_add3_u32 %5, %5, %8, %8\n v_and_b32 %6, %6, %8\n v_mad_u32_u24 %7, %7, %8, %8"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]) : "vcc");)
    } else if constexpr (V == 18) { // buffer-less LDS-DMA: global_load_lds_dwordx4 x 2 per 8 (+ 6 v_mov)
        R4(asm volatile("s_mov_b32 m0, %2\n global_load_lds_dwordx4 %0, off\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1\n global_load_lds_dwordx4 %0, off\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1\n v_mov_b32 %1, %1"
                        : : "v"(gp), "v"(r[0]), "s"(lds0) : "memory", "m0");)
    } else if constexpr (V == 19) { // v_exp_f16 (two per 32-bit register would need v_pk: there is none; rate check only)
        R4(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
    } else if constexpr (V == 12) { // v_mov_b32 independent
        R4(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]));)
    }
}

template <int V>
__global__ __launch_bounds__(512) void co_kernel(const u32x4* __restrict__ operands, float* sink, unsigned long long* out, int mfma_iters,
                                                 int reps, uint32_t* scratch) {
    __shared__ u32x4 lds_buf[512];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    lds_buf[tid] = operands[tid];
    __syncthreads();
    if (w < 4) {
        u32x4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = operands[(blockIdx.x * 8 + i) * 512 + tid];
            b[i] = operands[(blockIdx.x * 8 + 4 + i) * 512 + tid];
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = now();
        unsigned long long tp = t0;
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
            if (it == mfma_iters / 4 - 1) tp = now();  // the partner is still running during the first quarter
        }
        const unsigned long long t1 = now();
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (total == 123.456f) sink[tid] = total;
        if (lane == 0 && blockIdx.x == gridDim.x / 2) { out[w * 4 + 0] = tp - t0; out[w * 4 + 1] = t1 - tp; }
    } else {
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = 0.001f * (float)(lane + i + 1);
        u32x4 sv = operands[tid];
        uint32_t* gp = scratch + ((size_t)blockIdx.x * 512 + tid) * 4;
        const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_buf + lane * 16;
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = now();
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_buf + (w - 4) * 1024);
        for (int it = 0; it < reps; ++it) block32<V>(r, sv, gp, lds, lds0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = now();
        float total = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) total += r[i];
        if (total == 123.456f) sink[tid] = total + (float)sv[0];
        if (lane == 0 && blockIdx.x == gridDim.x / 2) out[w * 4 + 0] = t1 - t0;
    }
}

template <int V> static int run(const char* name, const u32x4* ops, float* sink, unsigned long long* out, uint32_t* scratch, int cus) {
    const int reps = 400;
    unsigned long long h[32];
    double res[2][3];
    for (int with = 0; with < 2; ++with) {
        const int iters = with ? 4000 : 0;
        for (int k = 0; k < 3; ++k) {
            hipLaunchKernelGGL((co_kernel<V>), dim3(cus), dim3(512), 0, 0, ops, sink, out, iters, V ? reps : 0, scratch);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        res[with][0] = V ? (double)h[4 * 4] / (reps * 32.0) : 0.0;                 // cycles per instruction, wave 4
        res[with][1] = with ? (double)h[0] / (iters / 4 * 16.0) : 0.0;            // cycles per MFMA, first quarter (partner active)
        res[with][2] = with ? (double)h[1] / ((iters - iters / 4) * 16.0) : 0.0;  // ... rest (partner done)
    }
    printf("  {\"variant\": \"%s\", \"cyc_per_instr_alone\": %.2f, \"cyc_per_instr_under_mfma\": %.2f, \"cyc_per_mfma_partner_active\": %.2f, \"cyc_per_mfma_partner_done\": %.2f},\n",
           name, res[0][0], res[1][0], res[1][1], res[1][2]);
    return 0;
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    std::vector<uint32_t> host((size_t)cus * 8 * 512 * 4);
    uint32_t s = 12345u;
    for (auto& v : host) {
        auto nxt = [&]() { s = s * 1664525u + 1013904223u; return (uint16_t)(0x3c00u + ((s >> 9) & 0x3ffu) - ((s >> 20) & 1u) * 0x100u) ^ (uint16_t)((s >> 5) & 0x8000u); };
        const uint16_t lo = nxt(), hi = nxt();
        v = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    u32x4* ops; float* sink; unsigned long long* out; uint32_t* scratch;
    CHECK(hipMalloc(&ops, host.size() * 4)); CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&out, 256)); CHECK(hipMalloc(&scratch, (size_t)cus * 512 * 16));
    CHECK(hipMemcpy(ops, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(out, 0, 256));
    printf("[\n");
    run<0>("none", ops, sink, out, scratch, cus);
    run<1>("v_exp_f32 independent", ops, sink, out, scratch, cus);
    run<2>("v_exp_f32 dependent chain", ops, sink, out, scratch, cus);
    run<7>("v_rcp_f32 independent", ops, sink, out, scratch, cus);
    run<3>("v_pk_mul_f32 independent", ops, sink, out, scratch, cus);
    run<4>("v_mul_f32 independent", ops, sink, out, scratch, cus);
    run<5>("v_mul_f32 dependent chain", ops, sink, out, scratch, cus);
    run<12>("v_mov_b32 independent", ops, sink, out, scratch, cus);
    run<6>("v_permlane16_swap_b32", ops, sink, out, scratch, cus);
    run<8>("v_cvt_pk_bf16_f32", ops, sink, out, scratch, cus);
    run<13>("v_mov_b64", ops, sink, out, scratch, cus);
    run<14>("v_pk_fma_f32", ops, sink, out, scratch, cus);
    run<15>("v_fma_f32", ops, sink, out, scratch, cus);
    run<16>("v_pk_mul_f16", ops, sink, out, scratch, cus);
    run<17>("integer VALU mix", ops, sink, out, scratch, cus);
    run<19>("v_exp_f16", ops, sink, out, scratch, cus);
    run<18>("global_load_lds_dwordx4 x2 + 6 v_mov (per 8 instructions)", ops, sink, out, scratch, cus);
    run<9>("s_add_u32", ops, sink, out, scratch, cus);
    run<10>("global_store_dwordx4 + 7 v_mov (per 8 instructions)", ops, sink, out, scratch, cus);
    run<11>("ds_read_b128 x8 + wait", ops, sink, out, scratch, cus);
    printf("]\n");
    return 0;
}
