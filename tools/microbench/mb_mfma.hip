// Micro-benchmark: what does THIS MI355X sustain on a bare dense bf16 MFMA loop with random operands?
// The "measured peak" denominator next to the 2.5 PFLOP/s spec figure: the chip lowers its clock
// under MFMA load (MI355X_MICROARCH.md, DVFS give-back), so no kernel reaches spec x 2.4 GHz on non-trivial data.
//
//   * operands live in registers (no LDS, no memory traffic inside the loop): the upper bound of any real kernel
//   * one workgroup of 256 (one wave per SIMD) or 512 threads (two) per CU, every CU busy
//   * v_mfma_f32_16x16x32_bf16 (the shape of the shipped 3x3 kernels) and v_mfma_f32_32x32x16_bf16
//   * operands: uniform random bf16 in [-1, 1) (zero-filled operands read ~20 % high: less switching, higher clock)
//   * >= 2 s of back-to-back launches per variant; in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
// Build: hipcc --offload-arch=gfx950 -O3 mb_mfma.hip -o mb_mfma      Output: one JSON object on stdout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

// 16 independent accumulators of 16x16 (a 64 x 64 wave tile = 4 A fragments x 4 B fragments, the register-resident part
// of the shipped kernel's per-wave tile); 16 MFMAs per iteration of the inner body, unrolled 4x
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ operands, float* sink, unsigned long long* clocks,
                                                 int iters) {
    const int tid = threadIdx.x;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(blockIdx.x * 8 + i) * 512 + tid];
        b[i] = operands[(blockIdx.x * 8 + 4 + i) * 512 + tid];
    }
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    float total = 0.f;
    if constexpr (SHAPE == 16 || SHAPE == 17) {  // 17: the same loop on v_mfma_f32_16x16x32_f16 (operands: random fp16)
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if constexpr (SHAPE == 16)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                                __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]),
                                                                               __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
            // keep the accumulators bounded without touching the matrix pipe's schedule: nothing here
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else if constexpr (SHAPE == 18) {
        // the 16x16x32 bf16 loop again, walked the way the shipped kernels walk a group since round 3: groups of two A fragments,
        // B-major, the A pair in serpentine order, odd groups backwards -- ONE operand changes per MFMA (tools/microbench/mb_order.hip:
        // +3 % over the raster order above at identical FLOPs, registers and instruction count; the chip is power-limited here)
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int grp = m >> 3, jj = (m & 7) >> 1, j = grp ? 3 - jj : jj, i = 2 * grp + ((jj & 1) ? 1 - (m & 1) : (m & 1));
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];  // the same 64 x 64 wave tile as 2 x 2 fragments of 32 x 32
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)  // 8 x 4 MFMAs of 32x32x16 = the FLOPs of 64 MFMAs of 16x16x32
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i + 2 * (u & 1)]),
                                                                            __builtin_bit_cast(bf16x8, b[j + 2 * (u & 1)]), acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) total += acc[i][j][k];
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (tid == 0) {
        clocks[2 * blockIdx.x] = t1 - t0;
        clocks[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (total == 1.2345e-30f) sink[blockIdx.x * 512 + tid] = total;  // keeps the loop alive, never true in practice
}

static uint32_t rng_state = 0x9e3779b9u;
static uint32_t rng() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return rng_state;
}
static uint16_t random_bf16() {  // uniform in [-1, 1): sign, exponent and mantissa all vary
    const float v = (float)(rng() >> 8) / 8388608.0f - 1.0f;
    uint32_t u;
    memcpy(&u, &v, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static uint16_t random_f16() {  // uniform in [-1, 1) as IEEE half (round to nearest even)
    const float v = (float)(rng() >> 8) / 8388608.0f - 1.0f;
    const _Float16 h = (_Float16)v;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

struct Result {
    double tflops, clock_mhz, seconds;
    int launches;
};

template <int SHAPE> static int run(int threads, int cus, const u32x4* ops, float* sink, unsigned long long* clocks, int iters,
                                    double min_seconds, Result& out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double flop_per_launch = (double)cus * (threads / 64) * iters * 64.0 * (2.0 * 16 * 16 * 32);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(cus), dim3(threads), 0, 0, ops, sink, clocks, iters);
    CHECK(hipDeviceSynchronize());
    // size the timed batch from a probe launch, then time >= min_seconds of back-to-back launches
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(cus), dim3(threads), 0, 0, ops, sink, clocks, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int n = std::max(8, (int)(min_seconds * 1e3 / std::max(ms, 1e-3f)) + 1);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(cus), dim3(threads), 0, 0, ops, sink, clocks, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * cus);
    CHECK(hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (int i = 0; i < cus; ++i)
        if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end());
    out.tflops = flop_per_launch * n / (ms * 1e-3) / 1e12;
    out.clock_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    out.seconds = ms * 1e-3;
    out.launches = n;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    const double min_seconds = argc > 1 ? atof(argv[1]) : 2.0;
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const size_t n_ops = (size_t)cus * 8 * 512;
    std::vector<u32x4> host(n_ops);
    for (auto& v : host)
        for (int k = 0; k < 4; ++k) v[k] = (uint32_t)random_bf16() | ((uint32_t)random_bf16() << 16);
    u32x4* ops;
    float* sink;
    unsigned long long* clocks;
    CHECK(hipMalloc((void**)&ops, n_ops * sizeof(u32x4)));
    CHECK(hipMalloc((void**)&sink, (size_t)cus * 512 * sizeof(float)));
    CHECK(hipMalloc((void**)&clocks, (size_t)cus * 2 * sizeof(unsigned long long)));
    CHECK(hipMemcpy(ops, host.data(), n_ops * sizeof(u32x4), hipMemcpyHostToDevice));
    const int iters = 20000;  // x 64 MFMAs of 16x16x32 per wave = ~20 ms per launch
    Result r16x1, r16x2, r32x1, r32x2;
    if (run<16>(256, cus, ops, sink, clocks, iters, min_seconds, r16x1)) return 1;
    if (run<16>(512, cus, ops, sink, clocks, iters, min_seconds, r16x2)) return 1;
    Result s16x1;
    if (run<18>(256, cus, ops, sink, clocks, iters, min_seconds, s16x1)) return 1;
    if (run<32>(256, cus, ops, sink, clocks, iters, min_seconds, r32x1)) return 1;
    if (run<32>(512, cus, ops, sink, clocks, iters, min_seconds, r32x2)) return 1;
    // the fp16 shape of the same loop, on random fp16 operands (does the chip clock the two 16-bit types alike?)
    for (auto& v : host)
        for (int k = 0; k < 4; ++k) v[k] = (uint32_t)random_f16() | ((uint32_t)random_f16() << 16);
    CHECK(hipMemcpy(ops, host.data(), n_ops * sizeof(u32x4), hipMemcpyHostToDevice));
    Result h16x1, h16x2;
    if (run<17>(256, cus, ops, sink, clocks, iters, min_seconds, h16x1)) return 1;
    if (run<17>(512, cus, ops, sink, clocks, iters, min_seconds, h16x2)) return 1;
    const double best = std::max(std::max(std::max(r16x1.tflops, r16x2.tflops), std::max(r32x1.tflops, r32x2.tflops)), s16x1.tflops);
    printf("{\"device\": \"%s\", \"cus\": %d, \"operands\": \"uniform random bf16 in [-1,1), register resident\", "
           "\"min_seconds_per_variant\": %.1f, \"measured_peak_tflops\": %.1f, \"variants\": {"
           "\"16x16x32_1wave_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}, "
           "\"16x16x32_2waves_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}, "
           "\"16x16x32_serpentine_1wave_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}, "
           "\"32x32x16_1wave_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}, "
           "\"32x32x16_2waves_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}, "
           "\"f16_16x16x32_1wave_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}, "
           "\"f16_16x16x32_2waves_per_simd\": {\"tflops\": %.1f, \"clock_mhz\": %.0f, \"seconds\": %.2f}}}\n",
           prop.name, cus, min_seconds, best, r16x1.tflops, r16x1.clock_mhz, r16x1.seconds, r16x2.tflops, r16x2.clock_mhz,
           r16x2.seconds, s16x1.tflops, s16x1.clock_mhz, s16x1.seconds, r32x1.tflops, r32x1.clock_mhz, r32x1.seconds, r32x2.tflops, r32x2.clock_mhz, r32x2.seconds, h16x1.tflops,
           h16x1.clock_mhz, h16x1.seconds, h16x2.tflops, h16x2.clock_mhz, h16x2.seconds);
    return 0;
}
