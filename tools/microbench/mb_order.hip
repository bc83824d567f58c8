// Micro-benchmark: does the ORDER in which a wave walks its accumulator tile change what the chip sustains?
// A 96 x 96 wave tile as in conv3r_kernel (6 A fragments = weights x 6 B fragments = pixels, 36 accumulators of 16x16, all operands
// register resident, random bf16), one wave per SIMD, every CU busy; the 36 MFMAs of a pass are issued in different orders.  Same
// FLOPs, same registers, same instruction count: any difference is the power the operand delivery costs (fewer operand changes
// between consecutive MFMAs -> less switching -> higher clock; the chip is power-limited in this loop).
// Build: hipcc --offload-arch=gfx950 -O3 mb_order.hip -o mb_order     Output: one JSON object on stdout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                                   \
        }                                                               \
    } while (0)

// position M (0..35) of a pass -> (A fragment i, B fragment j)
template <int ORDER, int M> struct Pos {
    static constexpr int i = ORDER == 0   ? M / 6                                              // A-major raster: A held for 6, B changes every MFMA
                             : ORDER == 1 ? M / 6                                              // A-major serpentine
                             : ORDER == 2 ? M % 6                                              // B-major raster: B held for 6
                             : ORDER == 3 ? ((M / 6) & 1 ? 5 - M % 6 : M % 6)                  // B-major serpentine
                             : ORDER == 4 ? 2 * (M / 12) + (M % 12) / 6                        // conv3r before: groups of 2 A x 6 B, A-major raster
                             : ORDER == 5 ? 2 * (M / 12) + ((((M % 12) / 2) & 1) ? 1 - (M & 1) : (M & 1))   // conv3r now: B-major serpentine inside a group
                                          : 2 * (M / 12) + ((((M % 12) / 2) & 1) ? 1 - (M & 1) : (M & 1));  // 6: as 5, odd groups walk B backwards
    static constexpr int j = ORDER == 0   ? M % 6
                             : ORDER == 1 ? ((M / 6) & 1 ? 5 - M % 6 : M % 6)
                             : ORDER == 2 ? M / 6
                             : ORDER == 3 ? M / 6
                             : ORDER == 4 ? M % 6
                             : ORDER == 5 ? (M % 12) / 2
                                          : (((M / 12) & 1) ? 5 - (M % 12) / 2 : (M % 12) / 2);
};

template <int ORDER, int M> __device__ __forceinline__ void pass(f32x4 (&acc)[6][6], const u32x4 (&a)[6], const u32x4 (&b)[6]) {
    if constexpr (M < 36) {
        constexpr int i = Pos<ORDER, M>::i, j = Pos<ORDER, M>::j;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        pass<ORDER, M + 1>(acc, a, b);
    }
}

template <int ORDER>
__global__ __launch_bounds__(256) void order_loop(const u32x4* __restrict__ operands, float* sink, unsigned long long* clocks, int iters) {
    const int tid = threadIdx.x;
    u32x4 a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        a[i] = operands[(blockIdx.x * 12 + i) * 256 + tid];
        b[i] = operands[(blockIdx.x * 12 + 6 + i) * 256 + tid];
    }
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    f32x4 acc[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        pass<ORDER, 0>(acc, a, b);
        pass<ORDER, 0>(acc, a, b);
    }
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (tid == 0) {
        clocks[2 * blockIdx.x] = t1 - t0;
        clocks[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (total == 1.2345e-30f) sink[blockIdx.x * 256 + tid] = total;
}

static uint32_t rng_state = 0x9e3779b9u;
static uint32_t rng() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return rng_state;
}
static uint16_t random_bf16() {
    const float v = (float)(rng() >> 8) / 8388608.0f - 1.0f;
    uint32_t u;
    memcpy(&u, &v, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

struct Result {
    double tflops, clock_mhz;
};

template <int ORDER> static int run(int cus, const u32x4* ops, float* sink, unsigned long long* clocks, int iters, double min_seconds, Result& out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double flop_per_launch = (double)cus * 4 * iters * 72.0 * (2.0 * 16 * 16 * 32);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(order_loop<ORDER>, dim3(cus), dim3(256), 0, 0, ops, sink, clocks, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(order_loop<ORDER>, dim3(cus), dim3(256), 0, 0, ops, sink, clocks, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int n = std::max(8, (int)(min_seconds * 1e3 / std::max(ms, 1e-3f)) + 1);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(order_loop<ORDER>, dim3(cus), dim3(256), 0, 0, ops, sink, clocks, iters);
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
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    const double min_seconds = argc > 1 ? atof(argv[1]) : 1.5;
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t n_ops = (size_t)cus * 12 * 256;
    std::vector<u32x4> host(n_ops);
    for (auto& v : host)
        for (int k = 0; k < 4; ++k) v[k] = (uint32_t)random_bf16() | ((uint32_t)random_bf16() << 16);
    u32x4* ops;
    float* sink;
    unsigned long long* clocks;
    CHECK(hipMalloc((void**)&ops, n_ops * sizeof(u32x4)));
    CHECK(hipMalloc((void**)&sink, (size_t)cus * 256 * sizeof(float)));
    CHECK(hipMalloc((void**)&clocks, (size_t)cus * 2 * sizeof(unsigned long long)));
    CHECK(hipMemcpy(ops, host.data(), n_ops * sizeof(u32x4), hipMemcpyHostToDevice));
    const int iters = 9000;
    Result r[7], again[7];
    const char* names[7] = {"a_major_raster", "a_major_serpentine", "b_major_raster", "b_major_serpentine", "groups_of_2a_a_major_raster",
                            "groups_of_2a_b_major_serpentine", "groups_of_2a_b_major_serpentine_odd_groups_backwards"};
    for (int round = 0; round < 2; ++round) {  // two rounds: the second shows how far the device drifts
        Result* o = round ? again : r;
        if (run<0>(cus, ops, sink, clocks, iters, min_seconds, o[0])) return 1;
        if (run<1>(cus, ops, sink, clocks, iters, min_seconds, o[1])) return 1;
        if (run<2>(cus, ops, sink, clocks, iters, min_seconds, o[2])) return 1;
        if (run<3>(cus, ops, sink, clocks, iters, min_seconds, o[3])) return 1;
        if (run<4>(cus, ops, sink, clocks, iters, min_seconds, o[4])) return 1;
        if (run<5>(cus, ops, sink, clocks, iters, min_seconds, o[5])) return 1;
        if (run<6>(cus, ops, sink, clocks, iters, min_seconds, o[6])) return 1;
    }
    printf("{\"tile\": \"6 A x 6 B fragments of v_mfma_f32_16x16x32_bf16, register resident, random bf16, one wave per SIMD\", \"min_seconds_per_variant\": %.1f, \"orders\": {",
           min_seconds);
    for (int k = 0; k < 7; ++k)
        printf("%s\"%s\": {\"tflops\": [%.1f, %.1f], \"clock_mhz\": [%.0f, %.0f]}", k ? ", " : "", names[k], r[k].tflops, again[k].tflops, r[k].clock_mhz,
               again[k].clock_mhz);
    printf("}}\n");
    return 0;
}
