// Micro-benchmark: what does THIS MI355X sustain on the memory pattern of AdaptiveResidualMix (read x, read z, write out; tensors
// far larger than L2 + Infinity Cache), and how much of it does the PATTERN cost?  The denominator for mix16_kernel's HBM-roofline
// fraction.  Tensors in the engine's plane-major layout: [image][plane of 8 channels][pixel][16 bytes].
//   linear     every wave moves 1-KiB contiguous runs (the best case of any layout)
//   planes256  mix16_kernel's pattern: a 256-pixel workgroup tile, a load instruction = 4 planes x 16 pixels (4 runs of 256 bytes),
//              24 such loads per wave and tensor issued DEPTH K-steps ahead; stores of 256-byte runs
//   planes1k   the same tile, but a load instruction = 64 consecutive pixels of ONE plane (1-KiB runs): what a layout-aware
//              variant of the kernel (cross-lane transposition into MFMA operands) would see
// Build: hipcc --offload-arch=gfx950 -O3 mb_stream.hip -o mb_stream      Output: one JSON object per line.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

__device__ __forceinline__ u32x4 mixv(u32x4 a, u32x4 b) { return u32x4{a[0] + b[0], a[1] ^ b[1], a[2] + b[2], a[3] ^ b[3]}; }

__global__ __launch_bounds__(256) void linear_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ z, u32x4* __restrict__ out,
                                                     long long n) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
        u32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = i + 256 * u;
            a[u] = j < n ? __builtin_nontemporal_load(x + j) : u32x4{0, 0, 0, 0};
            b[u] = j < n ? __builtin_nontemporal_load(z + j) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = i + 256 * u;
            if (j < n) out[j] = mixv(a[u], b[u]);
        }
    }
}

// P planes per image; tile = 256 consecutive pixels of one image (hw % 256 == 0 here); 8 waves
template <int P, int DEPTH, bool RUN1K>
__global__ __launch_bounds__(512) void planes_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ z, u32x4* __restrict__ out,
                                                     long long hw, int tiles_per_image) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int img = blockIdx.x / tiles_per_image, t = blockIdx.x - img * tiles_per_image;
    const long long base = (long long)img * P * hw + (long long)t * 256;
    constexpr int KS = P / 4;  // K steps of 4 planes per tensor
    u32x4 acc[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    u32x4 buf[DEPTH][2][2];
    auto addr = [&](int ks, int pf) -> long long {
        if constexpr (RUN1K) {
            // load (ks, pf) of wave w: plane 4 ks + (2 pf + (w >> 2)) ... every wave reads whole 1-KiB runs: plane q, pixels 64 (w & 3) + lane
            const int q = 4 * ks + 2 * pf + (w >> 2);
            return base + (long long)q * hw + 64 * (w & 3) + lane;
        } else {
            return base + (long long)(4 * ks + g) * hw + 32 * w + 16 * pf + c;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            buf[d][pf][0] = x[addr(d, pf)];
            buf[d][pf][1] = z[addr(d, pf)];
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) acc[pf] = mixv(acc[pf], mixv(buf[ks % DEPTH][pf][0], buf[ks % DEPTH][pf][1]));
        if (ks + DEPTH < KS) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                buf[ks % DEPTH][pf][0] = x[addr(ks + DEPTH, pf)];
                buf[ks % DEPTH][pf][1] = z[addr(ks + DEPTH, pf)];
            }
        }
    }
    // stores: P planes x 256 pixels per tile = P / 4 x 2 stores of 16 bytes per lane
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            u32x4 v = acc[pf];
            v[0] += ks;
            out[addr(ks, pf)] = v;
        }
}

template <class F> static int time_it(const char* name, double bytes, int reps, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"variant\": \"%s\", \"ms_per_launch\": %.4f, \"TB_per_s\": %.3f}\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
    return 0;
}

int main() {
    constexpr int P = 24;  // C = 192
    const int B = 3;
    const long long hw = 540LL * 960;  // 518400 = 2025 x 256
    const long long n = (long long)B * P * hw;  // 16-byte units per tensor
    const double bytes = 3.0 * n * 16.0;
    u32x4 *x, *z, *out;
    CHECK(hipMalloc(&x, n * 16));
    CHECK(hipMalloc(&z, n * 16));
    CHECK(hipMalloc(&out, n * 16));
    CHECK(hipMemset(x, 1, n * 16));
    CHECK(hipMemset(z, 2, n * 16));
    const int tpi = (int)(hw / 256);
    const int reps = 20;
    if (time_it("linear", bytes, reps, [&] { hipLaunchKernelGGL(linear_kernel, dim3(256 * 8), dim3(256), 0, 0, x, z, out, n); })) return 1;
    if (time_it("linear, grid 256 x 32", bytes, reps, [&] { hipLaunchKernelGGL(linear_kernel, dim3(256 * 32), dim3(256), 0, 0, x, z, out, n); })) return 1;
    if (time_it("planes256 depth 2", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 2, false>), dim3(B * tpi), dim3(512), 0, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes256 depth 3", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 3, false>), dim3(B * tpi), dim3(512), 0, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes256 depth 6", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 6, false>), dim3(B * tpi), dim3(512), 0, 0, x, z, out, hw, tpi); })) return 1;
    // the same with 96 KB of (unused) LDS per workgroup: ONE workgroup per CU, mix16_kernel's occupancy
    CHECK(hipFuncSetAttribute((const void*)planes_kernel<P, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CHECK(hipFuncSetAttribute((const void*)planes_kernel<P, 6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CHECK(hipFuncSetAttribute((const void*)planes_kernel<P, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CHECK(hipFuncSetAttribute((const void*)planes_kernel<P, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CHECK(hipFuncSetAttribute((const void*)planes_kernel<P, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    if (time_it("planes256 depth 1, one workgroup per CU (32 KB of loads in flight per CU: mix16_kernel's)", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 1, false>), dim3(B * tpi), dim3(512), 96 * 1024, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes256 depth 2, one workgroup per CU", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 2, false>), dim3(B * tpi), dim3(512), 96 * 1024, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes256 depth 3, one workgroup per CU", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 3, false>), dim3(B * tpi), dim3(512), 96 * 1024, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes256 depth 6, one workgroup per CU", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 6, false>), dim3(B * tpi), dim3(512), 96 * 1024, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes1k depth 6, one workgroup per CU", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 6, true>), dim3(B * tpi), dim3(512), 96 * 1024, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes1k depth 2", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 2, true>), dim3(B * tpi), dim3(512), 0, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes1k depth 3", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 3, true>), dim3(B * tpi), dim3(512), 0, 0, x, z, out, hw, tpi); })) return 1;
    if (time_it("planes1k depth 6", bytes, reps, [&] { hipLaunchKernelGGL((planes_kernel<P, 6, true>), dim3(B * tpi), dim3(512), 0, 0, x, z, out, hw, tpi); })) return 1;
    return 0;
}
