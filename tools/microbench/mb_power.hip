// Micro-benchmark: what does each ingredient of the 3x3 kernels' K loop cost this MI355X in CLOCK and POWER at the package cap?
// The MewZoom steps run at the 1400 W cap (profiles/r04_power_during_bench.txt): time per step is energy per step.  This program runs the
// conv3r work pattern in isolation -- one workgroup of eight waves per CU: team 0 (one wave per SIMD) streams v_mfma_f32_16x16x32_bf16 on a
// 6 x 6 fragment tile (144 accumulator registers, as conv3r_kernel's compute role), team 1 plays the helper -- and switches ingredients on
// one at a time:
//   L  the compute waves re-read their 12 operand fragments from LDS for every 36 MFMAs (ds_read_b128, the kernel's ratio)
//   D  the helper waves stream LDS-DMA pieces out of an L2-resident buffer (global_load_lds_dwordx4) at the kernel's rate: 86 KB per chunk of
//      5 184 cycles and CU = 2.4 one-KiB pieces per helper wave and 36-MFMA group
//   H  ... with every fifth piece out of HBM (a new kilobyte each time; the kernels' L2 hit rate is ~0.8): ~1.5 TB/s of HBM reads
//   V  the helper waves run SiLU chains (v_mul, v_exp, v_add, v_rcp, v_mul: six values per group, the 3-chunk tiles' rate)
// For every variant: >= `seconds` of back-to-back launches, MFMA TFLOP/s, in-kernel clock (s_memtime / s_memrealtime), and the package
// power sampled from sysfs / rocm-smi by a host thread.  Build: hipcc --offload-arch=gfx950 -O3 mb_power.hip -o mb_power -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

constexpr int V_L = 1, V_D = 2, V_V = 4, V_H = 8;
constexpr int FRAG_BYTES = 64 * 1024;  // operand fragments in LDS
constexpr int DUMP_BYTES = 32 * 1024;  // where the DMA pieces land
constexpr int LDS_BYTES = FRAG_BYTES + DUMP_BYTES + 64;

template <int OFF> __device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <int VAR>
__global__ __launch_bounds__(512) void power_loop(const u32x4* __restrict__ operands, const char* __restrict__ stream, size_t stream_bytes,
                                                  float* sink, unsigned long long* clocks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), team = w >> 2, wq = w & 3;
    volatile int* flag = (volatile int*)(smem + FRAG_BYTES + DUMP_BYTES);
    // random operand fragments into LDS (all waves), flag cleared
    for (int i = tid; i < FRAG_BYTES / 16; i += 512) ((u32x4*)smem)[i] = operands[(blockIdx.x % 64) * (FRAG_BYTES / 16) + i];
    if (tid == 0) *flag = 0;
    __syncthreads();
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    float total = 0.f;
    if (team == 0) {
        // ---- compute role: 36 MFMAs per group on a 6 x 6 fragment tile; with L, the 12 fragments of the NEXT group come from LDS meanwhile ----
        u32x4 a[2][6], b[2][6];
        const uint32_t base = lds_base + lane * 16;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            a[0][i] = a[1][i] = lds_read128<0>(base + i * 1024);
            b[0][i] = b[1][i] = lds_read128<0>(base + (6 + i) * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 acc[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned long long t0, r0, t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
        auto group = [&](auto cur_tag, int it) __attribute__((always_inline)) {
            constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
            const uint32_t src = base + (uint32_t)(((it * 12) & 63) * 1024);  // (rotates through the 64 KB of fragments)
#pragma unroll
            for (int m = 0; m < 36; ++m) {
                // serpentine: one operand changes per MFMA (the shipped kernels' order)
                const int j = m / 6, ii = m % 6, i = (j & 1) ? 5 - ii : ii;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[cur][i]), __builtin_bit_cast(bf16x8, b[cur][j]), acc[i][j], 0, 0, 0);
                if constexpr ((VAR & V_L) != 0) {
                    if (m % 2 == 0 && m < 24) {  // twelve reads in the group's first two thirds (landed before its closing wait), straight into the next group's operand registers
                        const int f = m / 2;
                        u32x4& dst = f < 6 ? a[nxt][f < 6 ? f : 0] : b[nxt][f < 6 ? 0 : f - 6];
                        asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(src + (uint32_t)(f * 1024)) : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr ((VAR & V_L) != 0) {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(a[nxt][0]), "+v"(a[nxt][1]), "+v"(a[nxt][2]), "+v"(a[nxt][3]), "+v"(a[nxt][4]), "+v"(a[nxt][5]),
                               "+v"(b[nxt][0]), "+v"(b[nxt][1]), "+v"(b[nxt][2]), "+v"(b[nxt][3]), "+v"(b[nxt][4]), "+v"(b[nxt][5])::"memory");
            }
        };
        for (int it = 0; it < iters; it += 2) {
            group(std::integral_constant<int, 0>{}, it);
            group(std::integral_constant<int, 1>{}, it + 1);
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        if (tid == 0) {
            clocks[2 * blockIdx.x] = t1 - t0;
            clocks[2 * blockIdx.x + 1] = r1 - r0;
            *flag = 1;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        // ---- helper role: DMA pieces and / or SiLU chains until the compute role is done ----
        float v0 = 0.3f + lane * 1e-3f, v1 = -0.7f + lane * 1e-3f, s0 = 0.f, s1 = 0.f;
        size_t off = ((size_t)blockIdx.x * 4 + wq) * 1024u * 5u;
        const size_t span = (VAR & V_H) ? stream_bytes : (size_t)(2u << 20);  // L2-resident: 2 MB shared by everybody
        int it = 0;
        unsigned long long last;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last)::"memory");
        while (*flag == 0) {
            if constexpr ((VAR & V_D) != 0) {
                // the kernel's rate: 86 pieces per chunk (5 184 cycles) on four helper waves = 2.4 pieces per wave and 576-cycle group:
                // two per iteration, four every fifth
                const int np = (it % 5 == 4) ? 4 : 2;
                for (int k = 0; k < np; ++k) {
                    // H: one piece in five comes out of HBM (the kernels' L2 hit rate is ~0.8), the others out of the L2-resident 2 MB
                    const bool far = (VAR & V_H) != 0 && ((it + k) % 5 == 0);
                    const char* src = stream + (far ? (off % span) : (off % (size_t)(2u << 20))) + lane * 16;
                    char* dst = smem + FRAG_BYTES + ((it * 4 + wq + 8 * k) & 31) * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    off += (VAR & V_H) ? (size_t)1024u * 1021u : (size_t)4096u;
                }
                if (it % 5 == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr ((VAR & V_V) != 0) {
                // six SiLU values per group-time (the 3-chunk tiles: 144 values per wave and 27 groups): x * rcp(1 + exp2(-x log2 e))
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %2\n\tv_mul_f32 %1, 0xbfb8aa3b, %3\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t"
                                 "v_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\t"
                                 "v_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %3, %1"
                                 : "=&v"(s0), "=&v"(s1) : "v"(v0), "v"(v1));
                    v0 = s0 + 0.25f; v1 = s1 - 0.25f;
                }
            }
            // pace: one iteration per 36-MFMA group of the compute role (576 core cycles)
            unsigned long long now;
            do {
                __builtin_amdgcn_s_sleep(1);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            } while (now - last < 576ull);
            last = now;
            ++it;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        total = v0 + v1;
    }
    if (total == 1.2345e-30f) sink[blockIdx.x * 512 + tid] = total;
}

static uint32_t rng_state = 0x9e3779b9u;
static uint32_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }
static uint16_t random_bf16() {
    const float v = (float)(rng() >> 8) / 8388608.0f - 1.0f;
    uint32_t u; memcpy(&u, &v, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// ---- package power: a host thread samples while a variant runs ----
static std::string g_power_file;
static double read_power_w() {
    if (!g_power_file.empty()) {
        FILE* f = fopen(g_power_file.c_str(), "r");
        if (f) { double uw = 0; int n = fscanf(f, "%lf", &uw); fclose(f); if (n == 1) return uw * 1e-6; }
    }
    FILE* p = popen("rocm-smi --showpower 2>/dev/null | grep -i 'power (W)' | head -1 | awk '{print $NF}'", "r");
    if (!p) return 0.0;
    double w = 0; int n = fscanf(p, "%lf", &w); pclose(p);
    return n == 1 ? w : 0.0;
}
static void find_power_file() {
    for (int card = 0; card < 16 && g_power_file.empty(); ++card)
        for (int hw = 0; hw < 16 && g_power_file.empty(); ++hw)
            for (const char* name : {"power1_average", "power1_input"}) {
                char path[256];
                snprintf(path, sizeof(path), "/sys/class/drm/card%d/device/hwmon/hwmon%d/%s", card, hw, name);
                FILE* f = fopen(path, "r");
                if (f) { double v; if (fscanf(f, "%lf", &v) == 1 && v > 1e6) g_power_file = path; fclose(f); }
                if (!g_power_file.empty()) break;
            }
}

struct Result { double tflops, clock_mhz, seconds, watts; };

template <int VAR> static int run(int cus, const u32x4* ops, const char* stream, size_t stream_bytes, float* sink, unsigned long long* clocks,
                                  int iters, double min_seconds, Result& out) {
    static bool ready = false;
    if (!ready) { CHECK(hipFuncSetAttribute((const void*)power_loop<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); ready = true; }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double flop_per_launch = (double)cus * 4 * iters * 36.0 * (2.0 * 16 * 16 * 32);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(power_loop<VAR>, dim3(cus), dim3(512), LDS_BYTES, 0, ops, stream, stream_bytes, sink, clocks, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(power_loop<VAR>, dim3(cus), dim3(512), LDS_BYTES, 0, ops, stream, stream_bytes, sink, clocks, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int n = std::max(8, (int)(min_seconds * 1e3 / std::max(ms, 1e-3f)) + 1);
    std::atomic<bool> stop{false};
    std::vector<double> watts;
    std::thread sampler([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(600));  // past the ramp
        while (!stop.load()) { const double w = read_power_w(); if (w > 0) watts.push_back(w); std::this_thread::sleep_for(std::chrono::milliseconds(150)); }
    });
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(power_loop<VAR>, dim3(cus), dim3(512), LDS_BYTES, 0, ops, stream, stream_bytes, sink, clocks, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    stop.store(true); sampler.join();
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * cus);
    CHECK(hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (int i = 0; i < cus; ++i) if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    std::sort(mhz.begin(), mhz.end()); std::sort(watts.begin(), watts.end());
    out.tflops = flop_per_launch * n / (ms * 1e-3) / 1e12;
    out.clock_mhz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    out.seconds = ms * 1e-3;
    out.watts = watts.empty() ? 0.0 : watts[watts.size() / 2];
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    const double min_seconds = argc > 1 ? atof(argv[1]) : 4.0;
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    find_power_file();
    const size_t n_ops = (size_t)64 * (FRAG_BYTES / 16);
    std::vector<u32x4> host(n_ops);
    for (auto& v : host) for (int k = 0; k < 4; ++k) v[k] = (uint32_t)random_bf16() | ((uint32_t)random_bf16() << 16);
    u32x4* ops; char* stream; float* sink; unsigned long long* clocks;
    const size_t stream_bytes = (size_t)2 << 30;
    CHECK(hipMalloc((void**)&ops, n_ops * sizeof(u32x4)));
    CHECK(hipMalloc((void**)&stream, stream_bytes + (1 << 20)));
    CHECK(hipMemset(stream, 0x3c, stream_bytes + (1 << 20)));
    CHECK(hipMalloc((void**)&sink, (size_t)cus * 512 * sizeof(float)));
    CHECK(hipMalloc((void**)&clocks, (size_t)cus * 2 * sizeof(unsigned long long)));
    CHECK(hipMemcpy(ops, host.data(), n_ops * sizeof(u32x4), hipMemcpyHostToDevice));
    const int iters = 40000;  // x 36 MFMAs per compute wave: ~12 ms per launch
    struct V { const char* name; int var; Result r; };
    std::vector<V> vs = {{"mfma only (helper waves asleep)", 0, {}}, {"+ LDS fragment reads", V_L, {}}, {"+ LDS-DMA from L2", V_D, {}},
                         {"+ LDS-DMA, 1 piece in 5 from HBM", V_D | V_H, {}}, {"+ helper SiLU chains", V_V, {}}, {"+ reads + DMA (L2)", V_L | V_D, {}},
                         {"+ reads + DMA (L2) + SiLU", V_L | V_D | V_V, {}}, {"+ reads + DMA (1 in 5 HBM) + SiLU", V_L | V_D | V_H | V_V, {}}};
    for (auto& v : vs) {
        int rc = 1;
        switch (v.var) {
            case 0: rc = run<0>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_L: rc = run<V_L>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_D: rc = run<V_D>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_D | V_H: rc = run<V_D | V_H>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_V: rc = run<V_V>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_L | V_D: rc = run<V_L | V_D>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_L | V_D | V_V: rc = run<V_L | V_D | V_V>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
            case V_L | V_D | V_H | V_V: rc = run<V_L | V_D | V_H | V_V>(cus, ops, stream, stream_bytes, sink, clocks, iters, min_seconds, v.r); break;
        }
        if (rc) return 1;
        fprintf(stderr, "%-34s %7.1f TFLOP/s  %5.0f MHz  %6.0f W  (%.1f s)\n", v.name, v.r.tflops, v.r.clock_mhz, v.r.watts, v.r.seconds);
    }
    printf("{\"cus\": %d, \"power_source\": \"%s\", \"seconds_per_variant\": %.1f, \"variants\": [", cus, g_power_file.empty() ? "rocm-smi" : g_power_file.c_str(), min_seconds);
    for (size_t i = 0; i < vs.size(); ++i)
        printf("%s{\"name\": \"%s\", \"tflops\": %.1f, \"clock_mhz\": %.0f, \"watts\": %.0f}", i ? ", " : "", vs[i].name, vs[i].r.tflops, vs[i].r.clock_mhz, vs[i].r.watts);
    printf("]}\n");
    return 0;
}
