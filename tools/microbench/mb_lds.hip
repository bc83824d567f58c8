// Micro-benchmark: how much LDS time does LDS-DMA (global_load_lds) cost next to ds_read_b128 traffic?
// One workgroup per CU: NR reader waves stream ds_read_b128, NL loader waves fill LDS from an L2-hot buffer,
// either by LDS-DMA or by global_load + ds_write_b128.  Build: hipcc --offload-arch=gfx950 -O3 mb_lds.hip -o mb_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// mode bit0: readers active, bit1: loaders use LDS-DMA, bit2: loaders use register staging
__global__ __launch_bounds__(1024) void mb(const char* src, unsigned* sink, int iters, int mode, int nread, int nload, int src_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned acc = 0;
    if (w < nread) {
        if (!(mode & 1)) return;
        // each reader streams its own 16 KiB window of LDS (conflict-free 1 KiB reads)
        const uint32_t a = lds_base + (w & 7) * 16384 + lane * 16;
        for (int it = 0; it < iters; ++it) {
            u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                         "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                         "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
            acc += v0[0] ^ v1[1] ^ v2[2] ^ v3[3] ^ v4[0] ^ v5[1] ^ v6[2] ^ v7[3];
        }
    } else if (w < nread + nload) {
        const int li = w - nread;
        char* dst = smem + 131072 + li * 8192;   // loaders fill a separate 8 KiB window each
        const char* s0 = src + (size_t)blockIdx.x % 4 * 0 + li * 8192 + lane * 16;
        if (mode & 2) {
            for (int it = 0; it < iters; ++it) {
                const char* s = s0 + (size_t)((it * 8192) % src_bytes);
#pragma unroll
                for (int j = 0; j < 8; ++j) glds16(s + j * 1024, dst + j * 1024);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode & 4) {
            for (int it = 0; it < iters; ++it) {
                const char* s = s0 + (size_t)((it * 8192) % src_bytes);
                uint4 r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = *(const uint4*)(s + j * 1024);
#pragma unroll
                for (int j = 0; j < 8; ++j) *(uint4*)(dst + j * 1024 + lane * 16) = r[j];
            }
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

int main() {
    const int src_bytes = 1 << 20;  // 1 MiB: L2-resident
    char* src; unsigned* sink;
    hipMalloc(&src, src_bytes + 65536); hipMemset(src, 1, src_bytes + 65536);
    hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void*)mb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char* name; int mode, nread, nload; } cfgs[] = {
        {"read only, 8 waves", 1, 8, 0}, {"read only, 4 waves", 1, 4, 0},
        {"DMA only, 1 loader", 2, 8, 1}, {"DMA only, 2 loaders", 2, 8, 2}, {"DMA only, 4 loaders", 2, 8, 4}, {"DMA only, 8 loaders", 2, 8, 8},
        {"regstage only, 1 loader", 4, 8, 1}, {"regstage only, 4 loaders", 4, 8, 4},
        {"read 8 + DMA 1", 3, 8, 1}, {"read 8 + DMA 2", 3, 8, 2}, {"read 8 + DMA 4", 3, 8, 4},
        {"read 8 + regstage 1", 5, 8, 1}, {"read 8 + regstage 4", 5, 8, 4},
    };
    const int iters = 20000;
    for (auto& c : cfgs) {
        const int threads = 64 * (c.nread + c.nload);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mb, dim3(256), dim3(threads), 160 * 1024, 0, src, sink, iters, c.mode, c.nread, c.nload, src_bytes);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double rd = (c.mode & 1) ? (double)c.nread * iters * 8192 : 0;   // bytes per CU
        const double ld = (c.mode & 6) ? (double)c.nload * iters * 8192 : 0;
        printf("%-28s %8.3f ms | per CU: read %7.1f GB/s  fill %6.1f GB/s | (whole kernel time: slower side dominates)\n", c.name, ms,
               rd / (ms * 1e-3) / 1e9, ld / (ms * 1e-3) / 1e9);
    }
    return 0;
}
