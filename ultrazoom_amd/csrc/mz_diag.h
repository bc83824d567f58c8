// Diagnostic hooks (in-kernel cycle stamps).  The shipped build defines every hook as a no-op; -DMZ_DIAG builds
// (tools/build_variant.sh diag -DMZ_DIAG) accumulate s_memtime differences per wave and dump them into ConvArgs::dbg
// (MZ_DEBUG_STAMPS=1 allocates it; mz_debug_read() copies it out; tools/stamp_probe_r.py prints it).
#pragma once
#ifdef MZ_DIAG
#define RS_DECL                                                                                              \
    uint32_t rs_[32] = {};                                                                                   \
    uint32_t rs_t0 = 0, rs_t1 = 0;                                                                           \
    unsigned long long rs_c0, rs_r0;                                                                         \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(rs_c0), "=s"(rs_r0)::"memory")
#define RS_NOW(v)                                                                         \
    do {                                                                                  \
        unsigned long long t__;                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");       \
        v = (uint32_t)t__;                                                                \
    } while (0)
#define RS_BEGIN() RS_NOW(rs_t0)
// adds the cycles since the previous RS_BEGIN / RS_LAP to counter i, counts the event in counter j (j < 0: not counted)
#define RS_LAP(i)                 \
    do {                          \
        RS_NOW(rs_t1);            \
        rs_[i] += rs_t1 - rs_t0;  \
        rs_t0 = rs_t1;            \
    } while (0)
#define RS_COUNT(i) do { rs_[i] += 1; } while (0)
#define RS_FENCE() __builtin_amdgcn_sched_barrier(0)
#define RS_DUMP()                                                                                              \
    do {                                                                                                       \
        unsigned long long rs_c1, rs_r1;                                                                       \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(rs_c1), "=s"(rs_r1)::"memory"); \
        rs_[28] = (uint32_t)(rs_c1 - rs_c0); /* shader-clock cycles of this wave's life */                         \
        rs_[29] = (uint32_t)(rs_r1 - rs_r0); /* the same in 100 MHz ticks */                                       \
        if (a.dbg && blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) {                                               \
            _Pragma("unroll") for (int i__ = 0; i__ < 32; ++i__) a.dbg[w * 32 + i__] = rs_[i__];              \
        }                                                                                                      \
    } while (0)
#else
#define RS_DECL do { } while (0)
#define RS_BEGIN() do { } while (0)
#define RS_LAP(i) do { } while (0)
#define RS_COUNT(i) do { } while (0)
#define RS_FENCE() do { } while (0)
#define RS_DUMP() do { } while (0)
#endif
