// stamps.hpp -- diagnostic builds only (-DZK_STAMPS: make libzkhip_diag.so): in-kernel s_memtime stamps into a buffer of their own;
// the macros expand to nothing in the product build.
#pragma once
namespace zk {
#ifdef ZK_STAMPS   // diagnostic build only (make libzkhip_diag.so): in-kernel s_memtime stamps into a buffer of their own
__device__ unsigned long long g_zk_stamps[64 * 8];
#define ZK_STAMP(slot)                                                                              \
    do {                                                                                            \
        if (threadIdx.x == 0) {                                                                     \
            unsigned long long _t;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
            g_zk_stamps[(round & 63) * 8 + (slot)] = _t;                                            \
        }                                                                                           \
    } while (0)
// the same from any one thread, filed under an explicit row (the serial kernel: row = round, rows 40.. = per-kernel marks)
#define ZK_STAMP_AT(tid, row, slot)                                                                 \
    do {                                                                                            \
        if (threadIdx.x == (tid)) {                                                                 \
            unsigned long long _t;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
            g_zk_stamps[((row) & 63) * 8 + (slot)] = _t;                                            \
        }                                                                                           \
    } while (0)
#else
#define ZK_STAMP(slot) do { } while (0)
#define ZK_STAMP_AT(tid, row, slot) do { } while (0)
#endif
}  // namespace zk
