// ubench_rows.hip -- pure-load model of the k-variable fold's access pattern on a 2^24-entry table (512 MiB): a wave owns 64
// consecutive outputs (2 KiB per row) and reads them in every one of R = 2^k rows that lie n/R entries apart (8 MiB at k = 6).
// Variants: load layout (32 B per lane as two dwordx4 16 B apart / 16 B per lane covering 1 KiB per instruction), loads in flight,
// tiles per wave, waves per workgroup, and the ORDER in which a wave walks the rows (all waves in step, or rotated per tile).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_rows tools/ubench_rows.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// LAYOUT 0: lane owns a whole 32-byte entry (two dwordx4, 32 B lane stride); 1: lane (n, h) owns bytes [16 h, 16 h + 16) of entries n and 32 + n
template <int U, int LAYOUT, bool NT>
__global__ __launch_bounds__(1024) void k_rows(const unsigned char* __restrict__ in, size_t m, uint32_t log_rows, uint32_t tpw, uint32_t rot,
                                               u32x4* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t rows = 1u << log_rows;
    const size_t tile0 = ((size_t)blockIdx.x * n_waves + wave) * tpw;
    const size_t row = 32 * m;
    const unsigned char* p = in + 2048 * tile0 + (LAYOUT == 0 ? 32 * lane : 32 * (lane & 31) + 16 * (lane >> 5));
    const uint32_t second = LAYOUT == 0 ? 16 : 1024;
    const uint32_t start = rot ? (uint32_t)((tile0 * rot) & (rows - 1)) : 0;          // row the wave starts at
    const uint32_t steps = tpw * rows;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t s = 0; s < steps; s += U) {
        u32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t st = s + u;
            const unsigned char* q = p + (size_t)((st + start) & (rows - 1)) * row + (size_t)(st >> log_rows) * 2048;
            if (NT) { a[u] = __builtin_nontemporal_load((const u32x4*)q); b[u] = __builtin_nontemporal_load((const u32x4*)(q + second)); }
            else { a[u] = *(const u32x4*)q; b[u] = *(const u32x4*)(q + second); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= a[u] ^ b[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// sequential-sum model (fine_sums_kernel): a wave reads one contiguous piece of KB KiB; piece index = identity or a stride
// permutation that puts consecutive waves (which run at the same time) n / S entries apart
template <int KB>
__global__ __launch_bounds__(256) void k_pieces(const unsigned char* __restrict__ in, uint32_t n_pieces, uint32_t log_s, u32x4* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_pieces) return;
    const uint32_t s = 1u << log_s;
    const uint32_t piece = log_s ? (w & (s - 1)) * (n_pieces >> log_s) + (w >> log_s) : w;
    const unsigned char* p = in + (size_t)piece * KB * 1024 + 32 * lane;
    u32x4 a[KB], b[KB];
#pragma unroll
    for (int u = 0; u < KB / 2; ++u) { a[u] = __builtin_nontemporal_load((const u32x4*)(p + 2048 * u)); b[u] = __builtin_nontemporal_load((const u32x4*)(p + 2048 * u + 16)); }
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < KB / 2; ++u) acc ^= a[u] ^ b[u];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[w * 64 + lane] = acc;
}

int main() {
    const size_t n = (size_t)1 << 24;
    unsigned char* d_in;
    u32x4* d_out;
    CHK(hipMalloc(&d_in, n * 32));
    CHK(hipMalloc(&d_out, 64 << 20));
    CHK(hipMemset(d_in, 1, n * 32));
    CHK(hipFuncSetAttribute((const void*)k_rows<8, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void*)k_rows<16, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHK(hipFuncSetAttribute((const void*)k_rows<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int reps = 20;
    auto run = [&](const char* name, auto launch) -> int {
        launch(); launch();
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.1f us  %7.1f GB/s\n", name, ms * 1e3 / reps, n * 32.0 * reps / (ms * 1e-3) / 1e9);
        return 0;
    };
    for (uint32_t log_s : {0u, 3u, 6u, 8u, 10u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "pieces of 16 KiB per wave, stride permutation 2^%u", log_s);
        run(nm, [&] { k_pieces<16><<<(unsigned)(n * 32 / 16384 / 4), 256>>>(d_in, (uint32_t)(n * 32 / 16384), log_s, d_out); });
        snprintf(nm, sizeof nm, "pieces of 8 KiB per wave, stride permutation 2^%u", log_s);
        run(nm, [&] { k_pieces<8><<<(unsigned)(n * 32 / 8192 / 4), 256>>>(d_in, (uint32_t)(n * 32 / 8192), log_s, d_out); });
    }
    return 0;
}
