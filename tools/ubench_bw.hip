// ubench_bw.hip -- HBM read ceilings on gfx950 for the access shapes the fold kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_bw.hip -o tools/ubench_bw ; run on the GPU box.
//   stream : every lane reads 32 B (two dwordx4), consecutive lanes consecutive elements, grid-stride (what
//            chunk_sums / fold do)
//   rows   : a wave reads one 2 KiB piece of each of R rows that lie `m` elements apart (what multifold<64> does),
//            U loads in flight per lane
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ in, size_t n_elems, uint4* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n_elems; j += U * stride) {
        uint4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t ju = j + u * stride;
            if (ju < n_elems) { a[u] = in[2 * ju]; b[u] = in[2 * ju + 1]; } else { a[u] = b[u] = make_uint4(0, 0, 0, 0); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= a[u].x ^ b[u].x; acc.y ^= a[u].y ^ b[u].y; acc.z ^= a[u].z ^ b[u].z; acc.w ^= a[u].w ^ b[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// contiguous chunk per workgroup (what chunk_sums does)
template <int U>
__global__ __launch_bounds__(256) void k_chunk(const uint4* __restrict__ in, uint32_t chunk, uint4* __restrict__ out) {
    const uint4* base = in + 2 * (size_t)blockIdx.x * chunk;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t j = threadIdx.x; j < chunk; j += U * 256) {
        uint4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = base[2 * (size_t)(j + u * 256)]; b[u] = base[2 * (size_t)(j + u * 256) + 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= a[u].x ^ b[u].x; acc.y ^= a[u].y ^ b[u].y; acc.z ^= a[u].z ^ b[u].z; acc.w ^= a[u].w ^ b[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// rows: workgroup b owns outputs [64 b, 64 b + 64); its W waves split the R rows
template <int U>
__global__ __launch_bounds__(1024) void k_rows(const uint4* __restrict__ in, size_t m, uint32_t rows, uint4* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t per = rows / n_waves;
    const size_t j = (size_t)blockIdx.x * 64 + lane;
    const uint4* p = in + 2 * ((size_t)wave * per * m + j);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t t = 0; t < per; t += U) {
        uint4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint4* q = p + 2 * (size_t)(t + u) * m; a[u] = q[0]; b[u] = q[1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= a[u].x ^ b[u].x; acc.y ^= a[u].y ^ b[u].y; acc.z ^= a[u].z ^ b[u].z; acc.w ^= a[u].w ^ b[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 64 + lane] = acc;
}

int main() {
    const size_t n = (size_t)1 << 24;   // 32-byte elements: 512 MiB
    uint4 *d_in, *d_out;
    CHK(hipMalloc(&d_in, n * 32));
    CHK(hipMalloc(&d_out, 64 << 20));
    CHK(hipMemset(d_in, 1, n * 32));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto report = [&](const char* name, int grid, float ms, int reps) {
        printf("%-34s grid %6d  %8.1f us  %7.1f GB/s\n", name, grid, ms * 1e3 / reps, n * 32.0 * reps / (ms * 1e-3) / 1e9);
    };
    const int reps = 20;
    auto run = [&](const char* name, int grid, auto launch) -> int {
        launch(); launch();
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        report(name, grid, ms, reps);
        return 0;
    };
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        run("stream U=2", grid, [&] { k_stream<2><<<grid, 256>>>(d_in, n, d_out); });
        run("stream U=4", grid, [&] { k_stream<4><<<grid, 256>>>(d_in, n, d_out); });
        run("stream U=8", grid, [&] { k_stream<8><<<grid, 256>>>(d_in, n, d_out); });
    }
    for (uint32_t chunk : {1024u, 4096u, 16384u}) {
        const int grid = (int)(n / chunk);
        run("chunk U=4", grid, [&] { k_chunk<4><<<grid, 256>>>(d_in, chunk, d_out); });
    }
    {
        const uint32_t rows = 256;
        const size_t m = n / rows;
        const int grid = (int)(m / 64);
        for (int waves : {4, 8, 16}) {
            char nm[64];
            snprintf(nm, sizeof nm, "rows(256) %d waves U=4", waves);
            run(nm, grid, [&] { k_rows<4><<<grid, 64 * waves>>>(d_in, m, rows, d_out); });
            snprintf(nm, sizeof nm, "rows(256) %d waves U=8", waves);
            run(nm, grid, [&] { k_rows<8><<<grid, 64 * waves>>>(d_in, m, rows, d_out); });
            snprintf(nm, sizeof nm, "rows(256) %d waves U=16", waves);
            run(nm, grid, [&] { k_rows<16><<<grid, 64 * waves>>>(d_in, m, rows, d_out); });
        }
    }
    return 0;
}
