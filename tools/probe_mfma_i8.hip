// probe_mfma_i8.hip -- hardware facts the MFMA form of the k-variable fold (csrc/mfma_fold.hpp) relies on, checked on gfx950:
//   (1) ds_read_b128 from a byte-aligned LDS address returns the 16 bytes at that address;
//   (2) v_mfma_i32_32x32x32_i8: lane (n = lane % 32, h = lane / 32) supplies 16 consecutive K of row / column n, the SAME K
//       numbering for A and B; D[i][j] sits in lane j + 32 ((i / 4) % 2), register (i % 4) + 4 (i / 8);
//   (3) v_permlane32_swap_b32 vdst, src: vdst' = [vdst.lo, src.lo], src' = [vdst.hi, src.hi].
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_mfma_i8 tools/probe_mfma_i8.hip ; prints "ok" lines or mismatches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_unaligned(const uint8_t* g, uint8_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t q[512];
    for (int i = threadIdx.x; i < 512; i += 64) q[i] = g[i];
    __syncthreads();
    v4i a;
    __builtin_memcpy(&a, q + threadIdx.x + 3 * (threadIdx.x >> 4), 16);   // offsets 0..63 + a little: every alignment
    __builtin_memcpy(out + 16 * threadIdx.x, &a, 16);
}

// A: 32 x 32 (row m, k), B: 32 x 32 (k, col n) as int8; lane (r, h) passes bytes [16 h, 16 h + 16) of row r of A / column r of B
__global__ void k_mfma(const int8_t* A, const int8_t* B, int* D) {
    const int r = threadIdx.x & 31, h = threadIdx.x >> 5;
    v4i a, b;
    int8_t ta[16], tb[16];
    for (int t = 0; t < 16; ++t) { ta[t] = A[r * 32 + 16 * h + t]; tb[t] = B[(16 * h + t) * 32 + r]; }
    __builtin_memcpy(&a, ta, 16);
    __builtin_memcpy(&b, tb, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[threadIdx.x * 16 + i] = c[i];
}

__global__ void k_swap(unsigned* out) {
    unsigned x = threadIdx.x, y = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}

int main() {
    int bad = 0;
    {
        uint8_t h[512], *d, *o, ho[1024];
        for (int i = 0; i < 512; ++i) h[i] = (uint8_t)(i * 7 + 1);
        hipMalloc(&d, 512); hipMalloc(&o, 1024);
        hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_unaligned, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 16; ++t)
                if (ho[16 * l + t] != h[l + 3 * (l >> 4) + t]) { if (bad < 5) printf("unaligned ds_read_b128: lane %d byte %d\n", l, t); ++bad; }
        printf("unaligned ds_read_b128: %s\n", bad ? "MISMATCH" : "ok");
    }
    int bad2 = 0;
    {
        int8_t A[1024], B[1024];
        int D[1024], want[1024];
        srand(5);
        for (int i = 0; i < 1024; ++i) { A[i] = (int8_t)(rand() % 256 - 128); B[i] = (int8_t)(rand() % 256 - 128); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += (int)A[i * 32 + kk] * (int)B[kk * 32 + j]; want[i * 32 + j] = s; }
        int8_t *dA, *dB; int* dD;
        hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
        hipMemcpy(dA, A, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            const int lane = j + 32 * ((i / 4) % 2), reg = (i % 4) + 4 * (i / 8);
            if (D[lane * 16 + reg] != want[i * 32 + j]) { if (bad2 < 5) printf("mfma layout: D[%d][%d] got %d want %d\n", i, j, D[lane * 16 + reg], want[i * 32 + j]); ++bad2; }
        }
        printf("v_mfma_i32_32x32x32_i8 operand / result layout: %s\n", bad2 ? "MISMATCH" : "ok");
    }
    int bad3 = 0;
    {
        unsigned* d; unsigned h[128];
        hipMalloc(&d, 512);
        hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, d);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            const unsigned w0 = l < 32 ? (unsigned)l : 1000u + (l - 32), w1 = l < 32 ? (unsigned)(l + 32) : 1000u + l;
            if (h[l] != w0 || h[64 + l] != w1) { if (bad3 < 5) printf("permlane32_swap lane %d: %u %u want %u %u\n", l, h[l], h[64 + l], w0, w1); ++bad3; }
        }
        printf("v_permlane32_swap_b32: %s\n", bad3 ? "MISMATCH" : "ok");
    }
    return (bad || bad2 || bad3) ? 1 : 0;
}
