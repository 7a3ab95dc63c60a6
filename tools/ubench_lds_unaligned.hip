// ubench_lds_unaligned.hip -- cycles per ds_read_b128 wave-instruction by address alignment (gfx950, unaligned access mode):
// the MFMA fold reads its Toeplitz rows at BYTE offsets (lane n wants 16 bytes starting at 31 - n + 16 h of a 96-byte string).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lds_unaligned tools/ubench_lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
// mode 0: lane * 16 + mis (conflict-free when mis = 0); mode 1: the fold's pattern (31 - n + 16 h) + mis, + 96 per iteration
__global__ void k(int mode, int mis, int iters, unsigned long long* out, int* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char q[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) q[i] = (unsigned char)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const unsigned char* p = q + (mode == 0 ? lane * 16 + mis : 31 - n + 16 * h + mis);
    v4i acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v4i a;
            __builtin_memcpy(&a, p + (mode == 0 ? 1024 * u : 96 * u + 1536 * (it & 7)), 16);
            acc ^= a;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567) sink[threadIdx.x] = acc.x;
}
int main() {
    unsigned long long* d; int* s;
    hipMalloc(&d, 8 * 1024); hipMalloc(&s, 4096);
    const int iters = 1000;
    for (int waves : {1, 4, 8})
        for (int mode : {0, 1})
            for (int mis : {0, 1, 2, 4, 8}) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, mode, mis, iters, d, s);
                hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, mode, mis, iters, d, s);
                unsigned long long t;
                hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
                printf("waves %d  mode %d (%s)  misalign %d: %.1f cycles per ds_read_b128 per wave (%.1f per CU-instruction)\n", waves, mode,
                       mode ? "fold pattern" : "lane*16", mis, (double)t / (16.0 * iters), (double)t / (16.0 * iters) / waves);
            }
    return 0;
}
