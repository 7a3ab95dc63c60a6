// probe_tr_b8.hip -- what ds_read_b64_tr_b8 returns (gfx950): LDS holds its own byte addresses (low byte, then high byte in a second run),
// every lane passes an address, and the 8 returned bytes per lane are printed as source addresses.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_tr_b8 tools/probe_tr_b8.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(int pattern, int hi, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) unsigned char q[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) q[i] = (unsigned char)(hi ? (i >> 8) : (i & 255));
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t off;
    if (pattern == 0) off = l * 8;
    else if (pattern == 1) off = l * 16;
    else if (pattern == 2) off = (l % 16) * 64 + (l / 16) * 8;       // 16 rows of 64 bytes, 4 column groups of 8 bytes
    else off = (l % 8) * 128 + (l / 8) * 8;                            // 8 rows of 128 bytes
    const uint32_t a = (uint32_t)(uintptr_t)(q + off);                // the low 32 bits of a generic LDS pointer are the LDS offset
    u32x2 r;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[2 * l] = r.x;
    out[2 * l + 1] = r.y;
}
int main() {
    uint32_t *d, lo[128], hi[128];
    hipMalloc(&d, 512);
    for (int pattern = 0; pattern < 4; ++pattern) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, pattern, 0, d);
        hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, pattern, 1, d);
        hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (lane: source LDS offsets of its 8 bytes, relative to the array)\n", pattern);
        // the array's own LDS offset is unknown but constant: print addresses relative to lane 0's first byte
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int b = 0; b < 8; ++b) {
                const uint32_t lb = (lo[2 * l + b / 4] >> (8 * (b % 4))) & 255, hb = (hi[2 * l + b / 4] >> (8 * (b % 4))) & 255;
                printf(" %5u", hb * 256 + lb);
            }
            printf("\n");
        }
    }
    return 0;
}
