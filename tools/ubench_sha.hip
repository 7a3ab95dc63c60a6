// ubench_sha.hip -- what one SHA-256 block costs a single wave on gfx950 (the serial floor of every prover round):
// state rounds only (K+W read from LDS, as the provers' hash wave does) and schedule + state rounds on the same wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_sha tools/ubench_sha.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zk-cryptography_amd/csrc/transcript.hpp"
using namespace zk;

__global__ void rounds_only(uint32_t* out, int n_blocks) {
    __shared__ uint32_t kw[64];
    if (threadIdx.x < 64) kw[threadIdx.x] = SHA256_K[threadIdx.x] + threadIdx.x * 2654435761u;
    __syncthreads();
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (int b = 0; b < n_blocks; ++b) sha256_rounds_block(h, kw);
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) out[i] = h[i];
}
__global__ void schedule_and_rounds(uint32_t* out, int n_blocks) {
    __shared__ uint32_t kw[64];
    __shared__ uint32_t blk[16];
    if (threadIdx.x < 16) blk[threadIdx.x] = threadIdx.x * 2246822519u;
    __syncthreads();
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (int b = 0; b < n_blocks; ++b) {
        sha256_schedule_block(blk, kw);
        __syncthreads();
        sha256_rounds_block(h, kw);
        if (threadIdx.x == 0) blk[b & 15] ^= h[b & 7];
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) out[i] = h[i];
}
template <class F> static float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    uint32_t* d; hipMalloc(&d, 64);
    const int n = 20000;
    float t1 = time_ms([&] { hipLaunchKernelGGL(rounds_only, dim3(1), dim3(64), 0, 0, d, n); });
    float t2 = time_ms([&] { hipLaunchKernelGGL(schedule_and_rounds, dim3(1), dim3(64), 0, 0, d, n); });
    std::printf("state rounds only (K+W from LDS): %.3f us per block\n", 1e3 * t1 / n);
    std::printf("schedule + state rounds, one wave:  %.3f us per block\n", 1e3 * t2 / n);
    return 0;
}
