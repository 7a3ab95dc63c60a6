// ubench_sha_split.hip -- SHA-256 state rounds on a lone wave: the 14-instruction round (sha256_rounds_block) against the
// six-lane form (sha256_rounds_block_split, csrc/transcript.hpp), same chained blocks; digests compared, time per block.
// Build: hipcc -O3 --offload-arch=gfx950 -I zk-cryptography_amd/csrc -o tools/ubench_sha_split tools/ubench_sha_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "transcript.hpp"
using namespace zk;

template <bool SPLIT> __global__ void chain_kernel(const uint32_t* blocks, int n_blocks, int reps, uint32_t* out) {
    __shared__ uint32_t kw[64 * 8];
    // schedules of up to 8 blocks: SPLIT: a block per row of 16 lanes (sha256_schedule_rows_to_lds); else one after another
    if (SPLIT) {
        for (int b0 = 0; b0 < n_blocks; b0 += 4) {
            const int b = b0 + (threadIdx.x >> 4);
            const bool active = b < n_blocks;
            const int bb = active ? b : b0;
            sha256_schedule_rows_to_lds(blocks[16 * bb + (threadIdx.x & 15)], kw + 64 * bb, nullptr, 0u, 0u, active);
        }
    } else {
        for (int b = 0; b < n_blocks; ++b) sha256_schedule_block(blocks + 16 * b, kw + 64 * b);
    }
    __syncthreads();
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    if (SPLIT) {
        ShaSplit sp;
        sp.init();
        uint32_t hs[4];
        sp.split(h, hs);
        for (int r = 0; r < reps; ++r)
            for (int b = 0; b < n_blocks; ++b) sha256_rounds_block_split(sp, hs, kw + 64 * b);
        sp.join(hs, h);
    } else {
        for (int r = 0; r < reps; ++r)
            for (int b = 0; b < n_blocks; ++b) sha256_rounds_block(h, kw + 64 * b);
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) out[i] = h[i];
}

// the schedules alone: `reps` times the schedules of four blocks, serial (one block after another) or a block per row
template <bool ROWS> __global__ void schedule_kernel(const uint32_t* blocks, int reps, uint32_t* out) {
    __shared__ uint32_t kw[64 * 4];
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        if (ROWS) sha256_schedule_rows_to_lds(blocks[threadIdx.x] + acc, kw + 64 * (threadIdx.x >> 4), nullptr, 0u, 0u, true);
        else sha256_schedule_block(blocks, kw);
        acc += kw[63];
    }
    if (threadIdx.x == 0) out[8] = acc;
}

// the closing kernels' hash wave as it is called there: sha256_message_split on four blocks whose schedules are complete
__global__ void message_kernel(const uint32_t* blocks, int reps, uint32_t* out) {
    __shared__ uint32_t kw[64 * 4];
    __shared__ uint32_t msg[16 * 4];
    __shared__ uint32_t ready[4];
    sha256_schedule_rows_to_lds(blocks[threadIdx.x], kw + 64 * (threadIdx.x >> 4), nullptr, 0u, 0u, true);
    msg[threadIdx.x] = blocks[threadIdx.x];
    if (threadIdx.x < 4) ready[threadIdx.x] = 4;
    __syncthreads();
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (int r = 0; r < reps; ++r) sha256_message_split(h, msg, kw, ready, 4);
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) out[i] = h[i];
}

// ... and beside eleven other waves of its workgroup (as in the closing kernels): idle, multiplying field elements, or streaming LDS
template <int OTHERS> __global__ __launch_bounds__(768) void message_beside_kernel(const uint32_t* blocks, int reps, uint32_t* out) {
    __shared__ uint32_t kw[64 * 4];
    __shared__ uint32_t msg[16 * 4];
    __shared__ uint32_t ready[4];
    __shared__ uint32_t traffic[8192];
    __shared__ volatile uint32_t done;
    if (threadIdx.x < 64) {
        sha256_schedule_rows_to_lds(blocks[threadIdx.x], kw + 64 * (threadIdx.x >> 4), nullptr, 0u, 0u, true);
        msg[threadIdx.x] = blocks[threadIdx.x];
        if (threadIdx.x < 4) ready[threadIdx.x] = 4;
        if (threadIdx.x == 0) done = 0;
    }
    for (uint32_t i = threadIdx.x; i < 8192; i += 768) traffic[i] = i * 2654435761u;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        for (int r = 0; r < reps; ++r) sha256_message_split(h, msg, kw, ready, 4);
        if (threadIdx.x == 0) {
            for (int i = 0; i < 8; ++i) out[i] = h[i];
            done = 1;
        }
    } else {
        Fr a = Fr::one(), b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b.l[i] = traffic[(threadIdx.x + i) & 8191] | 1u;
        uint32_t x = threadIdx.x;
        while (!done) {
            if (OTHERS == 0) __builtin_amdgcn_s_sleep(8);
            if (OTHERS == 1 || OTHERS == 3) a = fr_mul_outlined(a, b);
            if (OTHERS == 2 || OTHERS == 3) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { x = traffic[x & 8191] + k; traffic[(x >> 7) & 8191] = x; }
            }
        }
        if (a.l[0] == 0x12345678u && x == 77u) out[20] = a.l[1];
    }
}

static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void host_compress(uint32_t h[8], const uint32_t* blk) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe,
        0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7,
        0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b,
        0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = blk[i];
    for (int i = 16; i < 64; ++i) w[i] = w[i - 16] + (rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

int main() {
    const int nb = 4, reps = 2000;
    uint32_t hb[16 * nb];
    uint32_t x = 12345;
    for (int i = 0; i < 16 * nb; ++i) { x = x * 1664525u + 1013904223u; hb[i] = x; }
    uint32_t want[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (int r = 0; r < reps; ++r)
        for (int b = 0; b < nb; ++b) host_compress(want, hb + 16 * b);
    uint32_t *d_blk, *d_out;
    hipMalloc(&d_blk, sizeof hb); hipMalloc(&d_out, 64);
    hipMemcpy(d_blk, hb, sizeof hb, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int bad = 0;
    for (int split = 0; split < 2; ++split) {
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(a);
            if (split) hipLaunchKernelGGL(chain_kernel<true>, dim3(1), dim3(64), 0, 0, d_blk, nb, reps, d_out);
            else hipLaunchKernelGGL(chain_kernel<false>, dim3(1), dim3(64), 0, 0, d_blk, nb, reps, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        uint32_t got[8]; hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost);
        bool same = true;
        for (int i = 0; i < 8; ++i) same = same && got[i] == want[i];
        bad += !same;
        std::printf("%-34s %7.3f us per block   digest %s\n", split ? "six-lane rounds (9 instr / round)" : "one-lane rounds (14 instr / round)", 1e3 * ms / (reps * nb), same ? "matches the host's" : "DIFFERS");
    }
    for (int rows = 0; rows < 2; ++rows) {
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(a);
            if (rows) hipLaunchKernelGGL(schedule_kernel<true>, dim3(1), dim3(64), 0, 0, d_blk, reps, d_out);
            else hipLaunchKernelGGL(schedule_kernel<false>, dim3(1), dim3(64), 0, 0, d_blk, reps, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        std::printf("%-34s %7.3f us per call\n", rows ? "schedules of 4 blocks, one per row" : "schedule of 1 block, serial", 1e3 * ms / reps);
    }
    {
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(a);
            hipLaunchKernelGGL(message_kernel, dim3(1), dim3(64), 0, 0, d_blk, reps, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        uint32_t got[8]; hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost);
        bool same = true;
        for (int i = 0; i < 8; ++i) same = same && got[i] == want[i];
        bad += !same;
        std::printf("%-34s %7.3f us per message of 4 blocks   digest %s\n", "sha256_message_split", 1e3 * ms / reps, same ? "matches the host's" : "DIFFERS");
    }
    for (int others = 0; others < 4; ++others) {
        const int r2 = 400;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(a);
            if (others == 0) hipLaunchKernelGGL(message_beside_kernel<0>, dim3(1), dim3(768), 0, 0, d_blk, r2, d_out);
            if (others == 1) hipLaunchKernelGGL(message_beside_kernel<1>, dim3(1), dim3(768), 0, 0, d_blk, r2, d_out);
            if (others == 2) hipLaunchKernelGGL(message_beside_kernel<2>, dim3(1), dim3(768), 0, 0, d_blk, r2, d_out);
            if (others == 3) hipLaunchKernelGGL(message_beside_kernel<3>, dim3(1), dim3(768), 0, 0, d_blk, r2, d_out);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const char* what[4] = {"asleep", "multiplying field elements", "streaming LDS", "multiplying and streaming LDS"};
        std::printf("sha256_message_split beside 11 waves %-30s %7.3f us per message of 4 blocks\n", what[others], 1e3 * ms / r2);
    }
    return bad;
}
