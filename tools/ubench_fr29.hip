// ubench_fr29.hip -- go / no-go for an UNSATURATED Fr in the issue-bound kernels (NTT butterflies, 3-5-table product rounds).
//
// Candidate: 9 limbs of 29 bits (R' = 2^261): a column of limb products fits one 64-bit accumulator, so a Montgomery product is
// 81 + 81 v_mad_u64_u32 and a handful of shifts -- no carry chains -- against the saturated 8 x 32-bit product of csrc/fp.hpp (product
// scanning with a 96-bit column accumulator, ~330 instructions).  Values stay in arkworks' Montgomery form x 2^256: with the twiddle held
// as w 2^261 the product mont'(w 2^261, x 2^256) = w x 2^256 needs no conversion; results are lazy (< 2 r), additions are limb-wise.
//
// What is measured: products per ns on the whole chip, four independent chains per lane (as a lane's two butterflies give), for
//   (a) the shipped saturated product (zk::Fr::operator*),
//   (b) the 9 x 29 product + the carry normalisation its input needs after a lazy add / sub,
//   (c) a whole butterfly of each kind (product + add + sub; the unsaturated one with its normalise and an approximate range fold).
// Correctness of (b) is checked on the HOST first, against csrc/host_fr.hpp (same code compiled for the host), on random operands.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-cryptography_amd/csrc -o tools/ubench_fr29 tools/ubench_fr29.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "fp.hpp"
#include "host_fr.hpp"

namespace fr29 {
constexpr uint32_t MASK = (1u << 29) - 1;
struct E { uint32_t l[9]; };
// r in 29-bit limbs, and -r^-1 mod 2^29
__host__ __device__ constexpr uint32_t P29(int i) {
    constexpr uint32_t p[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
    return p[i];
}
constexpr uint32_t NINV29 = 0x1fffffffu;     // r = 1 mod 2^29 ... -1/r = -1 mod 2^29
// value form -> 29-bit limbs (exact, any 256-bit integer) and back (limbs must be normalised, value < 2^256)
__host__ __device__ inline E unpack(const uint32_t w[8]) {
    E e;
    uint64_t acc = 0;
    int bits = 0, k = 0;
    for (int i = 0; i < 8; ++i) {
        acc |= (uint64_t)w[i] << bits;
        bits += 32;
        while (bits >= 29 && k < 9) { e.l[k++] = (uint32_t)acc & MASK; acc >>= 29; bits -= 29; }
    }
    if (k < 9) e.l[k++] = (uint32_t)acc & MASK;
    return e;
}
__host__ __device__ inline void pack(const E& e, uint32_t w[8]) {
    uint64_t acc = 0;
    int bits = 0, k = 0;
    for (int i = 0; i < 9; ++i) {
        acc |= (uint64_t)e.l[i] << bits;
        bits += 29;
        while (bits >= 32 && k < 8) { w[k++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
    }
    if (k < 8) w[k++] = (uint32_t)acc;
}
// carries into place: limbs < 2^29 afterwards (the top limb takes what is left)
__host__ __device__ inline void normalise(E& e) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { e.l[i + 1] += e.l[i] >> 29; e.l[i] &= MASK; }
}
// Montgomery product x w 2^-261: x lazy (limbs < 2^30), w normalised and < r.  Result normalised, < 2 r (for x < 2^261).
__host__ __device__ inline E mul(const E& x, const E& w) {
    uint64_t t[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) t[i + j] += (uint64_t)x.l[i] * w.l[j];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t m = ((uint32_t)t[i] * NINV29) & MASK;
#pragma unroll
        for (int j = 0; j < 9; ++j) t[i + j] += (uint64_t)m * P29(j);
        t[i + 1] += t[i] >> 29;
    }
    E r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        r.l[k] = (uint32_t)t[9 + k] & MASK;
        if (k < 8) t[10 + k] += t[9 + k] >> 29;
    }
    r.l[8] = (uint32_t)t[17];      // (what is left: the value is < 2 r < 2^256)
    return r;
}
__host__ __device__ inline E add(const E& a, const E& b) {
    E r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + 4 r with every limb of the constant large enough that no limb underflows (b normalised, < 2 r)
__host__ __device__ inline E sub4(const E& a, const E& b) {
    // 4 r = sum c_i 2^(29 i) with c_i >= 2^30 - 2 > b_i for i < 8: every limb lends 2^30 = 2 x 2^29 to the one below (a_i + c_i < 2^32)
    E r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t c = 4u * P29(i) + (i < 8 ? (1u << 30) : 0u) - (i > 0 ? 2u : 0u);
        r.l[i] = a.l[i] + c - b.l[i];
    }
    return r;
}
}  // namespace fr29

// ---- host self-test -----------------------------------------------------------------------------------------------------
static bool selftest() {
    using namespace zkhost;
    std::mt19937_64 rng(12345);
    Fr r2, one = fr_one();
    std::memcpy(r2.l, FR_R2, 32);
    // zkhost::fr_mul(a, b) = a b 2^-256 mod r on the INTEGERS its arguments' limbs spell.  c517 = 2^517 mod r: fr_mul(w, c517) = w 2^261 mod r.
    const Fr c517 = fr_mul(r2, fr_from_u64(32));         // 2^512 * (32 * 2^256) * 2^-256
    for (int it = 0; it < 200000; ++it) {
        Fr x, w;
        for (int i = 0; i < 4; ++i) { x.l[i] = rng(); w.l[i] = rng(); }
        x.l[3] &= 0x3fffffffffffffffULL; w.l[3] &= 0x3fffffffffffffffULL;
        x = fr_mul(x, one); w = fr_mul(w, one);          // below r
        const Fr w261 = fr_mul(w, c517);
        uint32_t xw[8], ww[8];
        std::memcpy(xw, x.l, 32); std::memcpy(ww, w261.l, 32);
        const fr29::E xe = fr29::unpack(xw), we = fr29::unpack(ww);
        auto canon = [&](const fr29::E& e) {              // the lazy result (< 2^256) mod r
            uint32_t pw[8];
            fr29::pack(e, pw);
            Fr g;
            std::memcpy(g.l, pw, 32);
            return fr_mul(g, one);
        };
        const fr29::E pe = fr29::mul(xe, we);             // = x w 2^261 2^-261 = x w (mod r), < 2 r
        const Fr xw_int = fr_mul(fr_mul(x, w), r2);       // the integer x w mod r
        if (std::memcmp(canon(pe).l, xw_int.l, 32) != 0) { std::printf("SELFTEST FAILED at %d\n", it); return false; }
        // lazy add / sub feeding a product: (x + x w) w and (x - x w + 4 r) w
        fr29::E s = fr29::add(xe, pe), d = fr29::sub4(xe, pe);
        fr29::normalise(s); fr29::normalise(d);
        const Fr ws = fr_mul(fr_mul(fr_add(x, xw_int), w), r2), wd = fr_mul(fr_mul(fr_sub(x, xw_int), w), r2);
        if (std::memcmp(canon(fr29::mul(s, we)).l, ws.l, 32) != 0 || std::memcmp(canon(fr29::mul(d, we)).l, wd.l, 32) != 0) {
            std::printf("SELFTEST (lazy add / sub) FAILED at %d\n", it);
            return false;
        }
    }
    std::printf("SELFTEST OK: 9 x 29-bit Montgomery product, lazy add / sub, 200000 random cases against csrc/host_fr.hpp\n");
    return true;
}

// ---- device throughput ----------------------------------------------------------------------------------------------------
constexpr int CHAINS = 4;
__global__ __launch_bounds__(256) void k_sat_mul(const uint64_t* in, uint64_t* out, int iters) {
    zk::Fr x[CHAINS], w = zk::load_fr(in, 0);
    for (int c = 0; c < CHAINS; ++c) x[c] = zk::load_fr(in, 1 + c + (threadIdx.x & 3));
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = x[c] * w;
    zk::Fr s = x[0];
    for (int c = 1; c < CHAINS; ++c) s = s + x[c];
    if (s.l[0] == 0x12345678u) zk::store_fr(out, blockIdx.x * 256 + threadIdx.x, s);
}
__global__ __launch_bounds__(256) void k_sat_butterfly(const uint64_t* in, uint64_t* out, int iters) {
    zk::Fr u[CHAINS], v[CHAINS], w = zk::load_fr(in, 0);
    for (int c = 0; c < CHAINS; ++c) { u[c] = zk::load_fr(in, 1 + c); v[c] = zk::load_fr(in, 5 + c + (threadIdx.x & 3)); }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const zk::Fr t = v[c] * w;
            v[c] = u[c] - t;
            u[c] = u[c] + t;
        }
    zk::Fr s = u[0] + v[0];
    for (int c = 1; c < CHAINS; ++c) s = s + u[c] + v[c];
    if (s.l[0] == 0x12345678u) zk::store_fr(out, blockIdx.x * 256 + threadIdx.x, s);
}
__device__ inline fr29::E load29(const uint64_t* p, size_t i) {
    uint32_t w[8];
    const uint4* q = reinterpret_cast<const uint4*>(p + 4 * i);
    const uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    return fr29::unpack(w);
}
__global__ __launch_bounds__(256) void k_29_mul(const uint64_t* in, uint64_t* out, int iters) {
    fr29::E x[CHAINS], w = load29(in, 0);
    for (int c = 0; c < CHAINS; ++c) x[c] = load29(in, 1 + c + (threadIdx.x & 3));
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = fr29::mul(x[c], w);
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int k = 0; k < 9; ++k) s ^= x[c].l[k];
    if (s == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_29_butterfly(const uint64_t* in, uint64_t* out, int iters) {
    fr29::E u[CHAINS], v[CHAINS], w = load29(in, 0);
    for (int c = 0; c < CHAINS; ++c) { u[c] = load29(in, 1 + c); v[c] = load29(in, 5 + c + (threadIdx.x & 3)); }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            fr29::normalise(v[c]);                         // the product's input: carries of the lazy add / sub into place
            const fr29::E t = fr29::mul(v[c], w);
            v[c] = fr29::sub4(u[c], t);
            u[c] = fr29::add(u[c], t);
            // approximate range fold (keeps the running sums below 2^261 over a pass): if the top limb says >= 8 r, take 8 r off
            const bool big = u[c].l[8] >= 8u * fr29::P29(8) + 8u;
#pragma unroll
            for (int k = 0; k < 9; ++k) u[c].l[k] -= big ? 8u * fr29::P29(k) : 0u;   // (benchmark of the instruction count; limb borrows ignored here)
        }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int k = 0; k < 9; ++k) s ^= u[c].l[k] ^ v[c].l[k];
    if (s == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K> static double run(const char* name, K k, const uint64_t* d_in, uint64_t* d_out, int waves_per_simd, double ops_per_iter) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 256, grid = 256 * waves_per_simd;          // 256 CUs x (waves_per_simd workgroups of 4 waves)
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_in, d_out, 8);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_in, d_out, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double ops = (double)grid * 256 * iters * CHAINS * ops_per_iter;
    std::printf("%-44s %d waves/SIMD: %8.2f per ns\n", name, waves_per_simd, ops / (best * 1e6));
    return ops / (best * 1e6);
}

int main() {
    if (!selftest()) return 1;
    uint64_t h[4 * 16];
    std::mt19937_64 rng(7);
    for (int i = 0; i < 16; ++i) {
        zkhost::Fr x;
        for (int k = 0; k < 4; ++k) x.l[k] = rng();
        x.l[3] &= 0x3fffffffffffffffULL;
        x = zkhost::fr_mul(x, zkhost::fr_one());
        std::memcpy(h + 4 * i, x.l, 32);
    }
    uint64_t *d_in, *d_out;
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, 8 * 256 * 256 * 8);
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        const double a = run("products: saturated 8 x 32 (csrc/fp.hpp)", k_sat_mul, d_in, d_out, w, 1.0);
        const double b = run("products: unsaturated 9 x 29", k_29_mul, d_in, d_out, w, 1.0);
        const double c = run("butterflies: saturated", k_sat_butterfly, d_in, d_out, w, 1.0);
        const double d = run("butterflies: unsaturated (+ normalise, fold)", k_29_butterfly, d_in, d_out, w, 1.0);
        std::printf("   -> products x %.2f, butterflies x %.2f\n", b / a, d / c);
    }
    return 0;
}
