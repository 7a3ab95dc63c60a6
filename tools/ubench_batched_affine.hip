// ubench_batched_affine.hip -- would BATCHED AFFINE additions (6 Fq products per addition + a shared inversion, Montgomery's trick)
// beat the XYZZ mixed addition (10 products) the MSM's accumulate pass is built on, on gfx950?
//
// The comparison is tilted TOWARDS batched affine on purpose -- if it does not win here it does not win in the pass:
//   * the pairs (P_i, Q_i) of a batch are given (a real pass has to form them from the bucket lists: a tree over every list, five
//     rounds at list length 26, each a pass over global memory);
//   * no special cases (P = +-Q zeroes the shared product: a real pass must find and set aside such pairs before it multiplies);
//   * results are left unnormalised (x3 < 10p, y3 < 6p: a next round would have to bring them back under the table's bound);
//   * every lane inverts for itself, over k pairs: the Fermat inversion (~570 products; a 64-lane-wide batch would leave 63 lanes
//     idle for as long) is paid once per k additions, the k prefix products go through global memory (14 limbs each, coalesced).
// Both kernels read their points as the accumulate pass does (128-byte records of the internal 28-bit-limb layout, gathered by index).
// Every batched result is checked against the XYZZ addition of the same pair (x3 * ZZ == X, y3 * ZZZ == Y mod p).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-cryptography_amd/csrc -o tools/ubench_batched_affine tools/ubench_batched_affine.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "g1u.hpp"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace zk;

// a^(p-2): square and multiply over the 381 bits of p - 2
__device__ __noinline__ FqU fqu_inverse(FqU a) {
    FqU acc = FqU::one();
    for (int w = 11; w >= 0; --w) {
        uint32_t e = FqParams::p(0) - 2;
#pragma unroll
        for (int q = 0; q < 12; ++q) if (q == w) e = (q == 0) ? FqParams::p(0) - 2 : FqParams::p(q);
        for (int b = 31; b >= 0; --b) {
            acc = fqu_sqr(acc);
            if ((e >> b) & 1) acc = fqu_mul(acc, a);
        }
    }
    return acc;
}

// XYZZ: every lane sums a list of `len` points (the accumulate pass's inner loop): len mixed additions of 10 products
__global__ __launch_bounds__(256) void k_xyzz(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ idx, uint32_t len, uint32_t n_lanes,
                                              uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_lanes) return;
    G1XyzzU acc = G1XyzzU::identity();
    G1AffineU nxt = load_affine_u(pts, idx[t]);
    for (uint32_t k = 0; k < len; ++k) {
        const G1AffineU cur = nxt;
        if (k + 1 < len) nxt = load_affine_u(pts, idx[(size_t)(k + 1) * n_lanes + t]);
        g1u_madd(acc, cur, false);
    }
    store_xyzz_u(out, t, acc);
}

// batched affine: every lane adds k pairs (P_i, Q_i) = (pts[idx[2 i]], pts[idx[2 i + 1]]) with ONE inversion
__global__ __launch_bounds__(256) void k_batched(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ idx, uint32_t k_pairs, uint32_t n_lanes,
                                                 uint32_t* __restrict__ prefix /* k_pairs x 14 x n_lanes */, uint32_t* __restrict__ out /* 128-byte records */) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_lanes) return;
    FqU acc = FqU::one();
    for (uint32_t i = 0; i < k_pairs; ++i) {
        const uint32_t ip = idx[(size_t)(2 * i) * n_lanes + t], iq = idx[(size_t)(2 * i + 1) * n_lanes + t];
        const FqU px = load_fqu(pts + 32 * (size_t)ip), qx = load_fqu(pts + 32 * (size_t)iq);
#pragma unroll
        for (int l = 0; l < 14; ++l) prefix[((size_t)i * 14 + l) * n_lanes + t] = acc.l[l];
        acc = fqu_mul(acc, fqu_sub<4>(qx, px));
    }
    FqU inv = fqu_inverse(acc);
    for (uint32_t i = k_pairs; i-- > 0;) {
        const uint32_t ip = idx[(size_t)(2 * i) * n_lanes + t], iq = idx[(size_t)(2 * i + 1) * n_lanes + t];
        const G1AffineU p = load_affine_u(pts, ip), q = load_affine_u(pts, iq);
        FqU pre;
#pragma unroll
        for (int l = 0; l < 14; ++l) pre.l[l] = prefix[((size_t)i * 14 + l) * n_lanes + t];
        const FqU d = fqu_sub<4>(q.x, p.x);
        const FqU dinv = fqu_mul(inv, pre);
        inv = fqu_mul(inv, d);
        const FqU lam = fqu_mul(fqu_sub<4>(q.y, p.y), dinv);
        const FqU x3 = fqu_sub<4>(fqu_sub<4>(fqu_sqr(lam), p.x), q.x);             // < 10p
        const FqU y3 = fqu_sub<4>(fqu_mul(lam, fqu_sub<16>(p.x, x3)), p.y);        // < 6p
        uint32_t* o = out + 32 * ((size_t)i * n_lanes + t);
        store_fqu(o, x3);
        store_fqu(o + 16, y3);
    }
}
// the same pairs by XYZZ, compared with the batched results; *bad counts the differences
__global__ __launch_bounds__(256) void k_check(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ idx, uint32_t k_pairs, uint32_t n_lanes,
                                               const uint32_t* __restrict__ got, uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_lanes) return;
    for (uint32_t i = 0; i < k_pairs; ++i) {
        const uint32_t ip = idx[(size_t)(2 * i) * n_lanes + t], iq = idx[(size_t)(2 * i + 1) * n_lanes + t];
        G1XyzzU acc = G1XyzzU::identity();
        g1u_madd(acc, load_affine_u(pts, ip), false);
        g1u_madd(acc, load_affine_u(pts, iq), false);
        const uint32_t* o = got + 32 * ((size_t)i * n_lanes + t);
        const FqU x3 = load_fqu(o), y3 = load_fqu(o + 16);
        const bool ok = fqu_is_zero_mod_p(fqu_sub<16>(fqu_mul(x3, acc.zz), acc.x)) && fqu_is_zero_mod_p(fqu_sub<8>(fqu_mul(y3, acc.zzz), acc.y));
        if (!ok) atomicAdd(bad, 1u);
    }
}

// self-test of the pieces on one pair: inversion, the affine formulas without batching
__global__ void k_diag(const uint32_t* __restrict__ pts, uint32_t* __restrict__ flags) {
    const G1AffineU p = load_affine_u(pts, 1), q = load_affine_u(pts, 2);
    const FqU d = fqu_sub<4>(q.x, p.x);
    const FqU dinv = fqu_inverse(d);
    flags[0] = fqu_is_zero_mod_p(fqu_sub<4>(fqu_mul(d, dinv), FqU::one())) ? 1 : 0;
    const FqU lam = fqu_mul(fqu_sub<4>(q.y, p.y), dinv);
    const FqU x3 = fqu_sub<4>(fqu_sub<4>(fqu_sqr(lam), p.x), q.x);
    const FqU y3 = fqu_sub<4>(fqu_mul(lam, fqu_sub<16>(p.x, x3)), p.y);
    G1XyzzU acc = G1XyzzU::identity();
    g1u_madd(acc, p, false);
    g1u_madd(acc, q, false);
    flags[1] = fqu_is_zero_mod_p(fqu_sub<16>(fqu_mul(x3, acc.zz), acc.x)) ? 1 : 0;
    flags[2] = fqu_is_zero_mod_p(fqu_sub<8>(fqu_mul(y3, acc.zzz), acc.y)) ? 1 : 0;
    flags[3] = fqu_is_zero_mod_p(fqu_sub<4>(fqu_mul(lam, d), fqu_sub<4>(q.y, p.y))) ? 1 : 0;
}

int main() {
    const uint32_t n_pts = 1u << 20;            // 128 MiB of points, as the 2^20 commit's SRS
    const uint32_t n_lanes = 1u << 18;          // 4 waves per SIMD's worth of lanes
    uint32_t *d_pts, *d_idx, *d_prefix, *d_out, *d_bad;
    const uint32_t K_MAX = 256;
    CHK(hipMalloc(&d_pts, (size_t)n_pts * 128));
    CHK(hipMalloc(&d_idx, (size_t)2 * K_MAX * n_lanes * 4));
    CHK(hipMalloc(&d_prefix, (size_t)K_MAX * 14 * n_lanes * 4));
    CHK(hipMalloc(&d_out, (size_t)K_MAX * n_lanes * 128));
    CHK(hipMalloc(&d_bad, 4));
    {
        std::vector<uint32_t> h((size_t)n_pts * 32);
        uint64_t s = 88172645463325252ULL;
        auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 20); };
        for (size_t i = 0; i < n_pts; ++i)
            for (int c = 0; c < 2; ++c) {
                for (int l = 0; l < 13; ++l) h[i * 32 + c * 16 + l] = rnd() & 0x0fffffffu;
                h[i * 32 + c * 16 + 13] = rnd() & 0xffffu;              // value < 2^380 < p
            }
        CHK(hipMemcpy(d_pts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> ix((size_t)2 * K_MAX * n_lanes);
        for (auto& v : ix) v = rnd() & (n_pts - 1);
        for (size_t r = 1; r < (size_t)2 * K_MAX; r += 2)               // P != Q in every pair (see "no special cases" above)
            for (size_t t = 0; t < n_lanes; ++t)
                if (ix[r * n_lanes + t] == ix[(r - 1) * n_lanes + t]) ix[r * n_lanes + t] ^= 1u;
        CHK(hipMemcpy(d_idx, ix.data(), ix.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto timed = [&](auto launch, int reps, float* ms) -> int {
        launch();
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(ms, e0, e1));
        *ms /= reps;
        return 0;
    };
    const unsigned grid = n_lanes / 256;
    float ms;
    {
        uint32_t* d_flags;
        CHK(hipMalloc(&d_flags, 16));
        k_diag<<<1, 1>>>(d_pts, d_flags);
        uint32_t f[4];
        CHK(hipMemcpy(f, d_flags, 16, hipMemcpyDeviceToHost));
        printf("self-test: d * d^-1 == 1: %u, x3: %u, y3: %u, lambda * d == dy: %u\n", f[0], f[1], f[2], f[3]);
    }
    for (uint32_t len : {13u, 26u, 52u}) {
        if (timed([&] { k_xyzz<<<grid, 256>>>(d_pts, d_idx, len, n_lanes, d_out); }, 5, &ms)) return 1;
        printf("XYZZ mixed additions, lists of %2u per lane, %u lanes:            %8.1f us  %6.2f additions / ns\n", len, n_lanes, ms * 1e3,
               (double)len * n_lanes / (ms * 1e6));
    }
    for (uint32_t k : {16u, 32u, 64u, 128u, 256u}) {
        if (timed([&] { k_batched<<<grid, 256>>>(d_pts, d_idx, k, n_lanes, d_prefix, d_out); }, 3, &ms)) return 1;
        CHK(hipMemset(d_bad, 0, 4));
        k_check<<<grid, 256>>>(d_pts, d_idx, k, n_lanes, d_out, d_bad);
        uint32_t bad = 0;
        CHK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        char verdict[64];
        snprintf(verdict, sizeof verdict, bad ? "%u RESULTS DIFFER" : "results equal the XYZZ sums", bad);
        printf("batched affine, %3u pairs per lane and inversion, %u lanes:       %8.1f us  %6.2f additions / ns   %s\n", k, n_lanes, ms * 1e3,
               (double)k * n_lanes / (ms * 1e6), verdict);
    }
    for (uint32_t len : {26u, 52u}) {      // once more, after a second of load (clocks settled)
        if (timed([&] { k_xyzz<<<grid, 256>>>(d_pts, d_idx, len, n_lanes, d_out); }, 20, &ms)) return 1;
        printf("XYZZ mixed additions again, lists of %2u per lane, %u lanes:      %8.1f us  %6.2f additions / ns\n", len, n_lanes, ms * 1e3,
               (double)len * n_lanes / (ms * 1e6));
    }
    return 0;
}
