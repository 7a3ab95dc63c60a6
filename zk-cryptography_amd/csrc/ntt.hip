// ntt.hip -- C-ABI entry points of the NTT / Domain / polynomial product path.
// gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "host_fr.hpp"
#include "host_util.hpp"
#include "ntt_kernels.hpp"

using namespace zk;

// F::get_root_of_unity(2^log_n) (domain.rs:38): the 2^32-th root 7^((r-1)/2^32), squared (32 - log_n) times
static zkhost::Fr root_of_unity(uint32_t log_n) {
    zkhost::Fr g = zkhost::fr_from_u64(7);
    uint64_t e[4];
    uint64_t rm1[4] = {zkhost::FR_P[0] - 1, zkhost::FR_P[1], zkhost::FR_P[2], zkhost::FR_P[3]};
    for (int i = 0; i < 4; ++i) e[i] = (rm1[i] >> 32) | (i < 3 ? (rm1[i + 1] << 32) : 0);
    zkhost::Fr w = zkhost::fr_one();
    for (int i = 255; i >= 0; --i) {
        w = zkhost::fr_mul(w, w);
        if ((e[i / 64] >> (i % 64)) & 1) w = zkhost::fr_mul(w, g);
    }
    for (uint32_t i = log_n; i < 32; ++i) w = zkhost::fr_mul(w, w);
    return w;
}

extern "C" int zkhip_domain_params(uint64_t size, uint64_t* h_generator, uint64_t* h_generator_inv, uint64_t* h_size_inv) {
    if (!is_pow2((size_t)size) || !h_generator || !h_generator_inv || !h_size_inv) return ZKHIP_ERR_ARG;
    const uint32_t log_n = log2_exact((size_t)size);
    if (log_n > 32) return ZKHIP_ERR_SHAPE;   // get_root_of_unity(..).unwrap() panics beyond the 2-adicity
    zkhost::Fr w = root_of_unity(log_n);
    zkhost::Fr wi = zkhost::fr_inv(w);
    zkhost::Fr ni = zkhost::fr_inv(zkhost::fr_from_u64(size));
    std::memcpy(h_generator, w.l, 32);
    std::memcpy(h_generator_inv, wi.l, 32);
    std::memcpy(h_size_inv, ni.l, 32);
    return ZKHIP_OK;
}

// Twiddle tables and pass plans live in the CONTEXT that built them (zkhip_ctx::ntt_state): contexts on different host threads share
// nothing, and zkhip_ctx_destroy returns their device memory (they used to be process-wide maps keyed by the context's address).
struct TwiddleKey {
    uint32_t log_n; int inverse;
    bool operator<(const TwiddleKey& o) const {
        if (log_n != o.log_n) return log_n < o.log_n;
        return inverse < o.inverse;
    }
};
struct NttPass { uint32_t s0, T; uint64_t* table; };
struct NttPlan { uint64_t* tw1 = nullptr; std::vector<NttPass> passes; zkhost::Fr n_inv; };
struct NttCache {
    std::map<TwiddleKey, uint64_t*> twiddles;
    std::map<TwiddleKey, NttPlan> plans;
};
static void ntt_cache_free(void* p) {
    NttCache* nc = (NttCache*)p;
    for (auto& kv : nc->twiddles) (void)hipFree(kv.second);
    for (auto& kv : nc->plans) {
        if (kv.second.tw1) (void)hipFree(kv.second.tw1);
        for (auto& ps : kv.second.passes) if (ps.table) (void)hipFree(ps.table);
    }
    delete nc;
}
static NttCache* ntt_cache(zkhip_ctx* c) {
    if (!c->ntt_state) {
        c->ntt_state = new (std::nothrow) NttCache();
        c->ntt_free = ntt_cache_free;
    }
    return (NttCache*)c->ntt_state;
}

static int get_twiddles(zkhip_ctx* c, uint32_t log_n, int inverse, uint64_t** out) {
    NttCache* nc = ntt_cache(c);
    if (!nc) return ZKHIP_ERR_NOMEM;
    TwiddleKey key{log_n, inverse};
    auto it = nc->twiddles.find(key);
    if (it != nc->twiddles.end()) { *out = it->second; return ZKHIP_OK; }
    const uint32_t log_half = log_n ? log_n - 1 : 0;
    const size_t half = (size_t)1 << log_half;
    zkhost::Fr w = root_of_unity(log_n);
    if (inverse) w = zkhost::fr_inv(w);
    std::vector<zkhost::Fr> pw(log_half ? log_half : 1);
    for (uint32_t k = 0; k < log_half; ++k) { pw[k] = w; w = zkhost::fr_mul(w, w); }
    uint64_t *d_tab = nullptr, *d_pw = nullptr;
    ZK_HIP(c, hipMalloc(&d_tab, half * 32));
    ZK_HIP(c, hipMalloc(&d_pw, pw.size() * 32));
    ZK_HIP(c, hipMemcpyAsync(d_pw, pw.data(), pw.size() * 32, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3(mle_grid(half)), dim3(MLE_BLOCK), 0, c->stream, d_pw, log_half, d_tab);
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    ZK_HIP(c, hipFree(d_pw));
    nc->twiddles[key] = d_tab;
    *out = d_tab;
    return ZKHIP_OK;
}

// ---- transforms of >= 2^12 points (ntt_kernels.hpp, second half) ------------------------------------------------------------
static int get_plan(zkhip_ctx* c, uint32_t log_n, int inverse, NttPlan** out) {
    NttCache* nc = ntt_cache(c);
    if (!nc) return ZKHIP_ERR_NOMEM;
    TwiddleKey key{log_n, inverse};
    auto it = nc->plans.find(key);
    if (it != nc->plans.end()) { *out = &it->second; return ZKHIP_OK; }
    uint64_t* W = nullptr;
    ZK_TRY(get_twiddles(c, log_n, inverse, &W));
    NttPlan plan;
    plan.n_inv = zkhost::fr_inv(zkhost::fr_from_u64((uint64_t)1 << log_n));
    ZK_HIP(c, hipMalloc(&plan.tw1, 256 * 32));
    hipLaunchKernelGGL(ntt_first_table_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, W, log_n, plan.tw1);
    // the stages after the first eight, in passes of <= 7 spread evenly
    const uint32_t rest = log_n - NTT_FIRST_STAGES;
    const uint32_t n_pass = (rest + 6) / 7;
    uint32_t s0 = NTT_FIRST_STAGES;
    for (uint32_t p = 0; p < n_pass; ++p) {
        const uint32_t T = (rest - (s0 - NTT_FIRST_STAGES) + (n_pass - p) - 1) / (n_pass - p);
        NttPass ps{s0, T, nullptr};
        const size_t entries = (((size_t)1 << T) - 1) << s0;
        ZK_HIP(c, hipMalloc(&ps.table, entries * 32));
        FrArg sc = {};
        const bool scaled = inverse && p + 1 == n_pass;
        if (scaled) std::memcpy(sc.v, plan.n_inv.l, 32);
        hipLaunchKernelGGL(ntt_pass_table_kernel, dim3(mle_grid_stream(entries)), dim3(MLE_BLOCK), 0, c->stream, W, log_n, s0, T, sc,
                           scaled ? 1u : 0u, ps.table);
        plan.passes.push_back(ps);
        s0 += T;
    }
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *out = &(nc->plans[key] = plan);
    return ZKHIP_OK;
}

// src (n_src <= n elements, zero-padded; times src2 element-wise when given) -> dst (the first n_dst <= n outputs), which may
// be src itself; d_scratch: n elements.  The last pass writes dst, every earlier one d_scratch.
static int ntt_big(zkhip_ctx* c, const uint64_t* d_src, size_t n_src, const uint64_t* d_src2, uint64_t* d_dst, size_t n_dst,
                   uint32_t log_n, int inverse, uint64_t* d_scratch) {
    const size_t n = (size_t)1 << log_n;
    NttPlan* plan = nullptr;
    ZK_TRY(get_plan(c, log_n, inverse, &plan));
    const size_t lds = (size_t)NTT_BIG_TILE * 32;
    const unsigned grid = (unsigned)(n >> NTT_BIG_TILE_LOG);
    {
        ProfScope ps(c, "ntt_first8", 32.0 * (double)(n_src + n) + (d_src2 ? 32.0 * (double)n : 0.0));
        hipLaunchKernelGGL(ntt_first8_kernel, dim3(grid), dim3(NTT_BIG_BLOCK), lds, c->stream, d_src, n_src, d_src2, d_scratch, log_n, plan->tw1);
    }
    FrArg sc = {};
    std::memcpy(sc.v, plan->n_inv.l, 32);
    for (size_t p = 0; p < plan->passes.size(); ++p) {
        const NttPass& ps = plan->passes[p];
        const bool last = p + 1 == plan->passes.size();
        ProfScope pr(c, "ntt_pass", 32.0 * (double)(n + (last ? n_dst : n)) + 32.0 * (double)((((size_t)1 << ps.T) - 1) << ps.s0));
        if (last && inverse)
            hipLaunchKernelGGL(ntt_pass_kernel<true>, dim3(grid), dim3(NTT_BIG_BLOCK), lds, c->stream, d_scratch, d_dst, ps.s0, ps.T, ps.table, sc, n_dst);
        else
            hipLaunchKernelGGL(ntt_pass_kernel<false>, dim3(grid), dim3(NTT_BIG_BLOCK), lds, c->stream, d_scratch, last ? d_dst : d_scratch, ps.s0, ps.T,
                               ps.table, sc, last ? n_dst : n);
    }
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// d_data (n = 2^log_n, in place).  Forward: serial_fft with omega; inverse: omega^-1 then scale by n^-1.
static int ntt_inplace(zkhip_ctx* c, uint64_t* d_data, uint32_t log_n, int inverse, uint64_t* d_scratch) {
    const size_t n = (size_t)1 << log_n;
    if (log_n == 0) return ZKHIP_OK;
    if (log_n >= 12) return ntt_big(c, d_data, n, nullptr, d_data, n, log_n, inverse, d_scratch);
    uint64_t* tw = nullptr;
    ZK_TRY(get_twiddles(c, log_n, inverse, &tw));
    const uint32_t tile = (uint32_t)std::min<size_t>(NTT_TILE, n);
    {
        ProfScope ps(c, "ntt_first_stages", 64.0 * (double)n);
        hipLaunchKernelGGL(ntt_first_stages_kernel, dim3((unsigned)(n / tile)), dim3(MLE_BLOCK), 0, c->stream, d_data, d_scratch,
                           log_n, tw);
    }
    for (uint32_t s = NTT_TILE_LOG; s < log_n;) {
        uint32_t T = std::min<uint32_t>(NTT_MID_MAX, log_n - s);
        if (log_n - s > NTT_MID_MAX && log_n - s < 2 * NTT_MID_MAX) T = (log_n - s + 1) / 2;   // balance the last two passes
        ProfScope ps(c, "ntt_mid_stages", 64.0 * (double)n);
        hipLaunchKernelGGL(ntt_mid_stages_kernel, dim3((unsigned)(n >> NTT_MID_TILE_LOG)), dim3(MLE_BLOCK), 0, c->stream,
                           d_scratch, log_n, s, T, tw);
        s += T;
    }
    if (inverse) {
        zkhost::Fr ni = zkhost::fr_inv(zkhost::fr_from_u64((uint64_t)n));
        FrArg sc;
        std::memcpy(sc.v, ni.l, 32);
        hipLaunchKernelGGL(elementwise_kernel<2>, dim3(mle_grid_stream(n)), dim3(MLE_BLOCK), 0, c->stream, d_scratch,
                           (const uint64_t*)nullptr, sc, n, d_data);
    } else {
        ZK_HIP(c, hipMemcpyAsync(d_data, d_scratch, n * 32, hipMemcpyDeviceToDevice, c->stream));
    }
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_ntt(zkhip_ctx* c, uint64_t* d_data, uint32_t log_n, int inverse) {
    if (!c || !d_data) return ZKHIP_ERR_ARG;
    if (log_n > 30) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    const size_t n = (size_t)1 << log_n;
    ZK_TRY(c->reserve_ws(n * 32));
    return ntt_inplace(c, d_data, log_n, inverse, (uint64_t*)c->d_ws);
}

// Domain::fft / ifft as the reference calls them (domain.rs:108-118: clone, resize to the domain size with zeros, transform):
// d_src holds n_src <= 2^log_n values, d_dst receives the 2^log_n results; d_src is not modified (d_dst == d_src is allowed
// when n_src == 2^log_n).
extern "C" int zkhip_domain_transform(zkhip_ctx* c, const uint64_t* d_src, size_t n_src, uint64_t* d_dst, uint32_t log_n, int inverse) {
    if (!c || !d_dst || (n_src && !d_src)) return ZKHIP_ERR_ARG;
    if (log_n > 30) return ZKHIP_ERR_SHAPE;
    const size_t n = (size_t)1 << log_n;
    if (n_src > n) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    ZK_TRY(c->reserve_ws(n * 32));
    if (log_n >= 12) return ntt_big(c, d_src, n_src, nullptr, d_dst, n, log_n, inverse, (uint64_t*)c->d_ws);
    if (d_dst != d_src) {
        if (n_src < n) ZK_HIP(c, hipMemsetAsync(d_dst + 4 * n_src, 0, (n - n_src) * 32, c->stream));
        if (n_src) ZK_HIP(c, hipMemcpyAsync(d_dst, d_src, n_src * 32, hipMemcpyDeviceToDevice, c->stream));
    } else if (n_src != n) return ZKHIP_ERR_ARG;
    return ntt_inplace(c, d_dst, log_n, inverse, (uint64_t*)c->d_ws);
}

extern "C" int zkhip_pointwise_mul(zkhip_ctx* c, const uint64_t* d_a, const uint64_t* d_b, size_t n, uint64_t* d_out) {
    if (!c || !d_a || !d_b || !d_out) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    hipLaunchKernelGGL(pointwise_mul_kernel, dim3(mle_grid_stream(n)), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, n, d_out);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_univariate_multiply(zkhip_ctx* c, const uint64_t* d_a, size_t na, const uint64_t* d_b, size_t nb,
                                         uint64_t* d_out) {
    if (!c || !d_a || !d_b || !d_out) return ZKHIP_ERR_ARG;
    if (na == 0 || nb == 0) return ZKHIP_ERR_SHAPE;   // len_a + len_b - 1 underflows in the reference (evaluation.rs:66)
    ZK_TRY(c->activate());
    const size_t unscaled = na + nb - 1;
    uint32_t log_n = 0;
    while (((size_t)1 << log_n) < unscaled) ++log_n;
    if (log_n > 30) return ZKHIP_ERR_SHAPE;
    const size_t n = (size_t)1 << log_n;
    ZK_TRY(c->reserve_ws(3 * n * 32));
    uint64_t* scratch = (uint64_t*)c->d_ws;
    uint64_t* ea = scratch + 4 * n;
    uint64_t* eb = ea + 4 * n;
    if (log_n >= 12) {
        // three transforms and nothing else: the zero padding (resize, :73-74) happens in the forward transforms' gather, the
        // evaluation-form product (:79-82) in the inverse transform's gather, the truncation (:85) in its last pass
        ZK_TRY(ntt_big(c, d_a, na, nullptr, ea, n, log_n, 0, scratch));                              // domain.fft :76-77
        ZK_TRY(ntt_big(c, d_b, nb, nullptr, eb, n, log_n, 0, scratch));
        return ntt_big(c, ea, n, eb, d_out, unscaled, log_n, 1, scratch);                            // domain.ifft :84
    }
    ZK_HIP(c, hipMemsetAsync(ea, 0, 2 * n * 32, c->stream));                                        // resize(.., F::ZERO) :73-74
    ZK_HIP(c, hipMemcpyAsync(ea, d_a, na * 32, hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipMemcpyAsync(eb, d_b, nb * 32, hipMemcpyDeviceToDevice, c->stream));
    ZK_TRY(ntt_inplace(c, ea, log_n, 0, scratch));                                                   // domain.fft :76-77
    ZK_TRY(ntt_inplace(c, eb, log_n, 0, scratch));
    hipLaunchKernelGGL(pointwise_mul_kernel, dim3(mle_grid_stream(n)), dim3(MLE_BLOCK), 0, c->stream, ea, eb, n, ea);   // :79-82
    ZK_TRY(ntt_inplace(c, ea, log_n, 1, scratch));                                                   // domain.ifft :84
    ZK_HIP(c, hipMemcpyAsync(d_out, ea, unscaled * 32, hipMemcpyDeviceToDevice, c->stream));         // truncate :85
    return ZKHIP_OK;
}

// ---- host-side Fr helpers ------------------------------------------------------------------------------
extern "C" int zkhip_fr_from_i64(int64_t v, uint64_t* h_out) {
    if (!h_out) return ZKHIP_ERR_ARG;
    zkhost::Fr m = zkhost::fr_from_u64(v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v);
    if (v < 0) m = zkhost::fr_sub(zkhost::fr_zero(), m);
    std::memcpy(h_out, m.l, 32);
    return ZKHIP_OK;
}
extern "C" int zkhip_fr_to_canonical(const uint64_t* h_in, uint64_t* h_out) {
    if (!h_in || !h_out) return ZKHIP_ERR_ARG;
    zkhost::Fr a, one = zkhost::fr_zero();
    std::memcpy(a.l, h_in, 32);
    one.l[0] = 1;
    zkhost::Fr c = zkhost::fr_mul(a, one);
    std::memcpy(h_out, c.l, 32);
    return ZKHIP_OK;
}
#define ZK_FR_BINOP(NAME, FN)                                                                   \
    extern "C" int NAME(const uint64_t* h_a, const uint64_t* h_b, uint64_t* h_out) {            \
        if (!h_a || !h_b || !h_out) return ZKHIP_ERR_ARG;                                       \
        zkhost::Fr a, b;                                                                        \
        std::memcpy(a.l, h_a, 32); std::memcpy(b.l, h_b, 32);                                   \
        zkhost::Fr c = zkhost::FN(a, b);                                                        \
        std::memcpy(h_out, c.l, 32);                                                            \
        return ZKHIP_OK;                                                                        \
    }
ZK_FR_BINOP(zkhip_fr_add, fr_add)
ZK_FR_BINOP(zkhip_fr_sub, fr_sub)
ZK_FR_BINOP(zkhip_fr_mul, fr_mul)

// Synthetic benchmark inputs: splitmix64-seeded xoshiro256**; an element is four draws taken as a 256-bit
// little-endian integer with the top bit masked (255 bits), redrawn while >= r -- uniform over the field -- and handed out in
// the Montgomery form the tables hold.  Host side, deterministic for a seed whatever the machine.
extern "C" int zkhip_synthetic_fr(uint64_t seed, size_t n, uint64_t* h_out) {
    if (!h_out && n) return ZKHIP_ERR_ARG;
    uint64_t st[4];
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) {                      // splitmix64
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        st[i] = z ^ (z >> 31);
    }
    auto rotl = [](uint64_t v, int k) { return (v << k) | (v >> (64 - k)); };
    auto next = [&]() {                                // xoshiro256**
        const uint64_t r = rotl(st[1] * 5, 7) * 9, t = st[1] << 17;
        st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3];
        st[2] ^= t; st[3] = rotl(st[3], 45);
        return r;
    };
    zkhost::Fr r2;                                     // R^2 mod r: canonical -> Montgomery is one product with it
    std::memcpy(r2.l, zkhost::FR_R2, 32);
    for (size_t i = 0; i < n; ++i) {
        zkhost::Fr c;
        for (;;) {
            for (int k = 0; k < 4; ++k) c.l[k] = next();
            c.l[3] &= 0x7FFFFFFFFFFFFFFFull;
            bool lt = false;                           // c < r ?
            for (int k = 3; k >= 0; --k) {
                if (c.l[k] != zkhost::FR_P[k]) { lt = c.l[k] < zkhost::FR_P[k]; break; }
            }
            if (lt) break;
        }
        const zkhost::Fr m = zkhost::fr_mul(c, r2);    // c * R^2 * R^-1 = c R
        std::memcpy(h_out + 4 * i, m.l, 32);
    }
    return ZKHIP_OK;
}

