// mle_kernels.hpp -- multilinear evaluation-form kernels for gfx950.
//
// Replaces Multilinear::partial_evaluation / evaluation / sums
// (polynomial/src/multilinear/evaluation_form.rs:68-84,123-175) and the index
// generator pick_pairs_with_random_index (polynomial/src/utils.rs:26-53), whose
// pair list (i, i + n>>(k+1)) is computed in closed form here.
//
// HBM-bound streaming kernels: every lane moves whole 32-byte elements with
// 16-byte vector accesses, grid-stride over a grid of a few workgroups per CU.
// Algorithmic traffic of one fold of an n-entry table: read 32n + write 16n
// = 48n bytes.
#pragma once
#include "fp.hpp"
#include "transcript.hpp"

namespace zk {

constexpr int MLE_BLOCK = 256;
constexpr int MLE_MAX_GRID = 256 * 8;   // 8 workgroups of 4 waves per CU: 32 waves/CU
constexpr int TAIL_LOG = 12;            // tables of <= 2^12 entries (128 KiB of the CU's 160 KiB LDS) finish inside one workgroup
constexpr int TAIL_N = 1 << TAIL_LOG;

// Small host-side parameters travel as kernel arguments (captured at launch), never through a shared staging
// buffer: entry points return before the stream has run, so a staging slot could be overwritten by the next call.
struct FrArg { uint64_t v[4]; };
struct PtsArg { uint64_t v[4 * 48]; };   // up to ZK_MAX_ROUNDS points
__device__ __forceinline__ Fr fr_from_arg(const FrArg& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.l[2 * i] = (uint32_t)a.v[i]; r.l[2 * i + 1] = (uint32_t)(a.v[i] >> 32); }
    return r;
}
__device__ __forceinline__ Fr fr_from_pts(const PtsArg& a, uint32_t k) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.l[2 * i] = (uint32_t)a.v[4 * k + i]; r.l[2 * i + 1] = (uint32_t)(a.v[4 * k + i] >> 32); }
    return r;
}

// out[j] = lo + r*(hi - lo)  ==  r*hi + (1-r)*lo  (evaluation_form.rs:133), one Montgomery product.
// Both sides are canonical residues of the same field element, so limbs are identical.
__device__ __forceinline__ Fr fold_pair(const Fr& lo, const Fr& hi, const Fr& r) { return lo + r * (hi - lo); }

// input index of the "lo" partner of output j when folding variable k of an n-entry table:
// half = n >> (k+1) ; i = (j / half) * 2*half + (j % half)         (utils.rs:36-50)
__device__ __forceinline__ size_t fold_lo_index(size_t j, uint32_t log_half) {
    return ((j >> log_half) << (log_half + 1)) | (j & (((size_t)1 << log_half) - 1));
}

// Fold one variable.  WITH_SUMS additionally emits, per workgroup, the sums of its outputs that fall
// in the lower / upper half of the OUTPUT table (= next sumcheck round's
// split_poly_into_two_and_sum_each_part, evaluation_form.rs:68-74), saving a re-read of the output.
template <bool WITH_SUMS>
static __global__ __launch_bounds__(MLE_BLOCK) void fold_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                         size_t n_out, uint32_t log_half,
                                                         const uint64_t* __restrict__ r_ptr, FrArg r_val,
                                                         uint64_t* __restrict__ partials) {
    __shared__ Fr red[2 * MLE_BLOCK / 64];
    const Fr r = r_ptr ? load_fr(r_ptr, 0) : fr_from_arg(r_val);   // device-resident challenge, or a host point by value
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    const size_t half_in = (size_t)1 << log_half;
    const size_t half_out = n_out >> 1;
    Fr s_lo = Fr::zero(), s_hi = Fr::zero();
    for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < n_out; j += 2 * stride) {
        const size_t j2 = j + stride;
        const bool has2 = j2 < n_out;
        const size_t i1 = fold_lo_index(j, log_half);
        const size_t i2 = has2 ? fold_lo_index(j2, log_half) : i1;
        Fr a1 = load_fr(in, i1), b1 = load_fr(in, i1 + half_in);      // (non-temporal loads measured: 144.2 us either way at 2^24)
        Fr a2 = load_fr(in, i2), b2 = load_fr(in, i2 + half_in);
        Fr o1 = fold_pair(a1, b1, r);
        store_fr(out, j, o1);
        if (WITH_SUMS) {
            if (j < half_out) s_lo = s_lo + o1; else s_hi = s_hi + o1;
        }
        if (has2) {
            Fr o2 = fold_pair(a2, b2, r);
            store_fr(out, j2, o2);
            if (WITH_SUMS) {
                if (j2 < half_out) s_lo = s_lo + o2; else s_hi = s_hi + o2;
            }
        }
    }
    if (WITH_SUMS) {
        block_reduce_fr2(s_lo, s_hi, red);
        if (threadIdx.x == 0) {
            store_fr(partials, 2 * (size_t)blockIdx.x, s_lo);
            store_fr(partials, 2 * (size_t)blockIdx.x + 1, s_hi);
        }
    }
}

// Per-workgroup (lower-half sum, upper-half sum) of a table: split_poly_into_two_and_sum_each_part
// (evaluation_form.rs:68-74); their sum is sum_over_the_boolean_hypercube (:80-84) / poly_sum (sumcheck.rs:25-27).
static __global__ __launch_bounds__(MLE_BLOCK) void half_sums_kernel(const uint64_t* __restrict__ in, size_t n,
                                                              uint64_t* __restrict__ partials) {
    __shared__ Fr red[2 * MLE_BLOCK / 64];
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    const size_t half = n >> 1;
    Fr s_lo = Fr::zero(), s_hi = Fr::zero();
    for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < n; j += 4 * stride) {
        Fr v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t ju = j + u * stride;
            v[u] = (ju < n) ? load_fr(in, ju) : Fr::zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t ju = j + u * stride;
            if (ju < half) s_lo = s_lo + v[u]; else s_hi = s_hi + v[u];
        }
    }
    block_reduce_fr2(s_lo, s_hi, red);
    if (threadIdx.x == 0) {
        store_fr(partials, 2 * (size_t)blockIdx.x, s_lo);
        store_fr(partials, 2 * (size_t)blockIdx.x + 1, s_hi);
    }
}

// Workgroup-wide reduction of n_partials (lo, hi) pairs; result in thread 0.
__device__ __forceinline__ void reduce_partials(const uint64_t* __restrict__ partials, uint32_t n_partials, Fr* red,
                                                Fr& lo, Fr& hi) {
    lo = Fr::zero();
    hi = Fr::zero();
    for (uint32_t b = threadIdx.x; b < n_partials; b += 4 * blockDim.x) {   // 8 loads in flight per lane
        Fr vl[4], vh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t bu = b + u * blockDim.x;
            vl[u] = (bu < n_partials) ? load_fr(partials, 2 * (size_t)bu) : Fr::zero();
            vh[u] = (bu < n_partials) ? load_fr(partials, 2 * (size_t)bu + 1) : Fr::zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { lo = lo + vl[u]; hi = hi + vh[u]; }
    }
    block_reduce_fr2(lo, hi, red);
}

// Single-workgroup reduction: out[0] = lower-half sum, out[1] = upper-half sum, out[2] = total.
static __global__ __launch_bounds__(MLE_BLOCK) void finish_sums_kernel(const uint64_t* __restrict__ partials,
                                                                uint32_t n_partials, uint64_t* __restrict__ out) {
    __shared__ Fr red[2 * MLE_BLOCK / 64];
    Fr lo, hi;
    reduce_partials(partials, n_partials, red, lo, hi);
    if (threadIdx.x == 0) {
        store_fr(out, 0, lo);
        store_fr(out, 1, hi);
        store_fr(out, 2, lo + hi);
    }
}

// Finish an evaluation inside one workgroup: folds variable 0 repeatedly with points pts[0..n_pts) until the
// table (n <= TAIL_N entries, staged in LDS) has n >> n_pts entries left; writes them to out.
static __global__ __launch_bounds__(MLE_BLOCK) void fold_tail_kernel(const uint64_t* __restrict__ in, uint32_t n,
                                                              PtsArg pts, uint32_t first_pt, uint32_t n_pts,
                                                              uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    Fr* tab = reinterpret_cast<Fr*>(zk_dyn_lds);   // TAIL_N entries
    for (uint32_t j = threadIdx.x; j < n; j += MLE_BLOCK) tab[j] = load_fr(in, j);
    __syncthreads();
    uint32_t cur = n;
    for (uint32_t p = 0; p < n_pts; ++p) {
        const Fr r = fr_from_pts(pts, first_pt + p);
        const uint32_t half = cur >> 1;
        // in place: lane j reads (j, j+half) and writes j; no other lane touches index j this round
        for (uint32_t j = threadIdx.x; j < half; j += MLE_BLOCK) tab[j] = fold_pair(tab[j], tab[j + half], r);
        __syncthreads();
        cur = half;
    }
    for (uint32_t j = threadIdx.x; j < cur; j += MLE_BLOCK) store_fr(out, j, tab[j]);
}

// Outer sum / outer product of two tables (evaluation_form.rs:28-52): out[i*nb + j] = a[i] (+|*) b[j]
template <bool MUL>
static __global__ __launch_bounds__(MLE_BLOCK) void distinct_kernel(const uint64_t* __restrict__ a,
                                                             const uint64_t* __restrict__ b, size_t nb, size_t n_out,
                                                             uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t q = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; q < n_out; q += stride) {
        Fr x = load_fr(a, q / nb), y = load_fr(b, q % nb);
        store_fr(out, q, MUL ? x * y : x + y);
    }
}

// Elementwise ops used by the callers either side of the path (Add/Sub/Mul<F>, evaluation_form.rs:178-251)
template <int OP>   // 0 add, 1 sub, 2 scale by `scalar`
static __global__ __launch_bounds__(MLE_BLOCK) void elementwise_kernel(const uint64_t* __restrict__ a,
                                                                const uint64_t* __restrict__ b, FrArg scalar, size_t n,
                                                                uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    Fr s = Fr::zero();
    if (OP == 2) s = fr_from_arg(scalar);
    for (size_t q = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; q < n; q += stride) {
        Fr x = load_fr(a, q);
        Fr o = (OP == 0) ? x + load_fr(b, q) : (OP == 1) ? x - load_fr(b, q) : x * s;
        store_fr(out, q, o);
    }
}

// Canonical big-endian bytes of every element (Multilinear::to_bytes, evaluation_form.rs:54-62)
static __global__ __launch_bounds__(MLE_BLOCK) void to_bytes_kernel(const uint64_t* __restrict__ in, size_t n,
                                                             uint32_t* __restrict__ out_words) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t q = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; q < n; q += stride) {
        Fr c = load_fr(in, q).from_mont();
        uint4* p = reinterpret_cast<uint4*>(out_words + 8 * q);
        p[0] = make_uint4(__builtin_bswap32(c.l[7]), __builtin_bswap32(c.l[6]), __builtin_bswap32(c.l[5]),
                          __builtin_bswap32(c.l[4]));
        p[1] = make_uint4(__builtin_bswap32(c.l[3]), __builtin_bswap32(c.l[2]), __builtin_bswap32(c.l[1]),
                          __builtin_bswap32(c.l[0]));
    }
}

// One round of MultilinearKZG::open on an n-entry table: quotient[j] = in[j + n/2] - in[j]  (get_poly_quotient,
// kzg/src/utils.rs:12-17: f(1, .) - f(0, .)) and remainder[j] = in[j] + z (in[j + n/2] - in[j])  (get_poly_remainder,
// :5-10), j < n/2, from one read of the table.
static __global__ __launch_bounds__(MLE_BLOCK) void open_step_kernel(const uint64_t* __restrict__ in, size_t n, FrArg z_val,
                                                              uint64_t* __restrict__ quotient, uint64_t* __restrict__ remainder) {
    const Fr z = fr_from_arg(z_val);
    const size_t h = n >> 1, stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < h; j += stride) {
        const Fr lo = load_fr(in, j), hi = load_fr(in, j + h);
        const Fr d = hi - lo;
        store_fr(quotient, j, d);
        store_fr(remainder, j, lo + z * d);
    }
}

// All rounds of a SMALL opening (n <= 2^12) in one launch: a single workgroup walks the rounds, the remainders ping-pong between the
// two scratch tables exactly as the per-round launches leave them (round i reads `in` / ping / pong, writes quotients at their level's
// offset and the remainder to ping (i even) or pong (i odd)); a barrier between rounds -- one workgroup: what it stored is visible to it.
static __global__ __launch_bounds__(MLE_BLOCK) void open_steps_small_kernel(const uint64_t* __restrict__ in, uint32_t n, PtsArg z_pts, uint32_t n_rounds,
                                                                     uint64_t* __restrict__ quotients, uint64_t* ping, uint64_t* pong) {
    const uint64_t* cur = in;
    uint32_t cn = n, off = 0;
    for (uint32_t i = 0; i < n_rounds; ++i) {
        const Fr z = fr_from_pts(z_pts, i);
        uint64_t* rem = (i & 1) ? pong : ping;
        const uint32_t h = cn >> 1;
        for (uint32_t j = threadIdx.x; j < h; j += MLE_BLOCK) {
            const Fr lo = load_fr(cur, j), hi = load_fr(cur, (size_t)j + h);
            const Fr d = hi - lo;
            store_fr(quotients, (size_t)off + j, d);
            store_fr(rem, j, lo + z * d);
        }
        __syncthreads();
        off += h;
        cur = rem;
        cn = h;
    }
}

// ---- Horner suffix scan: UnivariateKZG::open ---------------------------------------------------------------
// V_i = sum_{j >= i} c_j z^(j-i)  (V_i = c_i + z V_{i+1}, V_n = 0).  V_0 = p(z) is DenseUnivariatePolynomial::evaluate
// (dense_univariate.rs:184-196) and V_1 .. V_{n-1} are the coefficients of the quotient (p(x) - k) / (x - z) for any
// constant k -- what univariate_kzg.rs:66-69 takes from divide_with_q_and_r (dense_univariate.rs:88-124; the constant only
// changes the remainder).  The recurrence is a scan: a lane owns HS_L coefficients, a workgroup HS_T lanes; pass 1 leaves
// every workgroup's value at its base for zero carry-in (block_vals), horner_top_kernel turns those into the carry
// entering each workgroup, pass 2 (APPLY) redoes the in-workgroup scan with that carry injected at the top lane and
// writes the quotient / the evaluation.
constexpr int HS_L = 8;
constexpr int HS_T = 256;
template <bool APPLY>
static __global__ __launch_bounds__(HS_T) void horner_scan_kernel(const uint64_t* __restrict__ coeffs, size_t n, FrArg z_val,
                                                           uint64_t* __restrict__ block_vals, const uint64_t* __restrict__ carries,
                                                           uint64_t* __restrict__ quotient, uint64_t* __restrict__ evaluation) {
    __shared__ Fr a_lds[HS_T];
    const Fr z = fr_from_arg(z_val);
    const uint32_t t = threadIdx.x;
    const size_t s = ((size_t)blockIdx.x * HS_T + t) * HS_L;
    Fr c[HS_L];
#pragma unroll
    for (int j = 0; j < HS_L; ++j) c[j] = (s + j < n) ? load_fr(coeffs, s + j) : Fr::zero();
    Fr h = c[HS_L - 1];
#pragma unroll
    for (int j = HS_L - 2; j >= 0; --j) h = c[j] + z * h;
    Fr rp = z;                                  // z^HS_L by squaring (HS_L = 8)
#pragma unroll
    for (int q = 1; q < HS_L; q <<= 1) rp = rp * rp;
    Fr carry_in = Fr::zero();
    if (APPLY) {
        carry_in = load_fr(carries, blockIdx.x);
        if (t == HS_T - 1) h = h + rp * carry_in;
    }
    a_lds[t] = h;
    __syncthreads();
    for (uint32_t d = 1; d < (uint32_t)HS_T; d <<= 1) {      // A_t += (z^L)^d A_{t+d}
        Fr v = Fr::zero();
        const bool has = t + d < (uint32_t)HS_T;
        if (has) v = a_lds[t + d];
        __syncthreads();
        if (has) { h = h + rp * v; a_lds[t] = h; }
        __syncthreads();
        rp = rp * rp;
    }
    if (!APPLY) {
        if (t == 0) store_fr(block_vals, blockIdx.x, h);
        return;
    }
    Fr v = (t == HS_T - 1) ? carry_in : a_lds[t + 1];        // V at the top of this lane's chunk
#pragma unroll
    for (int j = HS_L - 1; j >= 0; --j) {
        v = c[j] + z * v;                                    // V_{s+j}
        const size_t i = s + j;
        if (i == 0) store_fr(evaluation, 0, v);
        else if (i < n) store_fr(quotient, i - 1, v);
    }
}
// carries[b] = V at the base of workgroup b + 1 (0 for the last): Y_b = H_{b+1} + R Y_{b+1}, R = z^(HS_T HS_L); one workgroup
static __global__ __launch_bounds__(1024) void horner_top_kernel(const uint64_t* __restrict__ block_vals, uint32_t n_blocks, FrArg z_val,
                                                          uint64_t* __restrict__ carries, uint64_t* __restrict__ v0_out) {
    __shared__ Fr a_lds[1024];
    const uint32_t t = threadIdx.x;
    Fr R = fr_from_arg(z_val);
    for (int q = 1; q < HS_T * HS_L; q <<= 1) R = R * R;
    const uint32_t per = (n_blocks + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(lo + per, n_blocks);       // this lane's workgroups [lo, hi)
    Fr g = Fr::zero();
    for (uint32_t b = hi; b-- > lo;) g = load_fr(block_vals, b) + R * g;   // value at the base of `lo` for zero carry-in
    Fr rp = Fr::one();                                               // R^per
    for (uint32_t q = 0; q < per; ++q) rp = rp * R;
    a_lds[t] = g;
    __syncthreads();
    Fr h = g;
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        Fr v = Fr::zero();
        const bool has = t + d < 1024;
        if (has) v = a_lds[t + d];
        __syncthreads();
        if (has) { h = h + rp * v; a_lds[t] = h; }
        __syncthreads();
        rp = rp * rp;
    }
    Fr y = (t == 1023) ? Fr::zero() : a_lds[t + 1];                  // V at the base of workgroup `hi`
    for (uint32_t b = hi; b-- > lo;) {
        store_fr(carries, b, y);
        y = load_fr(block_vals, b) + R * y;
    }
    if (t == 0 && v0_out) store_fr(v0_out, 0, y);   // V at the base of workgroup 0 = p(z)
}
// DenseUnivariatePolynomial::degree (dense_univariate.rs:199-207): index of the last non-zero coefficient, 0 if none;
// *out must be zeroed first
static __global__ __launch_bounds__(MLE_BLOCK) void dense_degree_kernel(const uint64_t* __restrict__ coeffs, size_t n,
                                                                 unsigned long long* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n; i += stride)
        if (!load_fr(coeffs, i).is_zero()) best = i;
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_down(best, d, 64);
        best = best > o ? best : o;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

// add_to_front / add_to_back (evaluation_form.rs:86-110): out[i] = in[i mod n_in] (shift = 0) or in[i >> shift]
static __global__ __launch_bounds__(MLE_BLOCK) void repeat_kernel(const uint64_t* __restrict__ in, size_t n_in, uint32_t shift, size_t n_out,
                                                           uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n_out; i += stride)
        store_fr(out, i, load_fr(in, shift ? (i >> shift) : (i & (n_in - 1))));
}
// Circuit::evaluation, one layer (circuit/src/circuit.rs:41-50): gate = (type u8 | in0 u32 | in1 u32) packed as 3 words
static __global__ __launch_bounds__(MLE_BLOCK) void circuit_layer_kernel(const uint64_t* __restrict__ in, const uint32_t* __restrict__ gates,
                                                                  size_t n_gates, uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t g = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; g < n_gates; g += stride) {
        const uint32_t type = gates[3 * g], a = gates[3 * g + 1], b = gates[3 * g + 2];
        const Fr x = load_fr(in, a), y = load_fr(in, b);
        store_fr(out, g, type ? x * y : x + y);
    }
}
// Circuit::add_mult_mle (circuit.rs:59-97): the tables are zeroed first; one lane per gate sets its one
static __global__ __launch_bounds__(MLE_BLOCK) void wiring_ones_kernel(const uint32_t* __restrict__ gates, size_t n_gates, uint32_t shift,
                                                                uint64_t* __restrict__ add, uint64_t* __restrict__ mul) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t g = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; g < n_gates; g += stride) {
        const uint32_t type = gates[3 * g], a = gates[3 * g + 1], b = gates[3 * g + 2];
        const size_t idx = (g << (2 * shift)) | ((size_t)a << shift) | b;
        store_fr(type ? mul : add, idx, Fr::one());
    }
}

// For kernels that only stream (no per-workgroup output to reduce afterwards): up to 64 workgroups per CU's worth of
// grid.  Measured on the one-variable fold at 2^24: 163 us with the 2048-workgroup cap (8 iterations per lane), 144 us from
// 16384 workgroups on (one or two iterations per lane; the dispatcher balances the tail and more loads are in flight).
inline int mle_grid_stream(size_t n_items) {
    size_t g = (n_items + MLE_BLOCK - 1) / MLE_BLOCK;
    if (g < 1) g = 1;
    if (g > (size_t)MLE_MAX_GRID * 8) g = (size_t)MLE_MAX_GRID * 8;
    return (int)g;
}
inline int mle_grid(size_t n_items) {
    size_t g = (n_items + MLE_BLOCK - 1) / MLE_BLOCK;
    if (g < 1) g = 1;
    if (g > (size_t)MLE_MAX_GRID) g = MLE_MAX_GRID;
    return (int)g;
}

}  // namespace zk
