// gkr.hip -- GKRProtocol::prove as one C-ABI call: the host orchestration of the reference's prover over the
// device-resident tables.  Every table operation below is one of libzkhip's own entry points (HIP kernels);
// the outer Fiat-Shamir transcript absorbs the layer's proof bytes and runs on the host -- the ONE point per layer where the
// host waits for the GPU; everything else of a layer (both halves of its sumcheck, eq tables, V(u), w_b, w_c) is enqueued ahead.
// gfx950 only.  No CPU fallback: the tables never leave HBM.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "host_util.hpp"
#include "host_fr.hpp"
#include "mle_kernels.hpp"
#include "outer_transcript.hpp"
#include "shard.hpp"

// ---- a layer's sumcheck in time linear in its width -------------------------------------------------------------
// The reference proves, per layer,   sum_{b,c} add~(b,c) (V(b) + V(c)) + mul~(b,c) V(b) V(c)   with the multi-composed
// prover over DENSE tables of all (b, c): 2^(2s) entries for a layer of 2^s values (and builds add~ / mul~ from dense
// 0/1 tables over (a, b, c), 2^(3l+2) entries, folded over the gate variables: protocol.rs:67-87, circuit.rs:59-97).
// Both are sums over the GATES in disguise.  With w_g the gate's weight (its eq factor at r_b / r_c, times alpha / beta):
//   * rounds over b (the first s variables): summing c out first leaves   V(b) Ha0(b) + Ha1(b)   and   V(b) Hm(b)   with
//       Ha0(b) = sum_{add gates g, in0 = b} w_g,  Ha1(b) = sum_{add, in0 = b} w_g V(in1),  Hm(b) = sum_{mul, in0 = b} w_g V(in1);
//     the round polynomials of the dense prover are those of these two terms over s variables (folding commutes with
//     the sum over c), term by term -- which matters because each term's zero coefficients are dropped separately;
//   * rounds over c, b fixed at the challenges u:   Aa(c) (V(u) + V(c))   and   Am(c) (V(u) V(c))   with
//       Aa(c) = sum_{add, in1 = c} w_g eq_{in0}(u),  Am(c) likewise: again two product terms, over s variables.
// Every table has 2^s entries and is built by one pass over the gates (grouped by in0 / in1 on the host), field
// arithmetic is exact, so sums, round polynomials, challenges and w_b, w_c are those of the dense prover, bit for bit --
// and a layer of 2^20 values (dense: 2^40 entries) is a few dozen launches over 32 MiB tables.
namespace zk {
// gate weights: w_g = eq_g(r_b) (one point) or alpha eq_g(r_b) + beta eq_g(r_c); eq over the n_gate_vars index bits, MSB first
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_gate_weights_kernel(uint32_t n_gates, uint32_t n_gate_vars, PtsArg r_b, PtsArg r_c,
                                                                     FrArg alpha_v, FrArg beta_v, uint32_t two_points,
                                                                     uint64_t* __restrict__ wg) {
    const uint32_t g = blockIdx.x * MLE_BLOCK + threadIdx.x;
    if (g >= n_gates) return;
    const Fr one = Fr::one();
    Fr eb = one, ec = one;
    for (uint32_t j = 0; j < n_gate_vars; ++j) {
        const bool bit = (g >> (n_gate_vars - 1 - j)) & 1;
        const Fr tb = fr_from_pts(r_b, j);
        eb = eb * (bit ? tb : one - tb);
        if (two_points) {
            const Fr tc = fr_from_pts(r_c, j);
            ec = ec * (bit ? tc : one - tc);
        }
    }
    store_fr(wg, g, two_points ? fr_from_arg(alpha_v) * eb + fr_from_arg(beta_v) * ec : eb);
}
// rows of a CSR grouping of the gates by one of their inputs: row_off[x] .. row_off[x+1] index `ids` (gate numbers).
// phase 1 (rows = in0):  add_out[x] = sum_{add} w_g (Ha0),  lin_out[x] = sum_{add} w_g V[in1] (Ha1),  mul_out[x] = sum_{mul} w_g V[in1] (Hm)
// phase 2 (rows = in1):  add_out[x] = sum_{add} w_g eq_u[in0] (Aa),  mul_out[x] = sum_{mul} w_g eq_u[in0] (Am);  lin_out unused
// one row (every argument as gkr_gate_rows_kernel's)
struct GateRowsArgs {
    const uint32_t *row_off, *ids;
    const uint8_t* gate_type;
    const uint32_t* other_in;
    const uint64_t *wg, *factor;
    uint32_t n_rows, phase;
    uint64_t *add_out, *lin_out, *mul_out;
    const uint64_t *v, *vu_ptr;
    uint64_t *t1, *t2;
    uint32_t row_stride, row_first;
    uint64_t* v_shard;
    const uint64_t *eq_b, *eq_c;
    FrArg alpha_v, beta_v;
    uint64_t* wg_out;
    const uint64_t* ab_dev;
};
__device__ __forceinline__ void gkr_gate_row(const GateRowsArgs& ga, uint32_t j) {
    const uint32_t* __restrict__ row_off = ga.row_off; const uint32_t* __restrict__ ids = ga.ids;
    const uint8_t* __restrict__ gate_type = ga.gate_type; const uint32_t* __restrict__ other_in = ga.other_in;
    const uint64_t* __restrict__ wg = ga.wg; const uint64_t* __restrict__ factor = ga.factor;
    const uint32_t phase = ga.phase, row_stride = ga.row_stride, row_first = ga.row_first;
    uint64_t* add_out = ga.add_out; uint64_t* lin_out = ga.lin_out; uint64_t* mul_out = ga.mul_out;
    const uint64_t* v = ga.v; const uint64_t* vu_ptr = ga.vu_ptr; uint64_t* t1 = ga.t1; uint64_t* t2 = ga.t2; uint64_t* v_shard = ga.v_shard;
    const uint64_t* eq_b = ga.eq_b; const uint64_t* eq_c = ga.eq_c; const FrArg& alpha_v = ga.alpha_v; const FrArg& beta_v = ga.beta_v;
    uint64_t* wg_out = ga.wg_out; const uint64_t* ab_dev = ga.ab_dev;
    const uint32_t x = j * row_stride + row_first;
    if (t1) {   // phase 2 also lays out the other factor of each term (one launch less per layer): t1[c] = V(u) + V[c], t2[c] = V(u) V[c]
        const Fr vu = load_fr(vu_ptr, 0), vx = load_fr(v, x);
        store_fr(t1, j, vu + vx);
        store_fr(t2, j, vu * vx);
    }
    if (v_shard) store_fr(v_shard, j, load_fr(factor, x));            // phase 1: factor = V
    Fr a = Fr::zero(), l = Fr::zero(), m = Fr::zero();
    for (uint32_t q = row_off[x]; q < row_off[x + 1]; ++q) {
        const uint32_t g = ids[q];
        Fr w;
        if (eq_b) {
            const Fr al = ab_dev ? load_fr(ab_dev, 0) : fr_from_arg(alpha_v), be = ab_dev ? load_fr(ab_dev, 1) : fr_from_arg(beta_v);
            w = al * load_fr(eq_b, g) + be * load_fr(eq_c, g);
            store_fr(wg_out, g, w);
        } else {
            w = load_fr(wg, g);
        }
        const Fr wf = w * load_fr(factor, other_in[g]);     // phase 1: w_g V[in1];  phase 2: w_g eq_u[in0]
        if (gate_type[g]) m = m + wf;
        else if (phase == 1) { a = a + w; l = l + wf; }
        else a = a + wf;
    }
    store_fr(add_out, j, a);
    if (phase == 1) store_fr(lin_out, j, l);
    store_fr(mul_out, j, m);
}
// A rank of a sharded proof builds only ITS rows: lane j takes row x = j * row_stride + row_first and writes entry j of the outputs
// (row_stride = world, row_first = rank: the rank-interleaved shard the sumcheck sweeps); v_shard (phase 1, optional) receives V[x].
// What the rows GATHER from -- the gate weights, V, eq(u) -- is indexed by arbitrary wires and stays whole on every rank.
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_gate_rows_kernel(const uint32_t* __restrict__ row_off, const uint32_t* __restrict__ ids,
                                                                  const uint8_t* __restrict__ gate_type, const uint32_t* __restrict__ other_in,
                                                                  const uint64_t* __restrict__ wg, const uint64_t* __restrict__ factor,
                                                                  uint32_t n_rows, uint32_t phase, uint64_t* __restrict__ add_out,
                                                                  uint64_t* __restrict__ lin_out, uint64_t* __restrict__ mul_out,
                                                                  const uint64_t* __restrict__ v, const uint64_t* __restrict__ vu_ptr,
                                                                  uint64_t* __restrict__ t1, uint64_t* __restrict__ t2,
                                                                  uint32_t row_stride = 1, uint32_t row_first = 0, uint64_t* __restrict__ v_shard = nullptr,
                                                                  const uint64_t* __restrict__ eq_b = nullptr, const uint64_t* __restrict__ eq_c = nullptr,
                                                                  FrArg alpha_v = FrArg(), FrArg beta_v = FrArg(), uint64_t* __restrict__ wg_out = nullptr,
                                                                  const uint64_t* __restrict__ ab_dev = nullptr) {
    // ab_dev (not null): alpha and beta lie in device memory (entries 0 and 1), where gkr_layer_finish_kernel left them
    // eq_b (phase 1 of a single-GPU proof, every gate visited exactly once): the gate weights are formed HERE from the previous layer's two
    // eq tables -- w_g = alpha eq_g(r_b) + beta eq_g(r_c), the tables that layer built for its own second phase and for w_c -- and
    // filed in wg_out for phase 2: no weight launches per layer
    const uint32_t j = blockIdx.x * MLE_BLOCK + threadIdx.x;
    if (j >= n_rows) return;
    const GateRowsArgs a = {row_off, ids, gate_type, other_in, wg, factor, n_rows, phase, add_out, lin_out, mul_out, v, vu_ptr, t1, t2,
                            row_stride, row_first, v_shard, eq_b, eq_c, alpha_v, beta_v, wg_out, ab_dev};
    gkr_gate_row(a, j);
}
// eq_x(u) over n_vars index bits, MSB first (the b side of the wiring at the phase-1 challenges).  The points are read from
// DEVICE memory -- where the sumcheck that produced them left them -- so the host never waits for them.
// With v: also the workgroup's share of <eq(u), v> into partials[blockIdx.x] (V(u) = sum_x eq_x(u) V[x]: the same field element as
// MultilinearTrait::evaluation's chain of folds, exact arithmetic) -- the evaluation costs no pass of its own.
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_eq_table_kernel(const uint64_t* __restrict__ u, uint32_t n_vars, uint64_t* __restrict__ out,
                                                                 const uint64_t* __restrict__ v, uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const size_t n = (size_t)1 << n_vars, stride = (size_t)gridDim.x * MLE_BLOCK;
    const Fr one = Fr::one();
    Fr dot = Fr::zero();
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n; i += stride) {
        Fr acc = one;
        for (uint32_t j = 0; j < n_vars; ++j) {
            const Fr t = load_fr(u, j);
            acc = acc * (((i >> (n_vars - 1 - j)) & 1) ? t : one - t);
        }
        store_fr(out, i, acc);
        if (v) dot = dot + acc * load_fr(v, i);
    }
    if (v) {
        dot = block_reduce_fr(dot, red);
        if (threadIdx.x == 0) store_fr(partials, blockIdx.x, dot);
    }
}
// Wide layers (more than GKR_EQ_LO index bits) build their eq tables from two halves instead of n_vars products per entry:
//   eq_x(u) = hi[x >> n_lo] * lo[x & (2^n_lo - 1)],  hi over the first n_vars - n_lo points, lo over the last n_lo,
// one product per entry after a tiny kernel for the halves (exact arithmetic: the same field elements).  blockIdx.y selects
// the point (gate weights take two, scaled by alpha / beta through their hi halves); the points come from device memory
// (u_dev, the challenges a sumcheck left there) or by value.
constexpr uint32_t GKR_EQ_LO = 12;                         // (n_vars <= 2 GKR_EQ_LO = 24: the widest layer of a depth-24 circuit)
constexpr uint32_t GKR_EQ_HALVES = 2u << GKR_EQ_LO;        // entries reserved per point: hi (<= 2^12) then lo (2^12)
// A lone wave pays 0.9 us per product and 2 us per dependent load, so an entry is NOT built as a chain over its index bits
// (10 loads + 10 products: 28 us per launch): the points are fetched once into LDS with their complements, each half is split
// again into two parts of <= 6 bits whose <= 64-entry tables the 256 lanes build (6 products deep), and an entry is one product of two
// part entries -- 7-8 products deep in all.  Every workgroup rebuilds the part tables (they are tiny).
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_eq_halves_kernel(const uint64_t* __restrict__ u_dev, PtsArg p0, PtsArg p1, uint32_t n_vars,
                                                                  FrArg scale0, FrArg scale1, uint32_t scaled, uint64_t* __restrict__ out) {
    __shared__ Fr fac[2 * GKR_EQ_LO + 2][2];   // [variable][bit]: 1 - t, t
    __shared__ Fr part[2][2][64];              // [half: hi, lo][part: leading bits, trailing bits][index]
    static_assert(MLE_BLOCK == 256 && GKR_EQ_LO <= 12, "four part tables of <= 64 entries, a lane each");
    const uint32_t n_lo = GKR_EQ_LO, n_hi = n_vars - n_lo, pt = blockIdx.y, tid = threadIdx.x;
    {
        // by-value points are read with a UNIFORM index (scalar loads from the kernel arguments): indexing them by lane would put
        // both 640-byte arrays into scratch memory, and a dispatch that needs scratch costs ~15 us more
        Fr t = Fr::zero();
        if (u_dev) {
            if (tid < n_vars) t = load_fr(u_dev, tid);
        } else {
            for (uint32_t j = 0; j < n_vars; ++j) {
                const Fr tj = pt ? fr_from_pts(p1, j) : fr_from_pts(p0, j);
                if (tid == j) t = tj;
            }
        }
        if (tid < n_vars) {
            fac[tid][1] = t;
            fac[tid][0] = Fr::one() - t;
        }
    }
    __syncthreads();
    // part tables: half h covers the variables [first, first + cnt), its part 0 the leading cnt - cnt/2 of them
    {
        const uint32_t hh = tid >> 7, pp = (tid >> 6) & 1, idx = tid & 63;
        const uint32_t first = hh ? n_hi : 0, cnt = hh ? n_lo : n_hi;
        const uint32_t nb = cnt / 2, na = cnt - nb;                    // bits of part 0 / part 1
        const uint32_t bits = pp ? nb : na, v0 = first + (pp ? na : 0);
        if (idx < (1u << bits)) {
            Fr acc = (hh == 0 && pp == 0 && scaled) ? fr_from_arg(pt ? scale1 : scale0) : Fr::one();
            for (uint32_t j = 0; j < bits; ++j) acc = acc * fac[v0 + j][(idx >> (bits - 1 - j)) & 1];
            part[hh][pp][idx] = acc;
        }
    }
    __syncthreads();
    const uint32_t e = blockIdx.x * MLE_BLOCK + tid;
    const uint32_t hi_cnt = 1u << n_hi;
    if (e >= hi_cnt + (1u << n_lo)) return;
    const bool is_hi = e < hi_cnt;
    const uint32_t idx = is_hi ? e : e - hi_cnt, cnt = is_hi ? n_hi : n_lo, nb = cnt / 2;
    const Fr v = part[is_hi ? 0 : 1][0][idx >> nb] * part[is_hi ? 0 : 1][1][idx & ((1u << nb) - 1)];
    store_fr(out, (size_t)pt * GKR_EQ_HALVES + (is_hi ? idx : (GKR_EQ_HALVES >> 1) + idx), v);
}
// out[x] = sum over the n_points points of hi[x >> GKR_EQ_LO] * lo[x & (2^GKR_EQ_LO - 1)]
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_eq_expand_kernel(const uint64_t* __restrict__ halves, uint32_t n_points, size_t n,
                                                                  uint64_t* __restrict__ out, const uint64_t* __restrict__ v,
                                                                  uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    Fr dot = Fr::zero();
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t h = (uint32_t)(i >> GKR_EQ_LO), l = (uint32_t)(i & ((1u << GKR_EQ_LO) - 1));
        Fr acc = load_fr(halves, h) * load_fr(halves, (GKR_EQ_HALVES >> 1) + l);
        if (n_points == 2) acc = acc + load_fr(halves, GKR_EQ_HALVES + h) * load_fr(halves, GKR_EQ_HALVES + (GKR_EQ_HALVES >> 1) + l);
        store_fr(out, i, acc);
        if (v) dot = dot + acc * load_fr(v, i);
    }
    if (v) {   // as gkr_eq_table_kernel
        dot = block_reduce_fr(dot, red);
        if (threadIdx.x == 0) store_fr(partials, blockIdx.x, dot);
    }
}
// the per-workgroup shares of <eq(point), V> (gkr_eq_table_kernel / gkr_eq_expand_kernel) summed; the result stays on the device
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_dot_finish_kernel(const uint64_t* __restrict__ partials, uint32_t n_partials,
                                                                   uint64_t* __restrict__ out) {
    __shared__ Fr red[MLE_BLOCK / 64];
    Fr s = Fr::zero();
    for (uint32_t i = threadIdx.x; i < n_partials; i += MLE_BLOCK) s = s + load_fr(partials, i);
    s = block_reduce_fr(s, red);
    if (threadIdx.x == 0) store_fr(out, 0, s);
}
// The end of a layer with the outer transcript on the device (composed_kernels.hpp "an OUTER transcript"): every round of the layer's
// sumcheck has been absorbed by the hasher workgroups; alpha, beta = two challenges (protocol.rs:104-105), the next claim
// alpha w_b + beta w_c (:107), all left in device memory for the next layer's first kernels -- no host round trip per layer.
// next: alpha | beta | claimed (4 u64 each); evals: w_b | w_c.  One wave.
// ---- layers of at most GKR_SMALL_ROWS rows: their small launches, fused ---------------------------------------------------------------
// A layer is gate rows | rounds over b | eq(u), w_b | gate rows | rounds over c | eq(r_c), w_c | alpha, beta: on a small layer every launch
// between the sumchecks is a single workgroup that lives ~5 us whatever it does, so eq(u) + w_b + the second phase's rows are ONE
// launch, and eq(r_c) + w_c + alpha, beta + the NEXT layer's first rows another: 3 launches per layer instead of 6 (depth 8: 1.37 ->
// 1.2x ms).  The same device code as the kernels they replace, a barrier between the steps (one workgroup: what a step stored is
// visible to the next).
constexpr uint32_t GKR_SMALL_ROWS = 1024;
__device__ __forceinline__ void gkr_eq_table_block(const uint64_t* __restrict__ u, uint32_t n_vars, uint64_t* __restrict__ out,
                                                   const uint64_t* __restrict__ v, uint64_t* __restrict__ dot_out, Fr* red) {
    const size_t n = (size_t)1 << n_vars;
    const Fr one = Fr::one();
    Fr dot = Fr::zero();
    for (size_t i = threadIdx.x; i < n; i += MLE_BLOCK) {
        Fr acc = one;
        for (uint32_t j = 0; j < n_vars; ++j) {
            const Fr t = load_fr(u, j);
            acc = acc * (((i >> (n_vars - 1 - j)) & 1) ? t : one - t);
        }
        store_fr(out, i, acc);
        dot = dot + acc * load_fr(v, i);
    }
    dot = block_reduce_fr(dot, red);
    if (threadIdx.x == 0) store_fr(dot_out, 0, dot);
}
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_small_mid_kernel(const uint64_t* __restrict__ u, uint32_t n_vars, uint64_t* __restrict__ eq_out,
                                                                  const uint64_t* __restrict__ v, uint64_t* __restrict__ dot_out, GateRowsArgs rows) {
    __shared__ Fr red[MLE_BLOCK / 64];
    gkr_eq_table_block(u, n_vars, eq_out, v, dot_out, red);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rows.n_rows; j += MLE_BLOCK) gkr_gate_row(rows, j);
}
struct LayerFinishArgs {
    OuterDev* outer;
    const uint64_t* evals;
    uint64_t *next, *wb_out, *wc_out, *sum_out;
};
__device__ __forceinline__ void gkr_layer_finish_wave(const LayerFinishArgs& f, uint32_t* lds /* 64 words */) {        // one wave, all of its lanes
    Transcript tr;
    tr.load(&f.outer->state);
    const Fr alpha = tr.challenge_fr_wave(lds), beta = tr.challenge_fr_wave(lds);
    const Fr wb = load_fr(f.evals, 0), wc = load_fr(f.evals, 1);
    const Fr claimed = fr_mul_outlined(alpha, wb) + fr_mul_outlined(beta, wc);
    if ((threadIdx.x & 63) == 0) {
        tr.store(&f.outer->state);
        store_fr(f.next, 0, alpha);
        store_fr(f.next, 1, beta);
        store_fr(f.next, 2, claimed);
        store_fr(f.wb_out, 0, wb);
        store_fr(f.wc_out, 0, wc);
        if (f.sum_out) store_fr(f.sum_out, 0, claimed);
    }
}
static __global__ __launch_bounds__(64) void gkr_layer_finish_kernel(LayerFinishArgs fin) {
    __shared__ uint32_t kw[64];
    gkr_layer_finish_wave(fin, kw);
}
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_small_end_kernel(const uint64_t* __restrict__ u, uint32_t n_vars, uint64_t* __restrict__ eq_out,
                                                                  const uint64_t* __restrict__ v, uint64_t* __restrict__ dot_out, LayerFinishArgs fin,
                                                                  uint32_t with_rows, GateRowsArgs rows) {
    __shared__ Fr red[MLE_BLOCK / 64];
    __shared__ uint32_t kw[64];
    gkr_eq_table_block(u, n_vars, eq_out, v, dot_out, red);
    __syncthreads();
    if (threadIdx.x < 64) gkr_layer_finish_wave(fin, kw);
    if (!with_rows) return;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rows.n_rows; j += MLE_BLOCK) gkr_gate_row(rows, j);
}
// eq_x(u) for all x < 2^n_vars, u in device memory; with d_v also <eq(u), v> -> d_dot (one value), summed from the table
// kernel's per-workgroup shares (d_partials: MLE_MAX_GRID entries of scratch)
static void launch_eq_table(zkhip_ctx* c, const uint64_t* d_u, uint32_t n_vars, uint64_t* d_halves, uint64_t* d_out,
                            const uint64_t* d_v = nullptr, uint64_t* d_partials = nullptr, uint64_t* d_dot = nullptr) {
    const size_t n = (size_t)1 << n_vars;
    const int grid = d_v ? mle_grid(n) : mle_grid_stream(n);          // one share per workgroup: the capped grid
    uint64_t* shares = grid == 1 ? d_dot : d_partials;                 // a single workgroup's share IS the result
    if (n_vars <= GKR_EQ_LO) {
        hipLaunchKernelGGL(gkr_eq_table_kernel, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_u, n_vars, d_out, d_v, shares);
    } else {
        PtsArg none = {};
        FrArg z = {};
        const uint32_t cnt = (1u << (n_vars - GKR_EQ_LO)) + (1u << GKR_EQ_LO);
        hipLaunchKernelGGL(gkr_eq_halves_kernel, dim3((cnt + MLE_BLOCK - 1) / MLE_BLOCK, 1), dim3(MLE_BLOCK), 0, c->stream, d_u, none, none, n_vars, z, z, 0u, d_halves);
        hipLaunchKernelGGL(gkr_eq_expand_kernel, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_halves, 1u, n, d_out, d_v, shares);
    }
    if (d_v && grid > 1)
        hipLaunchKernelGGL(gkr_dot_finish_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_partials, (uint32_t)grid, d_dot);
}
// w_g = eq_g(r_b)  or  alpha eq_g(r_b) + beta eq_g(r_c)
static void launch_gate_weights(zkhip_ctx* c, size_t n_gates, uint32_t n_gate_vars, const PtsArg& pb, const PtsArg& pc, const FrArg& av,
                                const FrArg& bv, bool two_points, uint64_t* d_halves, uint64_t* d_wg) {
    if (!n_gates) return;
    if (n_gate_vars <= GKR_EQ_LO) {
        hipLaunchKernelGGL(gkr_gate_weights_kernel, dim3((unsigned)((n_gates + MLE_BLOCK - 1) / MLE_BLOCK)), dim3(MLE_BLOCK), 0, c->stream,
                           (uint32_t)n_gates, n_gate_vars, pb, pc, av, bv, two_points ? 1u : 0u, d_wg);
        return;
    }
    const uint32_t cnt = (1u << (n_gate_vars - GKR_EQ_LO)) + (1u << GKR_EQ_LO);
    hipLaunchKernelGGL(gkr_eq_halves_kernel, dim3((cnt + MLE_BLOCK - 1) / MLE_BLOCK, two_points ? 2 : 1), dim3(MLE_BLOCK), 0, c->stream,
                       (const uint64_t*)nullptr, pb, pc, n_gate_vars, av, bv, two_points ? 1u : 0u, d_halves);
    hipLaunchKernelGGL(gkr_eq_expand_kernel, dim3(mle_grid_stream(n_gates)), dim3(MLE_BLOCK), 0, c->stream, d_halves, two_points ? 2u : 1u, n_gates, d_wg, (const uint64_t*)nullptr, (uint64_t*)nullptr);
}
}  // namespace zk

int zk_multi_composed_enqueue(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes, const uint64_t* const* lin_ptrs,
                              uint32_t n_terms, size_t n, const uint64_t* h_sum, int cont, uint32_t out_base, const ZkMcExtra* ex = nullptr);   // composed.hip
int zk_multi_composed_collect(zkhip_ctx* c, uint32_t n_rounds, uint32_t* h_lens, uint64_t* h_round_polys, uint64_t* h_challenges);
const uint64_t* zk_composed_challenges_dev(zkhip_ctx* c);

namespace {

constexpr int GKR_MONO = 7;   // monomials per round polynomial in zkhip_multi_composed_prove's output

struct LayerOut {
    uint64_t *sums, *round_polys, *wb, *wc, *challenges;
    uint32_t *n_rounds, *lens;
    uint32_t stride;   // rounds reserved per proof
};

// to_bytes of a ComposedSumcheckProof (multi_composed_sumcheck.rs:24-31): per round, per monomial coeff || pow as
// 32-byte big-endian canonical integers (sparse_univariate.rs:27-34)
void absorb_proof(zkhost::Transcript& tr, const uint64_t* polys, const uint32_t* lens, uint32_t n_rounds) {
    uint8_t be[GKR_MONO * 64];                            // one round's bytes, one update per round
    for (uint32_t r = 0; r < n_rounds; ++r) {
        const uint32_t n_mono = lens[r] <= (uint32_t)GKR_MONO ? lens[r] : (uint32_t)GKR_MONO;
        for (uint32_t m = 0; m < n_mono; ++m)
            for (int part = 0; part < 2; ++part)
                zkhost::fr_mont_to_be(polys + ((size_t)r * GKR_MONO + m) * 8 + 4 * part, be + 64 * m + 32 * part);
        tr.commit(be, 64 * (size_t)n_mono);
    }
}

// CSR grouping of gates by key[g] in [0, n_rows): row offsets + gate ids (counting sort; stable, so sums run in gate order)
static void group_gates(const uint32_t* key, size_t n_gates, size_t n_rows, std::vector<uint32_t>& csr /* row_off (n_rows+1) | ids */) {
    csr.assign(n_rows + 1 + n_gates, 0);
    for (size_t g = 0; g < n_gates; ++g) csr[key[g] + 1]++;
    for (size_t x = 0; x < n_rows; ++x) csr[x + 1] += csr[x];
    std::vector<uint32_t> cur(csr.begin(), csr.begin() + n_rows);
    for (size_t g = 0; g < n_gates; ++g) csr[n_rows + 1 + cur[key[g]]++] = (uint32_t)g;
}

// device scratch of one layer (carved from the context's aux buffer)
struct LayerScratch {
    uint64_t *wg, *ha0, *ha1, *hm, *equ, *eqc, *aa, *am, *t1, *t2;   // equ / eqc: eq(r_b) and eq(r_c) of the layer just proved (the next layer's gate weights)
    uint64_t* eqh;                    // halves of the eq tables of wide layers (2 points x GKR_EQ_HALVES entries)
    uint64_t *dot_partials, *evals;   // workgroup shares of <eq, V>; evals[0..4) = V(u) = w_b, evals[4..8) = V(r_c) = w_c
};
// one layer of a device-resident circuit: the gate arrays and their two CSR groupings (by in0 and by in1)
struct LayerDev {
    size_t n_gates, w_len;   // w_len = 2^(l + 1) entries of the layer's input table
    bool bad_label;          // a gate label out of range: reported when the prover reaches the layer, as the reference panics there
    uint32_t *csr0, *csr1, *in0, *in1;
    uint8_t* type;
};

// One layer: the sumcheck of generate_layer_one_prove_sumcheck (gkr/src/utils.rs:27-55) / the loop body of
// GKRProtocol::prove (protocol.rs:64-107) in the linear-time form described above, then w_b, w_c, alpha, beta, the next claim.
int layer_prove(zkhip_ctx* c, const LayerDev& ld, uint32_t l, const uint64_t* d_w, size_t w_len, const LayerScratch& sc,
                zkhost::Fr& claimed, zkhost::Transcript& tr, const LayerOut& out,
                uint32_t k, zkhost::Fr& alpha, zkhost::Fr& beta, std::vector<zkhost::Fr>& r_b, std::vector<zkhost::Fr>& r_c, bool two_points) {
    using namespace zk;
    const size_t n_gates = ld.n_gates;
    const uint32_t s = log2_exact(w_len);                  // variables of b (and of c)
    const uint32_t n_gate_vars = l == 0 ? 1u : l;          // binary_string(a, layer_index) has at least one bit (circuit/src/utils.rs:27-33)
    if (s != l + 1 || w_len != ld.w_len || 2 * s > out.stride || s < 1) return ZKHIP_ERR_SHAPE;   // the wiring index has l + 1 bits for b and for c
    if (r_b.size() != n_gate_vars || (two_points && r_c.size() != n_gate_vars)) return ZKHIP_ERR_SHAPE;
    if (ld.bad_label) return ZKHIP_ERR_INDEX;              // add_evaluations[gate_decimal] out of bounds (circuit.rs:73-93)
    PtsArg pb = {}, pc = {};
    std::memcpy(pb.v, r_b[0].l, 32 * r_b.size());
    if (two_points) std::memcpy(pc.v, r_c[0].l, 32 * r_c.size());
    FrArg av = {}, bv = {};
    std::memcpy(av.v, alpha.l, 32);
    std::memcpy(bv.v, beta.l, 32);
    const unsigned gw = (unsigned)((w_len + MLE_BLOCK - 1) / MLE_BLOCK);
    // gate weights: the first layer's from n_r (host); every other layer's inside its first row pass, from the eq tables of r_b and r_c
    // that the layer before left in sc.equ / sc.eqc (2^l entries: the gate index has l bits)
    const bool weights_on_the_fly = two_points && n_gates <= ((size_t)1 << n_gate_vars);
    if (!weights_on_the_fly) launch_gate_weights(c, n_gates, n_gate_vars, pb, pc, av, bv, two_points, sc.eqh, sc.wg);
    // ---- rounds over b
    hipLaunchKernelGGL(gkr_gate_rows_kernel, dim3(gw), dim3(MLE_BLOCK), 0, c->stream, ld.csr0, ld.csr0 + w_len + 1, ld.type, ld.in1, sc.wg, d_w,
                       (uint32_t)w_len, 1u, sc.ha0, sc.ha1, sc.hm, (const uint64_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr,
                       1u, 0u, (uint64_t*)nullptr, weights_on_the_fly ? (const uint64_t*)sc.equ : (const uint64_t*)nullptr, (const uint64_t*)sc.eqc, av, bv, sc.wg);
    ZK_HIP(c, hipGetLastError());
    // Everything up to the end of the layer's sumcheck is enqueued without waiting for the host: the rounds over c read the
    // challenges of the rounds over b, V(u) and the eq table from device memory, the two composed-prover calls append to one
    // array of recorded rounds, and the host reads rounds, challenges, w_b and w_c back once.
    const uint32_t nv = 2 * s;
    uint64_t* polys = out.round_polys + (size_t)k * out.stride * GKR_MONO * 8;
    uint32_t* lens = out.lens + (size_t)k * out.stride;
    std::vector<uint64_t> challenges(4 * (size_t)nv);
    const uint32_t sizes[2] = {2, 2};
    const uint64_t* d_ch = zk_composed_challenges_dev(c);
    {
        const uint64_t* tables[4] = {sc.ha0, d_w, sc.hm, d_w};          // [Ha0, V] + Ha1,  [Hm, V]
        const uint64_t* lin[2] = {sc.ha1, nullptr};
        ZK_TRY(zk_multi_composed_enqueue(c, tables, sizes, lin, 2, w_len, claimed.l, 0, 0));
    }
    // ---- rounds over c, b at u = the challenges just recorded
    launch_eq_table(c, d_ch, s, sc.eqh, sc.equ, d_w, sc.dot_partials, sc.evals);   // eq(u) and V(u) = <eq(u), V>: w_b and the factor of the second phase
    hipLaunchKernelGGL(gkr_gate_rows_kernel, dim3(gw), dim3(MLE_BLOCK), 0, c->stream, ld.csr1, ld.csr1 + w_len + 1, ld.type, ld.in0, sc.wg, sc.equ,
                       (uint32_t)w_len, 2u, sc.aa, (uint64_t*)nullptr, sc.am, d_w, (const uint64_t*)sc.evals, sc.t1, sc.t2);
    ZK_HIP(c, hipGetLastError());
    {
        const uint64_t* tables[4] = {sc.aa, sc.t1, sc.am, sc.t2};        // [add~(u, c), V(u) + V(c)],  [mul~(u, c), V(u) V(c)]
        ZK_TRY(zk_multi_composed_enqueue(c, tables, sizes, nullptr, 2, w_len, nullptr, 1, s));
    }
    // w_c = V(r_c), r_c = the second half of the challenges (its eq table stays for the next layer's gate weights, beside eq(u) = eq(r_b))
    launch_eq_table(c, d_ch + 4 * (size_t)s, s, sc.eqh, sc.eqc, d_w, sc.dot_partials, sc.evals + 4);
    ZK_HIP(c, hipGetLastError());
    zkhost::Fr eval_wb, eval_wc;
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), sc.evals, 64, hipMemcpyDeviceToHost, c->stream));
    ZK_TRY(zk_multi_composed_collect(c, nv, lens, polys, challenges.data()));   // synchronises the stream
    std::memcpy(eval_wb.l, c->pinned_u64(ZK_PIN_RES), 32);
    std::memcpy(eval_wc.l, c->pinned_u64(ZK_PIN_RES) + 4, 32);
    if (out.challenges) std::memcpy(out.challenges + (size_t)k * out.stride * 4, challenges.data(), 32 * (size_t)nv);
    std::memcpy(out.sums + 4 * (size_t)k, claimed.l, 32);
    out.n_rounds[k] = nv;
    absorb_proof(tr, polys, lens, nv);                                     // transcript.commit(&sumcheck_proof.to_bytes())
    r_b.assign(s, zkhost::fr_zero());                                      // challenges.split_at(len / 2)
    std::memcpy(r_b.data(), challenges.data(), 32 * (size_t)s);
    r_c.assign(s, zkhost::fr_zero());
    std::memcpy(r_c.data(), challenges.data() + 4 * (size_t)s, 32 * (size_t)s);
    std::memcpy(out.wb + 4 * (size_t)k, eval_wb.l, 32);
    std::memcpy(out.wc + 4 * (size_t)k, eval_wc.l, 32);
    alpha = tr.challenge_fr();
    beta = tr.challenge_fr();
    claimed = zkhost::fr_add(zkhost::fr_mul(alpha, eval_wb), zkhost::fr_mul(beta, eval_wc));
    return ZKHIP_OK;
}


// ---- the same layer with the OUTER transcript on the device: nothing here waits for the host --------------------------------------------
// What a layer needs from the one before -- alpha, beta, the claimed sum -- lies in device memory (d_next: alpha | beta | claimed, left
// by gkr_layer_finish_kernel; layer one takes them by value: 1, 0 and w_0(n_r)), its rounds are recorded in the proof arena instead of
// the context's small buffer (ar_ch | ar_rp: the layer's slice), and every closing kernel feeds the outer transcript (ex.outer).
struct DeviceTranscript {
    zk::OuterDev* outer;
    uint64_t* next;          // alpha | beta | claimed
    uint64_t *sums, *wb, *wc;   // per layer, 4 u64 each
    uint64_t* arena;         // per layer: challenges (4 x ZK_MAX_ROUNDS) | round polynomials (64 x ZK_MAX_ROUNDS)
    static constexpr size_t LAYER_U64 = (size_t)(4 + 64) * ZK_MAX_ROUNDS;
};
// rows1_done: this layer's first rows were built by the layer before (gkr_small_end_kernel); next / d_w_next / w_len_next: the layer after
// this one (null: none), whose first rows this layer's last launch builds when both are small -- *next_rows1_done says so.
// replayable (zkhip_gkr_prove_batch's lanes replay the launch chain as a HIP graph): nothing of this proof may travel BY VALUE in a kernel
// argument -- layer one then reads its claimed sum from dt.sums and finds its gate weights (1 - n_r, n_r) uploaded in sc.wg.
int layer_enqueue_device(zkhip_ctx* c, const LayerDev& ld, uint32_t l, const uint64_t* d_w, size_t w_len, const LayerScratch& sc,
                         const DeviceTranscript& dt, const zkhost::Fr& claimed0, const zkhost::Fr& n_r, uint32_t stride, bool rows1_done,
                         const LayerDev* next, const uint64_t* d_w_next, size_t w_len_next, bool* next_rows1_done, bool replayable = false) {
    using namespace zk;
    const size_t n_gates = ld.n_gates;
    const uint32_t s = log2_exact(w_len);
    const uint32_t n_gate_vars = l == 0 ? 1u : l;
    const bool two_points = l > 0;
    if (s != l + 1 || w_len != ld.w_len || 2 * s > stride || s < 1) return ZKHIP_ERR_SHAPE;
    if (ld.bad_label) return ZKHIP_ERR_INDEX;
    if (two_points && n_gates > ((size_t)1 << n_gate_vars)) return ZKHIP_ERR_SHAPE;      // (gate g has l index bits: zkhip_circuit_create flags anything else)
    const unsigned gw = (unsigned)((w_len + MLE_BLOCK - 1) / MLE_BLOCK);
    FrArg av = {}, bv = {};
    if (!two_points) {
        PtsArg pb = {}, pc = {};
        std::memcpy(pb.v, n_r.l, 32);
        const zkhost::Fr one = zkhost::fr_one();
        std::memcpy(av.v, one.l, 32);
        if (!replayable) launch_gate_weights(c, n_gates, n_gate_vars, pb, pc, av, bv, false, sc.eqh, sc.wg);
        else if (n_gates > 2) return ZKHIP_ERR_SHAPE;       // (one gate variable: the caller uploaded at most two weights)
    }
    static const bool fuse_small = [] { const char* e = std::getenv("ZKHIP_GKR_FUSE_SMALL"); return !e || std::atoi(e) != 0; }();
    const bool small = fuse_small && w_len <= GKR_SMALL_ROWS;
    auto rows1_args = [&](const LayerDev& L, const uint64_t* V, size_t rows, bool two) {       // the first rows of layer L (every gate once: the weights are formed there)
        GateRowsArgs ra = {};
        ra.row_off = L.csr0; ra.ids = L.csr0 + rows + 1; ra.gate_type = L.type; ra.other_in = L.in1; ra.wg = sc.wg; ra.factor = V;
        ra.n_rows = (uint32_t)rows; ra.phase = 1u; ra.add_out = sc.ha0; ra.lin_out = sc.ha1; ra.mul_out = sc.hm;
        ra.row_stride = 1u; ra.row_first = 0u;
        ra.eq_b = two ? (const uint64_t*)sc.equ : nullptr; ra.eq_c = sc.eqc; ra.alpha_v = av; ra.beta_v = bv; ra.wg_out = sc.wg;
        ra.ab_dev = two ? (const uint64_t*)dt.next : nullptr;
        return ra;
    };
    if (!rows1_done) {
        const GateRowsArgs ra = rows1_args(ld, d_w, w_len, two_points);
        hipLaunchKernelGGL(gkr_gate_rows_kernel, dim3(gw), dim3(MLE_BLOCK), 0, c->stream, ra.row_off, ra.ids, ra.gate_type, ra.other_in, ra.wg, ra.factor,
                           ra.n_rows, 1u, ra.add_out, ra.lin_out, ra.mul_out, (const uint64_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                           (uint64_t*)nullptr, 1u, 0u, (uint64_t*)nullptr, ra.eq_b, ra.eq_c, av, bv, ra.wg_out, ra.ab_dev);
        ZK_HIP(c, hipGetLastError());
    }
    uint64_t* ar_ch = dt.arena + (size_t)l * DeviceTranscript::LAYER_U64;
    uint64_t* ar_rp = ar_ch + 4 * (size_t)ZK_MAX_ROUNDS;
    ZkMcExtra ex = {};
    ex.d_sum = two_points ? dt.next + 8 : replayable ? dt.sums : nullptr;
    ex.outer = dt.outer;
    ex.d_round_polys = ar_rp;
    ex.d_challenges = ar_ch;
    const uint32_t sizes[2] = {2, 2};
    {
        const uint64_t* tables[4] = {sc.ha0, d_w, sc.hm, d_w};          // [Ha0, V] + Ha1,  [Hm, V]
        const uint64_t* lin[2] = {sc.ha1, nullptr};
        ex.token = ++c->outer_token;
        ZK_TRY(zk_multi_composed_enqueue(c, tables, sizes, lin, 2, w_len, two_points || replayable ? nullptr : claimed0.l, 0, 0, &ex));
    }
    // ---- rounds over c, b at u = the challenges just recorded (in the arena)
    if (small) {                                           // eq(u), w_b and the second rows: one workgroup, one launch
        GateRowsArgs ra = {};
        ra.row_off = ld.csr1; ra.ids = ld.csr1 + w_len + 1; ra.gate_type = ld.type; ra.other_in = ld.in0; ra.wg = sc.wg; ra.factor = sc.equ;
        ra.n_rows = (uint32_t)w_len; ra.phase = 2u; ra.add_out = sc.aa; ra.mul_out = sc.am; ra.v = d_w; ra.vu_ptr = sc.evals; ra.t1 = sc.t1; ra.t2 = sc.t2;
        ra.row_stride = 1u;
        hipLaunchKernelGGL(gkr_small_mid_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, (const uint64_t*)ar_ch, s, sc.equ, d_w, sc.evals, ra);
    } else {
        launch_eq_table(c, ar_ch, s, sc.eqh, sc.equ, d_w, sc.dot_partials, sc.evals);
        hipLaunchKernelGGL(gkr_gate_rows_kernel, dim3(gw), dim3(MLE_BLOCK), 0, c->stream, ld.csr1, ld.csr1 + w_len + 1, ld.type, ld.in0, sc.wg, sc.equ,
                           (uint32_t)w_len, 2u, sc.aa, (uint64_t*)nullptr, sc.am, d_w, (const uint64_t*)sc.evals, sc.t1, sc.t2);
    }
    ZK_HIP(c, hipGetLastError());
    {
        const uint64_t* tables[4] = {sc.aa, sc.t1, sc.am, sc.t2};
        ex.d_sum = nullptr;
        ex.token = ++c->outer_token;
        ZK_TRY(zk_multi_composed_enqueue(c, tables, sizes, nullptr, 2, w_len, nullptr, 1, s, &ex));
    }
    // w_c = V(r_c) (eq(r_c) stays for the next layer); alpha, beta, the next claim; w_b, w_c and the next layer's claimed sum into the per-layer arrays
    const LayerFinishArgs fin = {dt.outer, sc.evals, dt.next, dt.wb + 4 * (size_t)l, dt.wc + 4 * (size_t)l, dt.sums + 4 * (size_t)(l + 1)};
    *next_rows1_done = false;
    if (small) {                                           // ... and the next layer's first rows, if that layer is small too
        const bool with_rows = next && w_len_next <= GKR_SMALL_ROWS && !next->bad_label && w_len_next == next->w_len && w_len_next == 2 * w_len &&
                               next->n_gates <= ((size_t)1 << (l + 1));      // (what the next call checks before it would launch them itself)
        const GateRowsArgs ra = with_rows ? rows1_args(*next, d_w_next, w_len_next, true) : GateRowsArgs();
        hipLaunchKernelGGL(gkr_small_end_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, (const uint64_t*)(ar_ch + 4 * (size_t)s), s, sc.eqc, d_w, sc.evals + 4,
                           fin, with_rows ? 1u : 0u, ra);
        *next_rows1_done = with_rows;
    } else {
        launch_eq_table(c, ar_ch + 4 * (size_t)s, s, sc.eqh, sc.eqc, d_w, sc.dot_partials, sc.evals + 4);
        hipLaunchKernelGGL(gkr_layer_finish_kernel, dim3(1), dim3(64), 0, c->stream, fin);
    }
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

}  // namespace

// ---- device-resident circuit ---------------------------------------------------------------------------------------
struct zkhip_circuit {
    zkhip_ctx* c = nullptr;
    uint32_t n_layers = 0;
    std::vector<LayerDev> layers;
    void* d_mem = nullptr;
};

extern "C" void zkhip_circuit_destroy(zkhip_circuit* cir) {
    if (!cir) return;
    if (cir->c) {       // the graphs the batch lanes recorded for this circuit hold its addresses (and another circuit may get this very address)
        for (zkhip_ctx* lane : cir->c->gkr_lanes) {
            zkhip_ctx::GkrGraph& gg = lane->gkr_graph;
            if (gg.cir == cir || gg.warm_cir == cir) {
                if (gg.exec) { (void)hipStreamSynchronize(lane->stream); (void)hipGraphExecDestroy((hipGraphExec_t)gg.exec); }
                gg = zkhip_ctx::GkrGraph();
            }
        }
    }
    if (cir->d_mem && cir->c && cir->c->activate() == ZKHIP_OK) (void)hipFree(cir->d_mem);
    delete cir;
}

extern "C" int zkhip_circuit_create(zkhip_ctx* c, uint32_t n_layers, const size_t* h_n_gates, const uint8_t* h_gate_type,
                                    const uint32_t* h_in0, const uint32_t* h_in1, zkhip_circuit** out) {
    if (!c || !h_n_gates || !h_gate_type || !h_in0 || !h_in1 || !out) return ZKHIP_ERR_ARG;
    if (n_layers < 1 || 2 * n_layers > (uint32_t)ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;   // 2 (l + 1) sumcheck rounds for layer l
    ZK_TRY(c->activate());
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // layer l: 2^l gate slots (one bit at l = 0), inputs are labels of the 2^(l + 1)-entry table below it
    size_t total = 0, g_off = 0;
    std::vector<char> bad(n_layers, 0);
    for (uint32_t l = 0; l < n_layers; ++l) {
        const size_t w_len = (size_t)2 << l, ng = h_n_gates[l];
        const uint32_t n_gate_vars = l == 0 ? 1u : l;
        for (size_t g = 0; g < ng && !bad[l]; ++g)
            if (h_in0[g_off + g] >= w_len || h_in1[g_off + g] >= w_len || (g >> n_gate_vars)) bad[l] = 1;
        total += 2 * al(4 * (w_len + 1 + ng)) + 2 * al(4 * ng) + al(ng);
        g_off += ng;
    }
    zkhip_circuit* cir = new (std::nothrow) zkhip_circuit();
    if (!cir) return ZKHIP_ERR_NOMEM;
    cir->c = c;
    cir->n_layers = n_layers;
    if (hipMalloc(&cir->d_mem, total ? total : 256) != hipSuccess) {
        delete cir;
        return ZKHIP_ERR_NOMEM;
    }
    // stage everything in one host buffer, one upload
    std::vector<char> host(total);
    std::vector<uint32_t> csr;
    size_t off = 0;
    g_off = 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        const size_t w_len = (size_t)2 << l, ng = h_n_gates[l];
        LayerDev ld = {};
        ld.n_gates = ng;
        ld.w_len = w_len;
        ld.bad_label = bad[l] != 0;
        char* base = (char*)cir->d_mem;
        if (!bad[l]) {
            group_gates(h_in0 + g_off, ng, w_len, csr);
            std::memcpy(host.data() + off, csr.data(), 4 * csr.size());
        }
        ld.csr0 = (uint32_t*)(base + off); off += al(4 * (w_len + 1 + ng));
        if (!bad[l]) {
            group_gates(h_in1 + g_off, ng, w_len, csr);
            std::memcpy(host.data() + off, csr.data(), 4 * csr.size());
        }
        ld.csr1 = (uint32_t*)(base + off); off += al(4 * (w_len + 1 + ng));
        if (ng) std::memcpy(host.data() + off, h_in0 + g_off, 4 * ng);
        ld.in0 = (uint32_t*)(base + off); off += al(4 * ng);
        if (ng) std::memcpy(host.data() + off, h_in1 + g_off, 4 * ng);
        ld.in1 = (uint32_t*)(base + off); off += al(4 * ng);
        if (ng) std::memcpy(host.data() + off, h_gate_type + g_off, ng);
        ld.type = (uint8_t*)(base + off); off += al(ng);
        cir->layers.push_back(ld);
        g_off += ng;
    }
    if (total && (hipMemcpyAsync(cir->d_mem, host.data(), total, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                  hipStreamSynchronize(c->stream) != hipSuccess)) {
        zkhip_circuit_destroy(cir);
        return ZKHIP_ERR_HIP;
    }
    *out = cir;
    return ZKHIP_OK;
}

// GKRProtocol::prove of `cir` on context c -- the circuit's own, or a lane of zkhip_gkr_prove_batch (the circuit's device arrays are
// read-only here: any context of the same device may prove it, each with its own stream, scratch and transcript state)
static int gkr_prove_circuit_on(zkhip_ctx* c, zkhip_circuit* cir, const uint64_t* const* h_layer_ptrs, const size_t* h_layer_len, uint64_t* h_sums,
                                uint32_t* h_n_rounds, uint32_t* h_round_poly_lens, uint64_t* h_round_polys, uint64_t* h_wb,
                                uint64_t* h_wc, uint64_t* h_w0, uint64_t* h_challenges) {
    const uint32_t n_layers = cir->n_layers;
    if (h_layer_len[0] != 1) return ZKHIP_ERR_SHAPE;              // w_0 = [output.., 0] must have 2^k entries; the wiring of layer 0 has one gate bit
    for (uint32_t k = 1; k <= n_layers; ++k)
        if (!is_pow2(h_layer_len[k])) return ZKHIP_ERR_SHAPE;  // Multilinear::new (evaluation_form.rs:16-20)
    ZK_TRY(c->activate());
    // The lanes of a batch run on threads of their own, and since proofs are handed out one by one they are no longer in step: a lane's first
    // proof of a circuit allocates (workspace, pinned slots, the lane's copy of the layer values), its second RECORDS the chain -- and a
    // stream capture in one thread is invalidated by allocations in another whatever the capture mode (hipErrorStreamCaptureInvalidated, and the
    // allocating proof fails with it too: tools/stress_parity.py, 9 proofs on 8 lanes).  So a lane that has no graph of this circuit yet takes
    // its turn under the owner's mutex; a lane that replays (the steady state) does not.
    std::unique_lock<std::mutex> warm_turn;
    if (c->gkr_lane && c->gkr_parent && !(c->gkr_graph.exec && c->gkr_graph.cir == cir)) warm_turn = std::unique_lock<std::mutex>(c->gkr_parent->gkr_warm_mu);
    // A lane of zkhip_gkr_prove_batch REPLAYS the proof's launch chain as a HIP graph (one hipGraphLaunch instead of 130-400 launches of
    // 3-5 us of host time each: with eight chains side by side the process's launch rate was the bound, tools/perf_gkr_batch.py).  A graph
    // holds addresses: the layer values are first copied to a buffer of the lane's own, so that every proof of the circuit on this lane is
    // the same chain on the same memory.  (ZKHIP_GKR_GRAPH=0: plain launches.)
    static const bool graph_env = [] { const char* e = std::getenv("ZKHIP_GKR_GRAPH"); return !e || std::atoi(e) != 0; }();
    const bool lane = c->gkr_lane && graph_env && !c->profiling;
    std::vector<const uint64_t*> staged;
    if (lane) {
        size_t total = 0;
        for (uint32_t k = 0; k <= n_layers; ++k) total += h_layer_len[k];
        if (32 * total > c->gkr_in_bytes) {
            ZK_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_gkr_in) (void)hipFree(c->d_gkr_in);
            c->d_gkr_in = nullptr; c->gkr_in_bytes = 0;
            if (hipMalloc(&c->d_gkr_in, 32 * total) != hipSuccess) return ZKHIP_ERR_NOMEM;
            c->gkr_in_bytes = 32 * total;
        }
        staged.resize(n_layers + 1);
        size_t off = 0;
        for (uint32_t k = 0; k <= n_layers; ++k) {
            uint64_t* dst = (uint64_t*)c->d_gkr_in + 4 * off;
            ZK_HIP(c, hipMemcpyAsync(dst, h_layer_ptrs[k], 32 * h_layer_len[k], hipMemcpyDeviceToDevice, c->stream));
            staged[k] = dst;
            off += h_layer_len[k];
        }
        h_layer_ptrs = staged.data();
    }
    // aux layout: w_0 (2) | nine tables of the widest layer | gate weights
    size_t max_w = 0, max_g = 0;
    for (uint32_t l = 0; l < n_layers; ++l) { max_w = std::max(max_w, h_layer_len[l + 1]); max_g = std::max(max_g, cir->layers[l].n_gates); }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t tb = al(32 * max_w);
    const size_t o_w0 = 0, o_tab = al(64), o_wg = o_tab + 9 * tb, o_dot = o_wg + al(32 * max_g), o_ev = o_dot + al(32 * (size_t)zk::MLE_MAX_GRID);
    const size_t o_eqh = o_ev + 256;
    // (+ the device transcript's block behind it, below: outer state, alpha | beta | claim, per-layer sums / w_b / w_c, the proof arena)
    const size_t dt_bytes = al(sizeof(zk::OuterDev)) + 256 + al(32 * ((size_t)n_layers + 1)) + 2 * al(32 * (size_t)n_layers) +
                            al(8 * (size_t)(4 + 64) * ZK_MAX_ROUNDS * n_layers);
    ZK_TRY(c->reserve_aux(o_eqh + al(32 * 2 * (size_t)zk::GKR_EQ_HALVES) + dt_bytes));
    char* aux = (char*)c->d_aux;
    uint64_t* d_w0 = (uint64_t*)(aux + o_w0);
    LayerScratch sc;
    sc.ha0 = (uint64_t*)(aux + o_tab); sc.ha1 = (uint64_t*)(aux + o_tab + tb); sc.hm = (uint64_t*)(aux + o_tab + 2 * tb);
    sc.equ = (uint64_t*)(aux + o_tab + 3 * tb); sc.aa = (uint64_t*)(aux + o_tab + 4 * tb); sc.am = (uint64_t*)(aux + o_tab + 5 * tb);
    sc.t1 = (uint64_t*)(aux + o_tab + 6 * tb); sc.t2 = (uint64_t*)(aux + o_tab + 7 * tb); sc.eqc = (uint64_t*)(aux + o_tab + 8 * tb);
    sc.wg = (uint64_t*)(aux + o_wg);
    sc.dot_partials = (uint64_t*)(aux + o_dot); sc.evals = (uint64_t*)(aux + o_ev); sc.eqh = (uint64_t*)(aux + o_eqh);
    LayerOut out = {h_sums, h_round_polys, h_wb, h_wc, h_challenges, h_n_rounds, h_round_poly_lens, 2 * n_layers};

    // w_0 = circuit_evaluation[0] padded with a zero (protocol.rs:30-33); commit its bytes, draw n_r
    zkhost::Transcript tr;
    ZK_HIP(c, hipMemsetAsync(d_w0, 0, 64, c->stream));
    ZK_HIP(c, hipMemcpyAsync(d_w0, h_layer_ptrs[0], 32, hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_w0, 64, hipMemcpyDeviceToHost, c->stream));   // (pinned: a copy into the caller's pageable buffer is staged)
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_w0, c->pinned_u64(ZK_PIN_RES), 64);
    {
        zkhost::Fr one_canon = zkhost::fr_zero();
        one_canon.l[0] = 1;
        uint8_t be[64];
        for (int e = 0; e < 2; ++e) {
            zkhost::Fr v;
            std::memcpy(v.l, h_w0 + 4 * e, 32);
            const zkhost::Fr cv = zkhost::fr_mul(v, one_canon);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 8; ++j) be[32 * e + 8 * i + j] = (uint8_t)(cv.l[3 - i] >> (56 - 8 * j));
        }
        tr.commit(be, 64);                                           // w_0_mle.to_bytes()
    }
    std::vector<zkhost::Fr> n_r(1, tr.challenge_fr());               // evaluate_n_challenge_into_field(&w_0_mle.n_vars)
    zkhost::Fr claimed;
    {   // w_0_mle.evaluation(&n_r) of the two-entry table [w, 0]: w + n_r (0 - w), on the host (no launch, no wait)
        zkhost::Fr w;
        std::memcpy(w.l, h_w0, 32);
        claimed = zkhost::fr_sub(w, zkhost::fr_mul(n_r[0], w));
    }

    // The outer transcript on the device (default): every layer is enqueued behind the one before, alpha / beta / the claims travel
    // through device memory, the proof is read back ONCE.  ZKHIP_GKR_HOST_TRANSCRIPT=1 keeps the host transcript (one synchronisation
    // per layer; same proof, for A/B runs); the pipelined closing kernels are what feed the device transcript, so ZKHIP_PIPE=0 does too.
    static const bool host_transcript = [] {
        const char* e = std::getenv("ZKHIP_GKR_HOST_TRANSCRIPT");
        const char* p = std::getenv("ZKHIP_PIPE");
        return (e && std::atoi(e) != 0) || (p && std::atoi(p) == 0);
    }();
    if (!host_transcript) {
        const uint32_t stride = 2 * n_layers;
        // device block behind the layer scratch: outer state | next | sums (n_layers + 1) | w_b | w_c | arena
        const size_t o_outer = o_eqh + al(32 * 2 * (size_t)zk::GKR_EQ_HALVES), o_next = o_outer + al(sizeof(zk::OuterDev)), o_sums = o_next + 256;
        const size_t o_wb = o_sums + al(32 * ((size_t)n_layers + 1)), o_wc = o_wb + al(32 * (size_t)n_layers), o_arena = o_wc + al(32 * (size_t)n_layers);
        const size_t arena_bytes = 8 * DeviceTranscript::LAYER_U64 * n_layers;     // (all inside the reservation above: dt_bytes)
        DeviceTranscript dt;
        dt.outer = (zk::OuterDev*)(aux + o_outer);
        dt.next = (uint64_t*)(aux + o_next);
        dt.sums = (uint64_t*)(aux + o_sums);
        dt.wb = (uint64_t*)(aux + o_wb);
        dt.wc = (uint64_t*)(aux + o_wc);
        dt.arena = (uint64_t*)(aux + o_arena);
        // the outer transcript as the host leaves it after n_r: a fresh hasher that has absorbed the 32-byte digest
        {
            zk::Sha256State* hs = (zk::Sha256State*)c->pinned_u64(ZK_PIN_RES);
            std::memset(hs, 0, sizeof(*hs));
            std::memcpy(hs->h, tr.hasher.h, 32);
            const uint32_t fill = (uint32_t)(tr.hasher.len % 64);
            for (uint32_t i = 0; i < fill / 4; ++i)
                hs->buf[i] = ((uint32_t)tr.hasher.buf[4 * i] << 24) | ((uint32_t)tr.hasher.buf[4 * i + 1] << 16) | ((uint32_t)tr.hasher.buf[4 * i + 2] << 8) | tr.hasher.buf[4 * i + 3];
            hs->fill = fill;
            hs->len = tr.hasher.len;
            // state | error | flags in ONE copy.  The flags MUST be cleared per proof: d_aux is uninitialised memory that an earlier
            // context (tokens restart at 1 per context) or a freed table of small integers may have left holding this proof's token --
            // a hasher would then absorb stale items without waiting and the proof would be silently wrong.  Tokens are never 0.
            static_assert(offsetof(zk::OuterDev, error) == sizeof(zk::Sha256State), "state | error | flags: one copy");
            constexpr size_t head_bytes = offsetof(zk::OuterDev, items);
            static_assert(head_bytes <= 8 * (ZK_PIN_PTS - ZK_PIN_RES), "the staging slot holds the head of OuterDev");
            std::memset((char*)hs + sizeof(*hs), 0, head_bytes - sizeof(*hs));   // OuterDev::error, pad_, flag[]
            ZK_HIP(c, hipMemcpyAsync(&dt.outer->state, hs, head_bytes, hipMemcpyHostToDevice, c->stream));
            ZK_HIP(c, hipMemcpyAsync(dt.sums, claimed.l, 32, hipMemcpyHostToDevice, c->stream));     // (pageable source: copied before the call returns)
            if (lane) {     // layer one's gate weights (1 - n_r, n_r): what gkr_gate_weights_kernel computes from n_r passed by value
                const zkhost::Fr w2[2] = {zkhost::fr_sub(zkhost::fr_one(), n_r[0]), n_r[0]};
                ZK_HIP(c, hipMemcpyAsync(sc.wg, w2[0].l, 32 * std::min<size_t>(2, std::max<size_t>(1, cir->layers[0].n_gates)), hipMemcpyHostToDevice, c->stream));
            }
            ZK_HIP(c, hipStreamSynchronize(c->stream));                  // the pinned staging words are reused below
        }
        // ---- the whole proof back in ONE copy: outer state (its error word) | next | sums | w_b | w_c | arena are neighbours on the device
        const size_t pin_bytes = o_arena + arena_bytes - o_outer;
        ZK_TRY(c->reserve_msm_pin(0, pin_bytes));
        char* pin0 = (char*)c->msm_pin[0];
        auto enqueue_chain = [&](bool replayable) -> int {
            bool rows1_done = false;
            for (uint32_t li = 1; li <= n_layers; ++li) {
                const bool has_next = li < n_layers;
                bool next_done = false;
                ZK_TRY(layer_enqueue_device(c, cir->layers[li - 1], li - 1, h_layer_ptrs[li], h_layer_len[li], sc, dt, claimed, n_r[0], stride, rows1_done,
                                            has_next ? &cir->layers[li] : nullptr, has_next ? h_layer_ptrs[li + 1] : nullptr, has_next ? h_layer_len[li + 1] : 0,
                                            &next_done, replayable));
                rows1_done = next_done;
            }
            ZK_HIP(c, hipMemcpyAsync(pin0, aux + o_outer, pin_bytes, hipMemcpyDeviceToHost, c->stream));
            return ZKHIP_OK;
        };
        // the graph of this (circuit, lane): valid while every buffer the chain touches is where it was when it was recorded
        zkhip_ctx::GkrGraph& gg = c->gkr_graph;
        const bool same = lane && gg.exec && gg.cir == cir && gg.aux == c->d_aux && gg.in == c->d_gkr_in && gg.ws == c->d_ws && gg.pin == c->msm_pin[0] &&
                          gg.composed == c->d_composed;
        if (lane && gg.exec && !same) { (void)hipGraphExecDestroy((hipGraphExec_t)gg.exec); gg = zkhip_ctx::GkrGraph(); }
        if (lane && gg.exec) {
            ZK_HIP(c, hipGraphLaunch((hipGraphExec_t)gg.exec, c->stream));
        } else if (lane && gg.warm_cir == cir && gg.warm_aux == c->d_aux && gg.warm_ws == c->d_ws) {
            // the second proof of this circuit on the lane (the first one, launched plainly, made every allocation): record, instantiate, launch
            // Relaxed mode: the recorded region is kernel launches and one copy; the OTHER lanes' threads allocate, free and synchronise at the
            // same time (a capture in the stricter modes was invalidated by them: hipErrorStreamCaptureInvalidated, tools/stress_parity.py)
            hipGraph_t graph = nullptr;
            ZK_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
            const int crc = enqueue_chain(true);
            const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
            if (crc != ZKHIP_OK || ee != hipSuccess || !graph) {
                // not recordable here (whatever the reason): nothing was launched; forget the attempt and launch this proof plainly -- a real
                // error (a shape, a launch failure) comes back from there
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                gg.warm_cir = nullptr;
                ZK_TRY(enqueue_chain(true));
            } else {
                hipGraphExec_t exec = nullptr;
                const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ie != hipSuccess || !exec) { c->last_hip = (int)ie; return ZKHIP_ERR_HIP; }
                gg.exec = exec; gg.cir = cir; gg.aux = c->d_aux; gg.in = c->d_gkr_in; gg.ws = c->d_ws; gg.pin = c->msm_pin[0]; gg.composed = c->d_composed;
                ZK_HIP(c, hipGraphLaunch(exec, c->stream));
            }
        } else {
            ZK_TRY(enqueue_chain(lane));      // (a lane's plain chain is the replayable one too: the same kernels, the same bits)
            if (lane) { gg.warm_cir = cir; gg.warm_aux = c->d_aux; gg.warm_ws = c->d_ws; }
        }
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        // a hasher gave up waiting for a round's items (OuterDev::error = 1 + round).  The hasher is the LAST workgroup of a closing launch and
        // the publisher its workgroup 0: both are resident as long as the grid is at most what the chip holds beside other work -- a
        // residency assumption, hence a status of its own (not a device fault); the outer state is void
        if (((const zk::OuterDev*)pin0)->error != 0) return ZKHIP_ERR_TIMEOUT;
        const char* pin = pin0 + (o_arena - o_outer);
        const char* pin_sums = pin0 + (o_sums - o_outer);
        const char* pin_wb = pin0 + (o_wb - o_outer);
        const char* pin_wc = pin0 + (o_wc - o_outer);
        const uint64_t* h_arena = (const uint64_t*)pin;
        for (uint32_t k = 0; k < n_layers; ++k) {
            const uint32_t nv = 2 * (k + 1);
            const uint64_t* ch = h_arena + (size_t)k * DeviceTranscript::LAYER_U64;
            const uint64_t* rp = ch + 4 * (size_t)ZK_MAX_ROUNDS;
            if (h_challenges) std::memcpy(h_challenges + (size_t)k * stride * 4, ch, 32 * (size_t)nv);
            uint64_t* polys = h_round_polys + (size_t)k * stride * GKR_MONO * 8;
            uint32_t* lens = h_round_poly_lens + (size_t)k * stride;
            for (uint32_t r = 0; r < nv; ++r) {
                lens[r] = (uint32_t)rp[64 * r];
                std::memcpy(polys + (size_t)r * GKR_MONO * 8, &rp[64 * r + 8], GKR_MONO * 64);
            }
            h_n_rounds[k] = nv;
        }
        std::memcpy(h_sums, pin_sums, 32 * (size_t)n_layers);
        std::memcpy(h_wb, pin_wb, 32 * (size_t)n_layers);
        std::memcpy(h_wc, pin_wc, 32 * (size_t)n_layers);
        return ZKHIP_OK;
    }
    zkhost::Fr alpha = zkhost::fr_one(), beta = zkhost::fr_zero();
    std::vector<zkhost::Fr> r_b, r_c;
    // layer one (gkr/src/utils.rs:12-56): the wiring of layer 0 with its gate variable fixed at n_r
    ZK_TRY(layer_prove(c, cir->layers[0], 0, h_layer_ptrs[1], h_layer_len[1], sc, claimed, tr, out, 0, alpha, beta, n_r, r_c, false));
    r_b = n_r;   // layer_prove left (b, c) of layer one in (n_r, r_c)
    for (uint32_t li = 2; li <= n_layers; ++li) {                    // protocol.rs:64-108
        const uint32_t l = li - 1;
        ZK_TRY(layer_prove(c, cir->layers[l], l, h_layer_ptrs[li], h_layer_len[li], sc, claimed, tr, out, li - 1, alpha, beta, r_b, r_c, true));
    }
    return ZKHIP_OK;
}

extern "C" int zkhip_gkr_prove_circuit(zkhip_circuit* cir, const uint64_t* const* h_layer_ptrs, const size_t* h_layer_len, uint64_t* h_sums,
                                       uint32_t* h_n_rounds, uint32_t* h_round_poly_lens, uint64_t* h_round_polys, uint64_t* h_wb,
                                       uint64_t* h_wc, uint64_t* h_w0, uint64_t* h_challenges) {
    if (!cir || !h_layer_ptrs || !h_layer_len || !h_sums || !h_n_rounds || !h_round_poly_lens || !h_round_polys || !h_wb || !h_wc || !h_w0)
        return ZKHIP_ERR_ARG;
    return gkr_prove_circuit_on(cir->c, cir, h_layer_ptrs, h_layer_len, h_sums, h_n_rounds, h_round_poly_lens, h_round_polys, h_wb, h_wc, h_w0, h_challenges);
}

// n_proofs independent proofs of ONE circuit (one GKRProtocol::prove per input: gkr/benches/gkr_benchmark.rs:11-27 proves in a loop) from one
// call.  A proof is a chain of ~130 (depth 8) to ~400 (depth 20) small dependent kernels that keeps one workgroup busy most of the time, so
// throughput comes from independent proofs side by side: the context owns up to GKR_BATCH_LANES child contexts -- a stream, scratch,
// workspace and transcript state each, like a host thread's context -- proof b runs on lane b mod lanes, and the lanes' launch chains are
// enqueued by the context's host pool (the calling thread takes part), because at 3-5 us of host time per launch ONE thread cannot feed
// eight such chains.  Every proof is the one zkhip_gkr_prove_circuit makes, bit for bit.
constexpr uint32_t GKR_BATCH_LANES = 12;
extern "C" int zkhip_gkr_prove_batch(zkhip_circuit* cir, uint32_t n_proofs, uint32_t max_lanes, const uint64_t* const* h_layer_ptrs,
                                     const size_t* h_layer_len, uint64_t* h_sums, uint32_t* h_n_rounds, uint32_t* h_round_poly_lens,
                                     uint64_t* h_round_polys, uint64_t* h_wb, uint64_t* h_wc, uint64_t* h_w0, uint64_t* h_challenges, int* h_status) {
    if (!cir || !h_layer_ptrs || !h_layer_len || !h_sums || !h_n_rounds || !h_round_poly_lens || !h_round_polys || !h_wb || !h_wc || !h_w0)
        return ZKHIP_ERR_ARG;
    if (n_proofs == 0) return ZKHIP_OK;
    zkhip_ctx* c = cir->c;
    ZK_TRY(c->activate());
    const uint32_t nl = cir->n_layers, stride = 2 * nl;
    const uint32_t lanes = std::min<uint32_t>(n_proofs, std::min<uint32_t>(max_lanes ? max_lanes : 8u, GKR_BATCH_LANES));
    while (c->gkr_lanes.size() < lanes) {
        zkhip_ctx* lc = nullptr;
        ZK_TRY(zkhip_ctx_create(&lc, c->device, nullptr));
        // The lanes' streams are spread over the three stream priorities: the runtime keeps a pool of hardware queues PER PRIORITY (four each
        // by default), and two streams that share a hardware queue take turns in it at a cost -- measured on depth-8 proofs: a queue with
        // one chain is 86 % busy, a queue with two 51 % (profiles/r06/e_gkr_batch_kernel_stats.txt).  All lanes do the same work, so
        // what the priorities order is only who goes first.
        hipStream_t s = nullptr;
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { zkhip_ctx_destroy(lc); return ZKHIP_ERR_HIP; }
        static const bool spread = [] { const char* e = std::getenv("ZKHIP_GKR_LANE_PRIO"); return !e || std::atoi(e) != 0; }();
        const int span = least - greatest + 1;                    // (numerically lower = higher priority)
        const int prio = spread && span > 1 ? greatest + (int)(c->gkr_lanes.size() % (size_t)span) : 0;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) { zkhip_ctx_destroy(lc); return ZKHIP_ERR_HIP; }
        lc->stream = s;
        lc->own_stream = true;
        lc->gkr_lane = true;
        lc->gkr_parent = c;
        c->gkr_lanes.push_back(lc);
    }
    // the lanes start behind what the caller's stream holds now (the layer values may still be on their way there)
    if (!c->done_ev && hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
    ZK_HIP(c, hipEventRecord(c->done_ev, c->stream));
    for (uint32_t l = 0; l < lanes; ++l) ZK_HIP(c, hipStreamWaitEvent(c->gkr_lanes[l]->stream, c->done_ev, 0));
    std::vector<int> rcs(n_proofs, ZKHIP_OK);
    const size_t o_sums = 4 * (size_t)nl, o_rounds = nl, o_lens = (size_t)nl * stride, o_polys = (size_t)nl * stride * GKR_MONO * 8, o_ch = (size_t)nl * stride * 4;
    // proofs are handed out one by one: a lane whose hardware queue gets less of the command processor (the lanes' busy shares in a kernel
    // trace range from 35 to 75 %, and differently from process to process) takes fewer of them instead of holding the call up
    std::atomic<uint32_t> next_proof{0};
    auto lane_work = [&](unsigned l) {
        zkhip_ctx* lc = c->gkr_lanes[l];
        for (uint32_t b = next_proof.fetch_add(1); b < n_proofs; b = next_proof.fetch_add(1)) {
            rcs[b] = lc->activate();
            if (rcs[b] != ZKHIP_OK) continue;
            rcs[b] = gkr_prove_circuit_on(lc, cir, h_layer_ptrs + (size_t)b * (nl + 1), h_layer_len, h_sums + b * o_sums, h_n_rounds + b * o_rounds,
                                          h_round_poly_lens + b * o_lens, h_round_polys + b * o_polys, h_wb + b * o_sums, h_wc + b * o_sums, h_w0 + 8 * (size_t)b,
                                          h_challenges ? h_challenges + b * o_ch : nullptr);
            if (rcs[b] == ZKHIP_ERR_HIP) c->last_hip = lc->last_hip;
        }
    };
    ZkHostPool* pool = lanes > 1 ? c->pool() : nullptr;
    if (pool) pool->run(lanes, lane_work);
    else lane_work(0);
    int rc = ZKHIP_OK;
    for (uint32_t b = 0; b < n_proofs; ++b) {
        if (h_status) h_status[b] = rcs[b];
        if (rc == ZKHIP_OK && rcs[b] != ZKHIP_OK) rc = rcs[b];
    }
    return rc;
}

// The layer's linear-size sumcheck tables for a rank of a sharded proof (world = 1, rank = 0: the whole tables).  phase 0: d_out =
// {Ha0, Ha1, Hm, V} rows j * world + rank; phase 1 (after the rounds over b: the challenges lie in the context): d_out = {Aa, V(u) + V,
// Am, V(u) V} rows likewise, V(u) left at ts.evals[0..4).  Every output holds w_len / world entries.  Enqueue only: nothing here waits.
namespace {
struct TablesScratch { uint64_t *wg, *equ, *dot_partials, *evals, *eqh; };
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t tables_scratch_bytes(size_t n_gates, size_t w_len) {
    return al256(32 * std::max<size_t>(n_gates, 1)) + al256(32 * w_len) + al256(32 * (size_t)zk::MLE_MAX_GRID) + 256 + 32 * 2 * (size_t)zk::GKR_EQ_HALVES;
}
TablesScratch carve_tables_scratch(char* base, size_t n_gates, size_t w_len) {
    TablesScratch ts;
    const size_t o_wg = 0, o_equ = o_wg + al256(32 * std::max<size_t>(n_gates, 1)), o_dot = o_equ + al256(32 * w_len),
                 o_ev = o_dot + al256(32 * (size_t)zk::MLE_MAX_GRID), o_eqh = o_ev + 256;
    ts.wg = (uint64_t*)(base + o_wg); ts.equ = (uint64_t*)(base + o_equ); ts.dot_partials = (uint64_t*)(base + o_dot);
    ts.evals = (uint64_t*)(base + o_ev); ts.eqh = (uint64_t*)(base + o_eqh);
    return ts;
}
int layer_tables_enqueue(zkhip_circuit* cir, uint32_t layer, const uint64_t* d_w, size_t w_len, const uint64_t* h_rb, const uint64_t* h_rc,
                         const uint64_t* h_alpha, const uint64_t* h_beta, int phase, uint32_t world, uint32_t rank, uint64_t* const* d_out,
                         const TablesScratch& ts) {
    using namespace zk;
    zkhip_ctx* c = cir->c;
    const LayerDev& ld = cir->layers[layer];
    if (!is_pow2(w_len) || w_len != ld.w_len || w_len < world) return ZKHIP_ERR_SHAPE;
    if (ld.bad_label) return ZKHIP_ERR_INDEX;
    const uint32_t s = log2_exact(w_len);
    const uint32_t n_gate_vars = layer == 0 ? 1u : layer;
    const bool two_points = h_rc != nullptr;
    PtsArg pb = {}, pc = {};
    std::memcpy(pb.v, h_rb, 32 * (size_t)n_gate_vars);
    if (two_points) std::memcpy(pc.v, h_rc, 32 * (size_t)n_gate_vars);
    FrArg av = {}, bv = {};
    std::memcpy(av.v, h_alpha, 32);
    std::memcpy(bv.v, h_beta, 32);
    const uint32_t rows = (uint32_t)(w_len / world);
    const unsigned gw = (unsigned)((rows + MLE_BLOCK - 1) / MLE_BLOCK);
    launch_gate_weights(c, ld.n_gates, n_gate_vars, pb, pc, av, bv, two_points, ts.eqh, ts.wg);
    if (phase == 0) {
        hipLaunchKernelGGL(gkr_gate_rows_kernel, dim3(gw), dim3(MLE_BLOCK), 0, c->stream, ld.csr0, ld.csr0 + w_len + 1, ld.type, ld.in1, ts.wg, d_w,
                           rows, 1u, d_out[0], d_out[1], d_out[2], (const uint64_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr,
                           world, rank, world > 1 ? d_out[3] : (uint64_t*)nullptr);
        ZK_HIP(c, hipGetLastError());
        return ZKHIP_OK;
    }
    const uint64_t* d_ch = zk_composed_challenges_dev(c);       // the s challenges of the rounds over b
    launch_eq_table(c, d_ch, s, ts.eqh, ts.equ, d_w, ts.dot_partials, ts.evals);
    hipLaunchKernelGGL(gkr_gate_rows_kernel, dim3(gw), dim3(MLE_BLOCK), 0, c->stream, ld.csr1, ld.csr1 + w_len + 1, ld.type, ld.in0, ts.wg, ts.equ,
                       rows, 2u, d_out[0], (uint64_t*)nullptr, d_out[2], d_w, (const uint64_t*)ts.evals, d_out[1], d_out[3], world, rank, (uint64_t*)nullptr);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}
}  // namespace
extern "C" int zkhip_gkr_layer_tables_sharded(zkhip_circuit* cir, uint32_t layer, const uint64_t* d_w, size_t w_len, const uint64_t* h_rb,
                                              const uint64_t* h_rc, const uint64_t* h_alpha, const uint64_t* h_beta, int phase,
                                              uint32_t world, uint32_t rank, uint64_t* const* d_out, uint64_t* h_wu) {
    if (!cir || !d_w || !h_rb || !h_alpha || !h_beta || !d_out || phase < 0 || phase > 1 || layer >= cir->n_layers) return ZKHIP_ERR_ARG;
    if (phase == 1 && !h_wu) return ZKHIP_ERR_ARG;
    if (world == 0 || (world & (world - 1)) || rank >= world) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = cir->c;
    const LayerDev& ld = cir->layers[layer];
    if (!is_pow2(w_len) || w_len != ld.w_len || w_len < world) return ZKHIP_ERR_SHAPE;
    if (ld.bad_label) return ZKHIP_ERR_INDEX;
    ZK_TRY(c->activate());
    ZK_TRY(c->reserve_aux(tables_scratch_bytes(ld.n_gates, w_len)));
    const TablesScratch ts = carve_tables_scratch((char*)c->d_aux, ld.n_gates, w_len);
    ZK_TRY(layer_tables_enqueue(cir, layer, d_w, w_len, h_rb, h_rc, h_alpha, h_beta, phase, world, rank, d_out, ts));
    if (phase == 0) return ZKHIP_OK;
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), ts.evals, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_wu, c->pinned_u64(ZK_PIN_RES), 32);
    return ZKHIP_OK;
}
extern "C" int zkhip_gkr_layer_tables(zkhip_circuit* cir, uint32_t layer, const uint64_t* d_w, size_t w_len, const uint64_t* h_rb,
                                      const uint64_t* h_rc, const uint64_t* h_alpha, const uint64_t* h_beta, int phase,
                                      uint64_t* const* d_out, uint64_t* h_wu) {
    return zkhip_gkr_layer_tables_sharded(cir, layer, d_w, w_len, h_rb, h_rc, h_alpha, h_beta, phase, 1, 0, d_out, h_wu);
}

// GKRProtocol::prove (protocol.rs:21-117) with every layer's sumcheck tables sharded over the ranks of `comm` (include/zkhip.h): per layer
// this rank's rows of the tables of the rounds over b -> a composed session on them (two rounds per exchange) -> the rows of the tables of
// the rounds over c (b bound to the challenges the first session left in the context) -> a session that continues the first one's
// transcript -> w_b, w_c -> the outer transcript on the host (the ONE synchronisation of the layer).  Exchanges are stream-ordered.
extern "C" int zkhip_gkr_prove_sharded(zkhip_circuit* cir, zkhip_comm* comm, const uint64_t* const* h_layer_ptrs, const size_t* h_layer_len,
                                       int use_stages, uint64_t* h_sums, uint32_t* h_n_rounds, uint32_t* h_round_poly_lens,
                                       uint64_t* h_round_polys, uint64_t* h_wb, uint64_t* h_wc, uint64_t* h_w0, uint64_t* h_challenges,
                                       uint32_t* exchanges) {
    if (!cir || !comm || !h_layer_ptrs || !h_layer_len || !h_sums || !h_n_rounds || !h_round_poly_lens || !h_round_polys || !h_wb || !h_wc || !h_w0)
        return ZKHIP_ERR_ARG;
    zkhip_ctx* c = cir->c;
    if (comm->c != c) return ZKHIP_ERR_ARG;                       // the circuit and the communicator live on one context (one GPU, one stream)
    const uint32_t n_layers = cir->n_layers, world = comm->world_, rank = comm->rank_;
    if (h_layer_len[0] != 1) return ZKHIP_ERR_SHAPE;
    for (uint32_t k = 1; k <= n_layers; ++k)
        if (!is_pow2(h_layer_len[k])) return ZKHIP_ERR_SHAPE;     // Multilinear::new (evaluation_form.rs:16-20)
    ZK_TRY(c->activate());
    // aux layout: w_0 (2 entries) | the table builders' scratch | 8 tables of this rank's rows of the widest layer
    size_t max_rows = 1, max_scratch = 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        const size_t w_len = h_layer_len[l + 1];
        max_rows = std::max(max_rows, w_len >= 2 * (size_t)world ? w_len / world : w_len);
        max_scratch = std::max(max_scratch, tables_scratch_bytes(cir->layers[l].n_gates, w_len));
    }
    const size_t tb = al256(32 * max_rows), o_scr = 256, o_tab = o_scr + al256(max_scratch);
    ZK_TRY(c->reserve_aux(o_tab + 8 * tb));
    char* aux = (char*)c->d_aux;
    uint64_t* d_w0 = (uint64_t*)aux;
    uint64_t* tab[8];
    for (int q = 0; q < 8; ++q) tab[q] = (uint64_t*)(aux + o_tab + (size_t)q * tb);
    zkhip_comm* solo = comm->solo_comm();                         // narrow layers run whole on every rank: a one-rank exchange (no transport)
    if (!solo) return ZKHIP_ERR_NOMEM;
    uint32_t n_ex = 0;
    const uint32_t stride = 2 * n_layers;

    // w_0 = circuit_evaluation[0] padded with a zero (protocol.rs:30-33); commit its bytes, draw n_r
    zkhost::Transcript tr;
    ZK_HIP(c, hipMemsetAsync(d_w0, 0, 64, c->stream));
    ZK_HIP(c, hipMemcpyAsync(d_w0, h_layer_ptrs[0], 32, hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_w0, 64, hipMemcpyDeviceToHost, c->stream));   // (pinned: a copy into the caller's pageable buffer is staged)
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_w0, c->pinned_u64(ZK_PIN_RES), 64);
    {
        zkhost::Fr one_canon = zkhost::fr_zero();
        one_canon.l[0] = 1;
        uint8_t be[64];
        for (int e = 0; e < 2; ++e) {
            zkhost::Fr v;
            std::memcpy(v.l, h_w0 + 4 * e, 32);
            const zkhost::Fr cv = zkhost::fr_mul(v, one_canon);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 8; ++j) be[32 * e + 8 * i + j] = (uint8_t)(cv.l[3 - i] >> (56 - 8 * j));
        }
        tr.commit(be, 64);                                           // w_0_mle.to_bytes()
    }
    std::vector<zkhost::Fr> r_b(1, tr.challenge_fr()), r_c;          // evaluate_n_challenge_into_field(&w_0_mle.n_vars)
    zkhost::Fr claimed;
    {   // w_0_mle.evaluation(&n_r) of the two-entry table [w, 0]: w + n_r (0 - w), on the host (no launch, no wait)
        zkhost::Fr w;
        std::memcpy(w.l, h_w0, 32);
        claimed = zkhost::fr_sub(w, zkhost::fr_mul(r_b[0], w));
    }
    zkhost::Fr alpha = zkhost::fr_one(), beta = zkhost::fr_zero();

    // A rank that fails must not hang its peers (shard_protocol.hpp): it enters every exchange the healthy ranks still make, with poison
    // records, up to the end of the first layer that has any; the healthy ranks read the sticky flag where they wait for the GPU anyway
    // (the end of each layer) and return ZKHIP_ERR_PEER there.  sessions_left: how many of layer li_from's two sessions are still to come.
    auto walk_failed = [&](uint32_t li_from, int sessions_left, int fail_rc) {
        for (uint32_t li = li_from; li <= n_layers; ++li, sessions_left = 2) {
            const size_t w_len = h_layer_len[li];
            if (w_len < 2 * (size_t)world || world == 1) continue;              // a narrow layer runs whole on every rank: no exchange
            const uint32_t sizes[2] = {2, 2};
            const bool stages = use_stages >= 0 ? use_stages != 0 : true;
            for (int sess = 2 - sessions_left; sess < 2; ++sess) {
                zkshard::HipMcEngine e{nullptr, comm};
                if (e.shape(sizes, 2, sess == 0 ? 1u : 0u, w_len / world) != ZKHIP_OK) return;
                uint32_t ex = 0;
                (void)zkshard::composed_prove(e, *comm, stages, &ex, fail_rc);
                n_ex += ex;
            }
            break;                                                                // the peers look at the flag behind this layer
        }
        (void)c->wait_stream();
        (void)comm->take_peer_failure();
    };
    int rc = ZKHIP_OK;
    uint32_t fail_layer = 0;     // where this rank failed: the layer, and how many of its sessions the peers still run
    int fail_sessions_left = 0;
#define GKR_FAIL(sessions_left) { fail_layer = li; fail_sessions_left = (sessions_left); break; }
    for (uint32_t li = 1; li <= n_layers && rc == ZKHIP_OK; ++li) {   // layer one (gkr/src/utils.rs:12-56), then protocol.rs:64-108
        const uint32_t l = li - 1, k = li - 1;
        const uint64_t* V = h_layer_ptrs[li];
        const size_t w_len = h_layer_len[li];
        const LayerDev& ld = cir->layers[l];
        const uint32_t s = log2_exact(w_len), n_gate_vars = l == 0 ? 1u : l;
        const bool two_points = li > 1;
        if (s != l + 1 || w_len != ld.w_len || 2 * s > stride) { rc = ZKHIP_ERR_SHAPE; break; }
        if (r_b.size() != n_gate_vars || (two_points && r_c.size() != n_gate_vars)) { rc = ZKHIP_ERR_SHAPE; break; }
        const uint32_t w = w_len >= 2 * (size_t)world ? world : 1;    // narrow layers: every rank proves them whole
        const uint32_t rk = w > 1 ? rank : 0;
        zkhip_comm* cm = w > 1 ? comm : solo;
        const bool stages = use_stages >= 0 ? use_stages != 0 : w > 1;
        const size_t rows = w_len / w;
        const TablesScratch ts = carve_tables_scratch(aux + o_scr, ld.n_gates, w_len);
        const uint32_t sizes[2] = {2, 2};
        // ---- rounds over b on this rank's rows of [Ha0, V] + Ha1, [Hm, V]
        if ((rc = layer_tables_enqueue(cir, l, V, w_len, r_b[0].l, two_points ? r_c[0].l : nullptr, alpha.l, beta.l, 0, w, rk, tab, ts)) != ZKHIP_OK) GKR_FAIL(2)
        const uint64_t* v_sh = w > 1 ? tab[3] : V;
        {
            const uint64_t* tables[4] = {tab[0], v_sh, tab[2], v_sh};
            const uint64_t* lin[2] = {tab[1], nullptr};
            zkhip_mc_state* st = nullptr;
            if ((rc = zkhip_mc_begin_ex(c, tables, sizes, 2, lin, rows, w, 1, claimed.l, 0, 0, &st)) != ZKHIP_OK) GKR_FAIL(2)
            uint32_t ex = 0;
            rc = zkhip_mc_prove_sharded(st, cm, stages ? 1 : 0, nullptr, nullptr, nullptr, &ex);   // recorded on the device; releases the session
            n_ex += ex;
            if (rc != ZKHIP_OK) GKR_FAIL(1)                                     // (its own remaining exchanges were walked inside)
        }
        // ---- rounds over c, b at u = the challenges just recorded: rows of [Aa, V(u) + V], [Am, V(u) V]; V(u) = w_b stays on the device
        if ((rc = layer_tables_enqueue(cir, l, V, w_len, r_b[0].l, two_points ? r_c[0].l : nullptr, alpha.l, beta.l, 1, w, rk, tab + 4, ts)) != ZKHIP_OK) GKR_FAIL(1)
        uint64_t* polys = h_round_polys + (size_t)k * stride * GKR_MONO * 8;
        uint32_t* lens = h_round_poly_lens + (size_t)k * stride;
        std::vector<uint64_t> challenges(4 * (size_t)(2 * s));
        {
            const uint64_t* tables[4] = {tab[4], tab[5], tab[6], tab[7]};
            zkhip_mc_state* st = nullptr;
            if ((rc = zkhip_mc_begin_ex(c, tables, sizes, 2, nullptr, rows, w, 1, nullptr, 1, s, &st)) != ZKHIP_OK) GKR_FAIL(1)
            zkshard::HipMcEngine e{st, cm};
            if ((rc = e.shape_of_session()) != ZKHIP_OK) { zkhip_mc_abort(st); GKR_FAIL(1) }
            uint32_t ex = 0;
            rc = zkshard::composed_prove(e, *cm, stages, &ex);
            n_ex += ex;
            if (rc != ZKHIP_OK) { zkhip_mc_abort(st); GKR_FAIL(0) }
            // w_c = V(r_c), r_c = the second half of the challenges: enqueued behind the rounds, read back with them
            zk::launch_eq_table(c, zk_composed_challenges_dev(c) + 4 * (size_t)s, s, ts.eqh, ts.equ, V, ts.dot_partials, ts.evals + 4);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), ts.evals, 64, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
                zkhip_mc_abort(st);
                rc = ZKHIP_ERR_HIP;
                GKR_FAIL(0)
            }
            if ((rc = zkhip_mc_finish(st, lens, polys, challenges.data())) != ZKHIP_OK) GKR_FAIL(0)   // all 2 s rounds; synchronises the stream
            if (comm->take_peer_failure()) { rc = ZKHIP_ERR_PEER; break; }     // a peer's poison record in one of this layer's exchanges: every healthy rank stops here
        }
        zkhost::Fr eval_wb, eval_wc;
        std::memcpy(eval_wb.l, c->pinned_u64(ZK_PIN_RES), 32);
        std::memcpy(eval_wc.l, c->pinned_u64(ZK_PIN_RES) + 4, 32);
        if (h_challenges) std::memcpy(h_challenges + (size_t)k * stride * 4, challenges.data(), 32 * (size_t)(2 * s));
        std::memcpy(h_sums + 4 * (size_t)k, claimed.l, 32);
        h_n_rounds[k] = 2 * s;
        absorb_proof(tr, polys, lens, 2 * s);                          // transcript.commit(&sumcheck_proof.to_bytes())
        r_b.assign(s, zkhost::fr_zero());                              // challenges.split_at(len / 2)
        std::memcpy(r_b.data(), challenges.data(), 32 * (size_t)s);
        r_c.assign(s, zkhost::fr_zero());
        std::memcpy(r_c.data(), challenges.data() + 4 * (size_t)s, 32 * (size_t)s);
        std::memcpy(h_wb + 4 * (size_t)k, eval_wb.l, 32);
        std::memcpy(h_wc + 4 * (size_t)k, eval_wc.l, 32);
        alpha = tr.challenge_fr();
        beta = tr.challenge_fr();
        claimed = zkhost::fr_add(zkhost::fr_mul(alpha, eval_wb), zkhost::fr_mul(beta, eval_wc));
    }
#undef GKR_FAIL
    if (fail_layer) walk_failed(fail_sessions_left ? fail_layer : fail_layer + 1, fail_sessions_left ? fail_sessions_left : 2, rc);
    if (exchanges) *exchanges = n_ex;
    return rc;
}

// one-shot form: the circuit is grouped and uploaded for this proof only
extern "C" int zkhip_gkr_prove(zkhip_ctx* c, uint32_t n_layers, const size_t* h_n_gates, const uint8_t* h_gate_type,
                               const uint32_t* h_in0, const uint32_t* h_in1, const uint64_t* const* h_layer_ptrs,
                               const size_t* h_layer_len, uint64_t* h_sums, uint32_t* h_n_rounds, uint32_t* h_round_poly_lens,
                               uint64_t* h_round_polys, uint64_t* h_wb, uint64_t* h_wc, uint64_t* h_w0, uint64_t* h_challenges) {
    if (!c || !h_n_gates || !h_gate_type || !h_in0 || !h_in1 || !h_layer_ptrs || !h_layer_len || !h_sums || !h_n_rounds ||
        !h_round_poly_lens || !h_round_polys || !h_wb || !h_wc || !h_w0)
        return ZKHIP_ERR_ARG;
    if (n_layers < 1 || 2 * n_layers > (uint32_t)ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    if (h_layer_len[0] != 1) return ZKHIP_ERR_SHAPE;
    for (uint32_t k = 1; k <= n_layers; ++k)
        if (!is_pow2(h_layer_len[k]) || h_layer_len[k] != ((size_t)2 << (k - 1))) return ZKHIP_ERR_SHAPE;   // shapes before labels, as the reference panics
    zkhip_circuit* cir = nullptr;
    ZK_TRY(zkhip_circuit_create(c, n_layers, h_n_gates, h_gate_type, h_in0, h_in1, &cir));
    const int rc = zkhip_gkr_prove_circuit(cir, h_layer_ptrs, h_layer_len, h_sums, h_n_rounds, h_round_poly_lens, h_round_polys, h_wb, h_wc,
                                           h_w0, h_challenges);
    zkhip_circuit_destroy(cir);
    return rc;
}
