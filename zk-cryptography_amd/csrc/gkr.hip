// gkr.hip -- GKRProtocol::prove as one C-ABI call: the host orchestration of the reference's prover over the
// device-resident tables.  Every table operation below is one of libzkhip's own entry points (HIP kernels);
// the outer Fiat-Shamir transcript absorbs a few hundred bytes per layer and runs on the host.
// gfx950 only.  No CPU fallback: the tables never leave HBM.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "host_util.hpp"
#include "host_fr.hpp"
#include "mle_kernels.hpp"

// ---- wiring tables folded at the gate variables, without the dense table ------------------------------------------
// The reference builds add_i / mul_i as dense 0/1 tables over (a, b, c) -- 2^(3l+2) entries, one 1 per gate
// (circuit.rs:59-97) -- folds them at r_b and at r_c over the l gate variables and combines the two with alpha, beta
// (protocol.rs:67-87).  A fold over `a` of a table that is 1 at (g, b_g, c_g) and 0 elsewhere is
//     T'[b, c] = sum over the gates g with inputs (b, c) of eq_g(r),   eq_g(r) = prod_j (bit_j(g) ? r_j : 1 - r_j),
// the same field element (the arithmetic is exact), so the (b, c) tables are written directly: one lane per distinct
// (type, b, c), summing alpha eq_g(r_b) + beta eq_g(r_c) over its gates (the host groups the gates).  2^(2l+2) entries
// of output instead of 2^(3l+2) of input: the 2^23-entry tables of an 8-layer circuit never exist, and deeper circuits
// than the dense representation can hold become provable.
namespace zk {
static __global__ __launch_bounds__(MLE_BLOCK) void gkr_wiring_kernel(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_out,
                                                               const uint32_t* __restrict__ gate_ids, uint32_t n_segs, uint32_t n_gate_vars,
                                                               PtsArg r_b, PtsArg r_c, FrArg alpha_v, FrArg beta_v, uint32_t two_points,
                                                               uint64_t* __restrict__ add_bc, uint64_t* __restrict__ mul_bc) {
    const uint32_t sidx = blockIdx.x * MLE_BLOCK + threadIdx.x;
    if (sidx >= n_segs) return;
    const Fr one = Fr::one();
    const Fr alpha = fr_from_arg(alpha_v), beta = fr_from_arg(beta_v);
    Fr acc = Fr::zero();
    for (uint32_t q = seg_off[sidx]; q < seg_off[sidx + 1]; ++q) {
        const uint32_t g = gate_ids[q];
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
        acc = acc + (two_points ? alpha * eb + beta * ec : eb);
    }
    const uint32_t o = seg_out[sidx];
    store_fr((o >> 31) ? mul_bc : add_bc, o & 0x7fffffffu, acc);
}
}  // namespace zk

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
    zkhost::Fr one_canon = zkhost::fr_zero();
    one_canon.l[0] = 1;
    for (uint32_t r = 0; r < n_rounds; ++r) {
        for (uint32_t m = 0; m < lens[r]; ++m) {
            for (int part = 0; part < 2; ++part) {
                zkhost::Fr v;
                std::memcpy(v.l, polys + ((size_t)r * GKR_MONO + m) * 8 + 4 * part, 32);
                const zkhost::Fr c = zkhost::fr_mul(v, one_canon);   // out of Montgomery form
                uint8_t be[32];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 8; ++j) be[8 * i + j] = (uint8_t)(c.l[3 - i] >> (56 - 8 * j));
                tr.commit(be, 32);
            }
        }
    }
}

// The shared tail of generate_layer_one_prove_sumcheck (gkr/src/utils.rs:27-55) and of the loop body of
// GKRProtocol::prove (protocol.rs:78-107).  d_add / d_mul: the wiring tables reduced to (b, c), n = w_len^2 entries.
// d_sum / d_prod: scratch of n entries each.
int layer_sumcheck(zkhip_ctx* c, const uint64_t* d_add, const uint64_t* d_mul, const uint64_t* d_w, size_t w_len, uint64_t* d_sum,
                   uint64_t* d_prod, zkhost::Fr& claimed, zkhost::Transcript& tr, const LayerOut& out, uint32_t k, zkhost::Fr& alpha,
                   zkhost::Fr& beta, std::vector<zkhost::Fr>& r_b, std::vector<zkhost::Fr>& r_c) {
    const size_t n = w_len * w_len;
    const uint32_t nv = log2_exact(n);
    if (nv > out.stride) return ZKHIP_ERR_SHAPE;
    ZK_TRY(zkhip_mle_add_distinct(c, d_w, w_len, d_w, w_len, d_sum));     // wb.add_distinct(&wc)
    ZK_TRY(zkhip_mle_mul_distinct(c, d_w, w_len, d_w, w_len, d_prod));    // wb.mul_distinct(&wc)
    const uint64_t* tables[4] = {d_add, d_sum, d_mul, d_prod};            // [add, wb + wc], [mul, wb * wc]
    const uint32_t sizes[2] = {2, 2};
    std::vector<uint64_t> challenges(4 * (size_t)nv);
    uint64_t* polys = out.round_polys + (size_t)k * out.stride * GKR_MONO * 8;
    uint32_t* lens = out.lens + (size_t)k * out.stride;
    ZK_TRY(zkhip_multi_composed_prove(c, tables, sizes, 2, n, claimed.l, 1, lens, polys, challenges.data()));
    if (out.challenges) std::memcpy(out.challenges + (size_t)k * out.stride * 4, challenges.data(), 32 * (size_t)nv);
    std::memcpy(out.sums + 4 * (size_t)k, claimed.l, 32);
    out.n_rounds[k] = nv;
    absorb_proof(tr, polys, lens, nv);                                     // transcript.commit(&sumcheck_proof.to_bytes())
    const uint32_t half = nv / 2;                                          // challenges.split_at(len / 2)
    r_b.assign(half, zkhost::fr_zero());
    r_c.assign(nv - half, zkhost::fr_zero());
    std::memcpy(r_b.data(), challenges.data(), 32 * (size_t)half);
    std::memcpy(r_c.data(), challenges.data() + 4 * (size_t)half, 32 * (size_t)(nv - half));
    zkhost::Fr eval_wb, eval_wc;
    ZK_TRY(zkhip_mle_evaluation(c, d_w, w_len, r_b.empty() ? nullptr : r_b[0].l, r_b.size(), eval_wb.l));
    ZK_TRY(zkhip_mle_evaluation(c, d_w, w_len, r_c.empty() ? nullptr : r_c[0].l, r_c.size(), eval_wc.l));
    std::memcpy(out.wb + 4 * (size_t)k, eval_wb.l, 32);
    std::memcpy(out.wc + 4 * (size_t)k, eval_wc.l, 32);
    alpha = tr.challenge_fr();
    beta = tr.challenge_fr();
    claimed = zkhost::fr_add(zkhost::fr_mul(alpha, eval_wb), zkhost::fr_mul(beta, eval_wc));
    return ZKHIP_OK;
}

// add_bc / mul_bc (bc entries each) for layer l's gates, folded at r_b (and r_c, weighted alpha / beta, when given)
int wiring_tables(zkhip_ctx* c, const uint8_t* gate_type, const uint32_t* in0, const uint32_t* in1, size_t n_gates, uint32_t l,
                  const std::vector<zkhost::Fr>& r_b, const std::vector<zkhost::Fr>* r_c, const zkhost::Fr& alpha, const zkhost::Fr& beta,
                  size_t bc, uint64_t* d_add, uint64_t* d_mul) {
    const uint32_t shift = l + 1;                       // bits of b and of c
    const uint32_t n_gate_vars = l == 0 ? 1u : l;       // binary_string(a, layer_index) has at least one bit (circuit/src/utils.rs:27-33)
    if (r_b.size() != n_gate_vars || (r_c && r_c->size() != n_gate_vars) || 2 * shift > 30) return ZKHIP_ERR_SHAPE;
    if (bc != ((size_t)1 << (2 * shift))) return ZKHIP_ERR_SHAPE;
    std::vector<uint64_t> keys(n_gates);                // (type, b, c, gate)
    for (size_t g = 0; g < n_gates; ++g) {
        if (in0[g] >> shift || in1[g] >> shift || (g >> n_gate_vars)) return ZKHIP_ERR_INDEX;   // add_evaluations[gate_decimal] out of bounds
        const uint64_t o = ((uint64_t)(gate_type[g] ? 1 : 0) << 31) | ((uint64_t)in0[g] << shift) | in1[g];
        keys[g] = (o << 32) | (uint32_t)g;
    }
    std::sort(keys.begin(), keys.end());
    std::vector<uint32_t> host;                          // [seg_off (n_segs + 1)] [seg_out (n_segs)] [gate_ids (n_gates)]
    std::vector<uint32_t> seg_off, seg_out, ids(n_gates);
    for (size_t q = 0; q < n_gates; ++q) {
        ids[q] = (uint32_t)keys[q];
        if (q == 0 || (keys[q] >> 32) != (keys[q - 1] >> 32)) { seg_off.push_back((uint32_t)q); seg_out.push_back((uint32_t)(keys[q] >> 32)); }
    }
    const uint32_t n_segs = (uint32_t)seg_out.size();
    seg_off.push_back((uint32_t)n_gates);
    host.insert(host.end(), seg_off.begin(), seg_off.end());
    host.insert(host.end(), seg_out.begin(), seg_out.end());
    host.insert(host.end(), ids.begin(), ids.end());
    ZK_HIP(c, hipMemsetAsync(d_add, 0, 32 * bc, c->stream));   // F::zero() is all-zero limbs
    ZK_HIP(c, hipMemsetAsync(d_mul, 0, 32 * bc, c->stream));
    if (n_segs == 0) return ZKHIP_OK;
    ZK_TRY(c->reserve_ws(4 * host.size() + 256));
    ZK_HIP(c, hipMemcpyAsync(c->d_ws, host.data(), 4 * host.size(), hipMemcpyHostToDevice, c->stream));
    zk::PtsArg pb = {}, pc = {};
    std::memcpy(pb.v, r_b[0].l, 32 * r_b.size());
    if (r_c) std::memcpy(pc.v, (*r_c)[0].l, 32 * r_c->size());
    zk::FrArg a = {}, b = {};
    std::memcpy(a.v, alpha.l, 32);
    std::memcpy(b.v, beta.l, 32);
    const uint32_t* d = (const uint32_t*)c->d_ws;
    hipLaunchKernelGGL(zk::gkr_wiring_kernel, dim3((n_segs + zk::MLE_BLOCK - 1) / zk::MLE_BLOCK), dim3(zk::MLE_BLOCK), 0, c->stream, d, d + n_segs + 1,
                       d + 2 * n_segs + 1, n_segs, n_gate_vars, pb, pc, a, b, r_c ? 1u : 0u, d_add, d_mul);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipStreamSynchronize(c->stream));   // `host` is a temporary and the staging lives in the shared workspace
    return ZKHIP_OK;
}

}  // namespace

extern "C" int zkhip_gkr_prove(zkhip_ctx* c, uint32_t n_layers, const size_t* h_n_gates, const uint8_t* h_gate_type,
                               const uint32_t* h_in0, const uint32_t* h_in1, const uint64_t* const* h_layer_ptrs,
                               const size_t* h_layer_len, uint64_t* h_sums, uint32_t* h_n_rounds, uint32_t* h_round_poly_lens,
                               uint64_t* h_round_polys, uint64_t* h_wb, uint64_t* h_wc, uint64_t* h_w0, uint64_t* h_challenges) {
    if (!c || !h_n_gates || !h_gate_type || !h_in0 || !h_in1 || !h_layer_ptrs || !h_layer_len || !h_sums || !h_n_rounds ||
        !h_round_poly_lens || !h_round_polys || !h_wb || !h_wc || !h_w0)
        return ZKHIP_ERR_ARG;
    if (n_layers < 1 || n_layers > 14) return ZKHIP_ERR_SHAPE;   // (b, c) tables of 2^(2 n_layers) entries
    if (h_layer_len[0] != 1) return ZKHIP_ERR_SHAPE;              // w_0 = [output.., 0] must have 2^k entries; the wiring of layer 0 has one gate bit
    for (uint32_t k = 1; k <= n_layers; ++k)
        if (!is_pow2(h_layer_len[k])) return ZKHIP_ERR_SHAPE;  // Multilinear::new (evaluation_form.rs:16-20)
    ZK_TRY(c->activate());
    // aux layout: w_0 (2) | the two wiring tables over (b, c) | wb + wc, wb * wc  (all of the layer's (b, c) size)
    size_t max_bc = 0;
    for (uint32_t l = 0; l < n_layers; ++l) max_bc = std::max(max_bc, h_layer_len[l + 1] * h_layer_len[l + 1]);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_w0 = 0, o_add = al(64), o_mul = o_add + al(32 * max_bc), o_sum = o_mul + al(32 * max_bc), o_prod = o_sum + al(32 * max_bc);
    ZK_TRY(c->reserve_aux(o_prod + al(32 * max_bc)));
    char* aux = (char*)c->d_aux;
    uint64_t* d_w0 = (uint64_t*)(aux + o_w0);
    uint64_t* d_add = (uint64_t*)(aux + o_add);
    uint64_t* d_mul = (uint64_t*)(aux + o_mul);
    uint64_t* d_sum = (uint64_t*)(aux + o_sum);
    uint64_t* d_prod = (uint64_t*)(aux + o_prod);
    LayerOut out = {h_sums, h_round_polys, h_wb, h_wc, h_challenges, h_n_rounds, h_round_poly_lens, 2 * n_layers};

    // w_0 = circuit_evaluation[0] padded with a zero (protocol.rs:30-33); commit its bytes, draw n_r
    zkhost::Transcript tr;
    ZK_HIP(c, hipMemsetAsync(d_w0, 0, 64, c->stream));
    ZK_HIP(c, hipMemcpyAsync(d_w0, h_layer_ptrs[0], 32, hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipMemcpyAsync(h_w0, d_w0, 64, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
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
    ZK_TRY(zkhip_mle_evaluation(c, d_w0, 2, n_r[0].l, 1, claimed.l));

    zkhost::Fr alpha = zkhost::fr_one(), beta = zkhost::fr_zero();
    std::vector<zkhost::Fr> r_b, r_c;
    size_t g_off = 0;
    {   // layer one (gkr/src/utils.rs:12-56): the wiring of layer 0 with its gate variable fixed at n_r
        const size_t bc = h_layer_len[1] * h_layer_len[1];
        ZK_TRY(wiring_tables(c, h_gate_type, h_in0, h_in1, h_n_gates[0], 0, n_r, nullptr, alpha, beta, bc, d_add, d_mul));
        ZK_TRY(layer_sumcheck(c, d_add, d_mul, h_layer_ptrs[1], h_layer_len[1], d_sum, d_prod, claimed, tr, out, 0, alpha, beta, r_b, r_c));
        g_off += h_n_gates[0];
    }
    for (uint32_t li = 2; li <= n_layers; ++li) {                    // protocol.rs:64-108
        const uint32_t l = li - 1;
        const size_t bc = h_layer_len[li] * h_layer_len[li];
        // alpha * add(r_b, b, c) + beta * add(r_c, b, c); the same for mul  (:67-87)
        ZK_TRY(wiring_tables(c, h_gate_type + g_off, h_in0 + g_off, h_in1 + g_off, h_n_gates[l], l, r_b, &r_c, alpha, beta, bc, d_add, d_mul));
        ZK_TRY(layer_sumcheck(c, d_add, d_mul, h_layer_ptrs[li], h_layer_len[li], d_sum, d_prod, claimed, tr, out, li - 1, alpha, beta, r_b, r_c));
        g_off += h_n_gates[l];
    }
    return ZKHIP_OK;
}
