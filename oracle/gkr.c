/* gkr.c -- CPU restatement of the GKR prover / verifier and the layered circuit (TEST INFRASTRUCTURE ONLY).
 *
 * Follows, function by function:
 *   circuit/src/circuit.rs:31-57      Circuit::evaluation
 *   circuit/src/circuit.rs:59-97      Circuit::add_mult_mle  (dense 0/1 wiring tables)
 *   circuit/src/utils.rs:1-34         size_of_mle_n_var_at_each_layer, transform_label_to_binary_and_to_decimal
 *   gkr/src/utils.rs:8-56             w_mle, generate_layer_one_prove_sumcheck
 *   gkr/src/utils.rs:58-98            generate_layer_one_verify_sumcheck
 *   gkr/src/protocol.rs:21-117        GKRProtocol::prove
 *   gkr/src/protocol.rs:119-196       GKRProtocol::verify
 * Pinned by the reference's own tests: circuit outputs (circuit.rs:140-260, protocol.rs:280 -> 224), wiring-table
 * positions (circuit.rs:263-518) and prove -> verify == true (protocol.rs:209-286).  Proof BYTES are parity-unpinned
 * (no reference test fixes them); they rest on the KAT-pinned pieces they are composed of.
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

/* circuit/src/utils.rs:1-10 */
size_t ora_gkr_mle_size(size_t layer_index) {
    if (layer_index == 0) return (size_t)1 << 3;
    return (size_t)1 << (layer_index + 2 * (layer_index + 1));
}

/* circuit/src/utils.rs:12-25: the binary strings of a (layer_index bits, at least 1), b and c (layer_index + 1 bits
 * each) concatenated.  For labels that fit their width this is a << 2(l+1) | b << (l+1) | c. */
static size_t wiring_index(size_t layer_index, size_t a, size_t b, size_t c) {
    return (a << (2 * (layer_index + 1))) | (b << (layer_index + 1)) | c;
}

static size_t gate_offset(const size_t *n_gates, size_t layer) {
    size_t off = 0;
    for (size_t l = 0; l < layer; ++l) off += n_gates[l];
    return off;
}

/* Circuit::evaluation (circuit.rs:31-57): out = layer 0 (the output) first ... the input last, concatenated;
 * layer_len[0..n_layers] their lengths.  Returns -1 on an out-of-range gate input (index panic). */
int ora_circuit_evaluation(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                           const uint32_t *in1, const fr_t *input, size_t n_input, fr_t *out, size_t *layer_len) {
    size_t total = n_input;
    for (size_t l = 0; l < n_layers; ++l) total += n_gates[l];
    /* fill from the back: the input is the last block */
    size_t pos = total - n_input;
    memcpy(out + pos, input, n_input * sizeof(fr_t));
    layer_len[n_layers] = n_input;
    const fr_t *cur = out + pos;
    size_t cur_len = n_input;
    for (size_t l = n_layers; l-- > 0;) {
        const size_t g0 = gate_offset(n_gates, l);
        pos -= n_gates[l];
        for (size_t g = 0; g < n_gates[l]; ++g) {
            const uint32_t a = in0[g0 + g], b = in1[g0 + g];
            if (a >= cur_len || b >= cur_len) return -1;
            if (gate_type[g0 + g] == 0) ora_fr_add(&out[pos + g], &cur[a], &cur[b]);
            else ora_fr_mul(&out[pos + g], &cur[a], &cur[b]);
        }
        layer_len[l] = n_gates[l];
        cur = out + pos;
        cur_len = n_gates[l];
    }
    return 0;
}

/* Circuit::add_mult_mle (circuit.rs:59-97) */
int ora_circuit_add_mult_mle(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                             const uint32_t *in1, size_t layer_index, fr_t *add, fr_t *mul) {
    if (layer_index >= n_layers) return -1;
    const size_t size = ora_gkr_mle_size(layer_index);
    for (size_t i = 0; i < size; ++i) { ora_fr_zero(&add[i]); ora_fr_zero(&mul[i]); }
    const size_t g0 = gate_offset(n_gates, layer_index);
    for (size_t g = 0; g < n_gates[layer_index]; ++g) {
        const size_t idx = wiring_index(layer_index, g, in0[g0 + g], in1[g0 + g]);
        if (idx >= size) return -1;
        if (gate_type[g0 + g] == 0) ora_fr_one(&add[idx]); else ora_fr_one(&mul[idx]);
    }
    return 0;
}

static size_t log2_exact(size_t n) {
    size_t k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}

/* one layer's sumcheck: builds the two product terms, proves, appends to the proof, derives the next claim.
 * add_bc / mul_bc: the wiring tables already reduced to the (b, c) variables.  w: the layer's values. */
static int prove_layer(const fr_t *add_bc, const fr_t *mul_bc, const fr_t *w, size_t w_len, const fr_t *claimed,
                       ora_transcript_t *tr, ora_gkr_proof_t *proof, fr_t *alpha, fr_t *beta, fr_t *r_b, fr_t *r_c,
                       size_t *r_len, fr_t *next_claim) {
    const size_t n = w_len * w_len, nv = log2_exact(n);
    if (proof->n_proofs >= ORA_GKR_MAX_LAYERS || nv > ORA_GKR_MAX_ROUNDS) return -1;
    fr_t *tables = (fr_t *)malloc(4 * n * sizeof(fr_t));
    memcpy(tables, add_bc, n * sizeof(fr_t));                    /* term 0: [add, wb + wc] */
    ora_mle_add_distinct(tables + n, w, w_len, w, w_len);
    memcpy(tables + 2 * n, mul_bc, n * sizeof(fr_t));            /* term 1: [mul, wb * wc] */
    ora_mle_mul_distinct(tables + 3 * n, w, w_len, w, w_len);
    const size_t sizes[2] = {2, 2};
    const size_t k = proof->n_proofs;
    fr_t challenges[ORA_GKR_MAX_ROUNDS];
    int rc = ora_multi_composed_prove(tables, sizes, 2, n, claimed, 1, proof->round_polys[k], challenges);
    free(tables);
    if (rc != 0) return rc;
    proof->sums[k] = *claimed;
    proof->n_rounds[k] = nv;
    memcpy(proof->challenges[k], challenges, nv * sizeof(fr_t));
    uint8_t bytes[64 * ORA_SPARSE_MAX];
    for (size_t r = 0; r < nv; ++r) {                            /* transcript.commit(&sumcheck_proof.to_bytes()) */
        size_t nb = ora_sparse_to_bytes(bytes, &proof->round_polys[k][r]);
        ora_transcript_commit(tr, bytes, nb);
    }
    const size_t half = nv / 2;                                  /* challenges.split_at(len / 2) */
    memcpy(r_b, challenges, half * sizeof(fr_t));
    memcpy(r_c, challenges + half, (nv - half) * sizeof(fr_t));
    *r_len = half;
    if (ora_mle_evaluation(&proof->wb[k], w, w_len, r_b, half) != 0) return -1;
    if (ora_mle_evaluation(&proof->wc[k], w, w_len, r_c, nv - half) != 0) return -1;
    ora_transcript_challenge_fr(tr, alpha);
    ora_transcript_challenge_fr(tr, beta);
    fr_t t1, t2;
    ora_fr_mul(&t1, alpha, &proof->wb[k]);
    ora_fr_mul(&t2, beta, &proof->wc[k]);
    ora_fr_add(next_claim, &t1, &t2);
    proof->n_proofs = k + 1;
    return 0;
}

/* GKRProtocol::prove (protocol.rs:21-117).  layers / layer_len as produced by ora_circuit_evaluation. */
int ora_gkr_prove(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                  const uint32_t *in1, const fr_t *layers, const size_t *layer_len, ora_gkr_proof_t *proof) {
    memset(proof, 0, sizeof(*proof));
    if (n_layers < 1 || layer_len[0] + 1 != 2) return -1;       /* w_0 = [output, 0]: Multilinear::new needs a power of two */
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    proof->w0[0] = layers[0];
    ora_fr_zero(&proof->w0[1]);
    uint8_t wbytes[64];
    ora_mle_to_bytes(wbytes, proof->w0, 2);
    ora_transcript_commit(&tr, wbytes, 64);
    fr_t n_r[1];
    ora_transcript_challenge_fr(&tr, &n_r[0]);                   /* evaluate_n_challenge_into_field(&1) */
    fr_t claimed;
    ora_mle_evaluation(&claimed, proof->w0, 2, n_r, 1);

    fr_t alpha, beta, next;
    fr_t r_b[ORA_GKR_MAX_ROUNDS], r_c[ORA_GKR_MAX_ROUNDS];
    size_t r_len = 0;
    size_t off = layer_len[0];
    int rc = 0;
    {   /* layer one: gkr/src/utils.rs:12-56 */
        const size_t size = ora_gkr_mle_size(0);
        fr_t *add = (fr_t *)malloc(size * sizeof(fr_t)), *mul = (fr_t *)malloc(size * sizeof(fr_t));
        fr_t *add_bc = (fr_t *)malloc(size * sizeof(fr_t)), *mul_bc = (fr_t *)malloc(size * sizeof(fr_t));
        size_t on;
        rc = ora_circuit_add_mult_mle(n_layers, n_gates, gate_type, in0, in1, 0, add, mul);
        const size_t zeros[1] = {0};
        if (rc == 0) rc = ora_mle_partial_evaluations(add_bc, &on, add, size, n_r, zeros, 1);
        if (rc == 0) rc = ora_mle_partial_evaluations(mul_bc, &on, mul, size, n_r, zeros, 1);
        if (rc == 0 && on != layer_len[1] * layer_len[1]) rc = -1;
        if (rc == 0) rc = prove_layer(add_bc, mul_bc, layers + off, layer_len[1], &claimed, &tr, proof, &alpha, &beta, r_b, r_c, &r_len, &next);
        free(add); free(mul); free(add_bc); free(mul_bc);
        if (rc != 0) return rc;
        claimed = next;
        off += layer_len[1];
    }
    for (size_t li = 2; li <= n_layers; ++li) {                  /* protocol.rs:64-108 */
        const size_t size = ora_gkr_mle_size(li - 1);
        fr_t *add = (fr_t *)malloc(size * sizeof(fr_t)), *mul = (fr_t *)malloc(size * sizeof(fr_t));
        fr_t *t1 = (fr_t *)malloc(size * sizeof(fr_t)), *t2 = (fr_t *)malloc(size * sizeof(fr_t));
        size_t zeros[ORA_GKR_MAX_ROUNDS] = {0};
        size_t on = 0, on2 = 0;
        rc = ora_circuit_add_mult_mle(n_layers, n_gates, gate_type, in0, in1, li - 1, add, mul);
        /* add_alpha_beta = add(r_b, ., .) * alpha + add(r_c, ., .) * beta ; same for mul */
        fr_t *add_ab = NULL, *mul_ab = NULL;
        if (rc == 0) rc = ora_mle_partial_evaluations(t1, &on, add, size, r_b, zeros, r_len);
        if (rc == 0) rc = ora_mle_partial_evaluations(t2, &on2, add, size, r_c, zeros, r_len);
        if (rc == 0 && (on != on2 || on != layer_len[li] * layer_len[li])) rc = -1;
        if (rc == 0) {
            add_ab = (fr_t *)malloc(on * sizeof(fr_t));
            for (size_t i = 0; i < on; ++i) {
                fr_t x, y;
                ora_fr_mul(&x, &t1[i], &alpha);
                ora_fr_mul(&y, &t2[i], &beta);
                ora_fr_add(&add_ab[i], &x, &y);
            }
            rc = ora_mle_partial_evaluations(t1, &on, mul, size, r_b, zeros, r_len);
            if (rc == 0) rc = ora_mle_partial_evaluations(t2, &on2, mul, size, r_c, zeros, r_len);
        }
        if (rc == 0) {
            mul_ab = (fr_t *)malloc(on * sizeof(fr_t));
            for (size_t i = 0; i < on; ++i) {
                fr_t x, y;
                ora_fr_mul(&x, &t1[i], &alpha);
                ora_fr_mul(&y, &t2[i], &beta);
                ora_fr_add(&mul_ab[i], &x, &y);
            }
            rc = prove_layer(add_ab, mul_ab, layers + off, layer_len[li], &claimed, &tr, proof, &alpha, &beta, r_b, r_c, &r_len, &next);
        }
        free(add); free(mul); free(t1); free(t2); free(add_ab); free(mul_ab);
        if (rc != 0) return rc;
        claimed = next;
        off += layer_len[li];
    }
    return 0;
}

/* MultiComposedSumcheckVerifier::verify_partial (multi_composed_sumcheck.rs:141-181): 1 ok + subclaim, 0 failed */
static int verify_partial(const fr_t *sum, const ora_sparse_t *rps, size_t n_rounds, fr_t *sub_sum, fr_t *challenges) {
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t bytes[64 * ORA_SPARSE_MAX];
    ora_fr_to_bytes_be(bytes, sum);
    ora_transcript_commit(&tr, bytes, 32);
    fr_t claimed = *sum, zero, one;
    ora_fr_zero(&zero);
    ora_fr_one(&one);
    for (size_t r = 0; r < n_rounds; ++r) {
        size_t nb = ora_sparse_to_bytes(bytes, &rps[r]);
        ora_transcript_commit(&tr, bytes, nb);
        ora_transcript_challenge_fr(&tr, &challenges[r]);
        fr_t e0, e1, s;
        ora_sparse_evaluate(&e0, &rps[r], &zero);
        ora_sparse_evaluate(&e1, &rps[r], &one);
        ora_fr_add(&s, &e0, &e1);
        if (!ora_fr_eq(&s, &claimed)) return 0;
        ora_sparse_evaluate(&claimed, &rps[r], &challenges[r]);
    }
    *sub_sum = claimed;
    return 1;
}

/* GKRProtocol::verify (protocol.rs:119-196): 1 = accepted, 0 = rejected */
int ora_gkr_verify(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                   const uint32_t *in1, const fr_t *input, size_t n_input, const ora_gkr_proof_t *proof) {
    if (proof->n_proofs < 1 || proof->n_proofs > ORA_GKR_MAX_LAYERS) return 0;
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t wbytes[64], bytes[64 * ORA_SPARSE_MAX];
    ora_mle_to_bytes(wbytes, proof->w0, 2);
    ora_transcript_commit(&tr, wbytes, 64);
    fr_t n_r[1 + ORA_GKR_MAX_ROUNDS];
    ora_transcript_challenge_fr(&tr, &n_r[0]);
    fr_t claimed;
    ora_mle_evaluation(&claimed, proof->w0, 2, n_r, 1);
    fr_t alpha, beta, r_b[ORA_GKR_MAX_ROUNDS], r_c[ORA_GKR_MAX_ROUNDS], sub, challenges[ORA_GKR_MAX_ROUNDS];
    size_t r_len = 0;
    ora_fr_zero(&alpha);
    ora_fr_zero(&beta);
    {   /* generate_layer_one_verify_sumcheck, gkr/src/utils.rs:58-98 */
        if (!ora_fr_eq(&claimed, &proof->sums[0])) return 0;
        const size_t nr = proof->n_rounds[0];
        for (size_t r = 0; r < nr; ++r) {
            size_t nb = ora_sparse_to_bytes(bytes, &proof->round_polys[0][r]);
            ora_transcript_commit(&tr, bytes, nb);
        }
        if (!verify_partial(&proof->sums[0], proof->round_polys[0], nr, &sub, challenges)) return 0;
        const size_t size = ora_gkr_mle_size(0);
        if (1 + nr != log2_exact(size)) return 0;
        fr_t *add = (fr_t *)malloc(size * sizeof(fr_t)), *mul = (fr_t *)malloc(size * sizeof(fr_t));
        ora_circuit_add_mult_mle(n_layers, n_gates, gate_type, in0, in1, 0, add, mul);
        memcpy(n_r + 1, challenges, nr * sizeof(fr_t));          /* rbc = n_r ++ challenges */
        fr_t add_bc, mul_bc, s1, s2, f;
        ora_mle_evaluation(&add_bc, add, size, n_r, 1 + nr);
        ora_mle_evaluation(&mul_bc, mul, size, n_r, 1 + nr);
        free(add); free(mul);
        ora_fr_add(&s1, &proof->wb[0], &proof->wc[0]);
        ora_fr_mul(&s1, &add_bc, &s1);
        ora_fr_mul(&s2, &proof->wb[0], &proof->wc[0]);
        ora_fr_mul(&s2, &mul_bc, &s2);
        ora_fr_add(&f, &s1, &s2);
        if (!ora_fr_eq(&f, &sub)) return 0;
        ora_transcript_challenge_fr(&tr, &alpha);
        ora_transcript_challenge_fr(&tr, &beta);
        fr_t t1, t2;
        ora_fr_mul(&t1, &alpha, &proof->wb[0]);
        ora_fr_mul(&t2, &beta, &proof->wc[0]);
        ora_fr_add(&claimed, &t1, &t2);
        /* NB: the reference leaves r_b / r_c empty here (protocol.rs:133-134 are only set inside the loop) */
    }
    for (size_t i = 1; i < proof->n_proofs; ++i) {
        if (!ora_fr_eq(&claimed, &proof->sums[i])) return 0;
        const size_t nr = proof->n_rounds[i];
        for (size_t r = 0; r < nr; ++r) {
            size_t nb = ora_sparse_to_bytes(bytes, &proof->round_polys[i][r]);
            ora_transcript_commit(&tr, bytes, nb);
        }
        if (!verify_partial(&proof->sums[i], proof->round_polys[i], nr, &sub, challenges)) return 0;
        const size_t half = nr / 2;
        memcpy(r_b, challenges, half * sizeof(fr_t));
        memcpy(r_c, challenges + half, (nr - half) * sizeof(fr_t));
        r_len = half;
        ora_transcript_challenge_fr(&tr, &alpha);
        ora_transcript_challenge_fr(&tr, &beta);
        fr_t t1, t2;
        ora_fr_mul(&t1, &alpha, &proof->wb[i]);
        ora_fr_mul(&t2, &beta, &proof->wc[i]);
        ora_fr_add(&claimed, &t1, &t2);
    }
    /* final check against the input layer (protocol.rs:183-193); with a single proof r_b / r_c are empty and
     * evaluation(&[]) on a table of more than one entry is the evaluation_form.rs:163-167 panic -> rejected here */
    fr_t wb_in, wc_in, t1, t2, s;
    if (ora_mle_evaluation(&wb_in, input, n_input, r_b, r_len) != 0) return 0;
    if (ora_mle_evaluation(&wc_in, input, n_input, r_c, r_len) != 0) return 0;
    ora_fr_mul(&t1, &alpha, &wb_in);
    ora_fr_mul(&t2, &beta, &wc_in);
    ora_fr_add(&s, &t1, &t2);
    return ora_fr_eq(&s, &claimed) ? 1 : 0;
}
