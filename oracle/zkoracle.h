/*
 * zkoracle.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the proving hot path of aagbotemi/zk-cryptography
 * (pure Rust over arkworks 0.4.2; it cannot be built in this image: no
 * cargo/rustc, no vendored crates).  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libzkhip.so) never
 * links or calls it.
 *
 * Parity status
 *   pinned by the reference's own unit-test values (tests/test_oracle_kats.py):
 *     fold / evaluate / half-sums / hypercube sum / add+mul_distinct /
 *     composed product+sum / byte encoding / sparse interpolation /
 *     eq-point (SRS scalar) tables / size-16 root of unity / dense mul.
 *   parity UNPINNED (no reference test fixes a value; consistency checks only):
 *     Fiat-Shamir challenges, round-polynomial / proof bytes, commitment
 *     coordinates, NTT output vectors.  These rest on the published behaviour
 *     of ark-ff/ark-ec 0.4.2 and sha2 0.10 (Montgomery Fp, big-endian
 *     canonical bytes, from_be_bytes_mod_order, SHA-256) and are checked here
 *     against python ints + hashlib, prove->verify round trips and the
 *     commit == p(tau)*G identity.
 *
 * Third-party arithmetic restated (source not in /root/reference):
 *   ark-ff ^0.4.2, ark-ec ^0.4.2, ark-test-curves ^0.4.2 (bls12_381), sha2 ^0.10.8
 *   (Cargo.toml:21-23,32; no Cargo.lock, patch versions unpinned).
 */
#ifndef ZKORACLE_H
#define ZKORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BLS12-381 scalar field element, 4x64 little-endian limbs, Montgomery form
 * (R = 2^256 mod r) -- the in-memory form of ark_ff::Fp<MontBackend<FrConfig,4>,4>. */
typedef struct { uint64_t l[4]; } fr_t;
/* BLS12-381 base field element, 6x64 LE limbs, Montgomery form (R = 2^384 mod p). */
typedef struct { uint64_t l[6]; } fq_t;
/* G1 point, affine (x, y) + infinity flag. */
typedef struct { fq_t x, y; uint64_t inf; } g1_affine_t;
/* G1 point, Jacobian (X, Y, Z); Z == 0 is the identity (ark-ec Projective). */
typedef struct { fq_t x, y, z; } g1_jac_t;

/* ---- Fr ---------------------------------------------------------------- */
void ora_fr_add(fr_t *o, const fr_t *a, const fr_t *b);
void ora_fr_sub(fr_t *o, const fr_t *a, const fr_t *b);
void ora_fr_mul(fr_t *o, const fr_t *a, const fr_t *b);
void ora_fr_neg(fr_t *o, const fr_t *a);
int  ora_fr_inv(fr_t *o, const fr_t *a);               /* 0 if a == 0 */
void ora_fr_pow_u64(fr_t *o, const fr_t *a, uint64_t e);
void ora_fr_from_u64(fr_t *o, uint64_t v);
void ora_fr_one(fr_t *o);
void ora_fr_zero(fr_t *o);
int  ora_fr_is_zero(const fr_t *a);
int  ora_fr_eq(const fr_t *a, const fr_t *b);
void ora_fr_to_canonical(uint64_t out[4], const fr_t *a);   /* into_bigint() */
void ora_fr_from_canonical(fr_t *o, const uint64_t in[4]);  /* in < r */
void ora_fr_to_bytes_be(uint8_t out[32], const fr_t *a);    /* into_bigint().to_bytes_be() */
void ora_fr_from_be_bytes_mod_order(fr_t *o, const uint8_t *bytes, size_t len);
int  ora_fr_get_root_of_unity(fr_t *o, uint64_t n);         /* F::get_root_of_unity */

/* ---- Fq ---------------------------------------------------------------- */
void ora_fq_add(fq_t *o, const fq_t *a, const fq_t *b);
void ora_fq_sub(fq_t *o, const fq_t *a, const fq_t *b);
void ora_fq_mul(fq_t *o, const fq_t *a, const fq_t *b);
int  ora_fq_inv(fq_t *o, const fq_t *a);
void ora_fq_to_canonical(uint64_t out[6], const fq_t *a);
void ora_fq_from_canonical(fq_t *o, const uint64_t in[6]);

/* ---- SHA-256 + Fiat-Shamir transcript (fiat_shamir.rs:10-40) ------------- */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } ora_sha256_t;
void ora_sha256_init(ora_sha256_t *s);
void ora_sha256_update(ora_sha256_t *s, const uint8_t *d, size_t n);
void ora_sha256_final(ora_sha256_t *s, uint8_t out[32]);
typedef struct { ora_sha256_t hasher; } ora_transcript_t;
void ora_transcript_new(ora_transcript_t *t);
void ora_transcript_commit(ora_transcript_t *t, const uint8_t *d, size_t n);
void ora_transcript_challenge(ora_transcript_t *t, uint8_t out[32]);
void ora_transcript_challenge_fr(ora_transcript_t *t, fr_t *o);

/* ---- Multilinear, evaluation form (evaluation_form.rs) ------------------- */
int  ora_mle_partial_evaluation(fr_t *out, const fr_t *in, size_t n, const fr_t *r, size_t var_index);
int  ora_mle_partial_evaluations(fr_t *out, size_t *out_n, const fr_t *in, size_t n,
                                 const fr_t *pts, const size_t *var_indices, size_t n_pts);
int  ora_mle_evaluation(fr_t *out, const fr_t *in, size_t n, const fr_t *pts, size_t n_pts);
void ora_mle_half_sums(fr_t out[2], const fr_t *in, size_t n);
void ora_mle_sum(fr_t *out, const fr_t *in, size_t n);
void ora_mle_add_distinct(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
void ora_mle_mul_distinct(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
void ora_mle_to_bytes(uint8_t *out, const fr_t *in, size_t n);
/* add_to_front (:86-96): out[n * 2 * 2^k]; add_to_back (:98-110): out[n * 2^k] */
void ora_mle_add_to_front(fr_t *out, const fr_t *in, size_t n, size_t variable_length);
void ora_mle_add_to_back(fr_t *out, const fr_t *in, size_t n, size_t variable_length);
/* fast variant used ONLY as an all-cores CPU baseline (same arithmetic, OpenMP) */
int  ora_mle_partial_evaluation_mt(fr_t *out, const fr_t *in, size_t n, const fr_t *r, size_t var_index);

/* ---- Sparse univariate (sparse_univariate.rs) ----------------------------- */
#define ORA_SPARSE_MAX 16
typedef struct { fr_t coeff[ORA_SPARSE_MAX]; fr_t pow[ORA_SPARSE_MAX]; size_t len; } ora_sparse_t;
int  ora_sparse_interpolation(ora_sparse_t *o, const fr_t *xs, const fr_t *ys, size_t n);
void ora_sparse_add(ora_sparse_t *o, const ora_sparse_t *a, const ora_sparse_t *b);
void ora_sparse_evaluate(fr_t *o, const ora_sparse_t *p, const fr_t *x);
size_t ora_sparse_to_bytes(uint8_t *out, const ora_sparse_t *p);

/* ---- Sumcheck provers / verifiers ---------------------------------------- */
/* basic: sumcheck.rs:25-95.  round_polys: n_vars x 2 Fr; challenges: n_vars Fr */
int  ora_sumcheck_prove(const fr_t *evals, size_t n, fr_t *sum_out, fr_t *round_polys, fr_t *challenges);
int  ora_sumcheck_verify(const fr_t *evals, size_t n, const fr_t *sum, const fr_t *round_polys);
/* composed (product of K tables): composed_sumcheck.rs:28-95. round_polys: n_vars x (K+1) */
void ora_composed_sum(fr_t *sum, const fr_t *tables, size_t k, size_t n);
int  ora_composed_prove(const fr_t *tables, size_t k, size_t n, fr_t *round_polys, fr_t *challenges);
int  ora_composed_verify(const fr_t *tables, size_t k, size_t n, const fr_t *sum, const fr_t *round_polys);
/* multi-composed (sum of P product terms): multi_composed_sumcheck.rs:36-181.
 * tables: concatenation over terms of term_sizes[p] tables of n entries each. */
void ora_multi_composed_sum(fr_t *sum, const fr_t *tables, const size_t *term_sizes, size_t n_terms, size_t n);
int  ora_multi_composed_prove(const fr_t *tables, const size_t *term_sizes, size_t n_terms, size_t n,
                              const fr_t *sum, int partial, ora_sparse_t *round_polys, fr_t *challenges);
/* returns 1 ok, 0 oracle check failed, -1 "Verification failed" */
int  ora_multi_composed_verify(const fr_t *tables, const size_t *term_sizes, size_t n_terms, size_t n,
                               const fr_t *sum, const ora_sparse_t *round_polys, size_t n_rounds);

/* ---- GKR (gkr/src/protocol.rs, gkr/src/utils.rs) over the layered circuit (circuit/src/circuit.rs) ----------
 * A circuit travels as flat arrays: n_gates[l] gates in layer l (layer 0 = output), then per gate (layers
 * concatenated) gate_type (0 = Add, 1 = Mul) and the two input labels. */
#define ORA_GKR_MAX_LAYERS 24
#define ORA_GKR_MAX_ROUNDS 48
typedef struct {
    size_t n_proofs;
    fr_t sums[ORA_GKR_MAX_LAYERS];                                 /* ComposedSumcheckProof::sum */
    size_t n_rounds[ORA_GKR_MAX_LAYERS];
    ora_sparse_t round_polys[ORA_GKR_MAX_LAYERS][ORA_GKR_MAX_ROUNDS];
    fr_t wb[ORA_GKR_MAX_LAYERS], wc[ORA_GKR_MAX_LAYERS];
    fr_t w0[2];                                                    /* w_0_mle: [output, 0] */
    fr_t challenges[ORA_GKR_MAX_LAYERS][ORA_GKR_MAX_ROUNDS];       /* what prove_partial returned beside each proof (b then c) */
} ora_gkr_proof_t;
size_t ora_gkr_mle_size(size_t layer_index);
int  ora_circuit_evaluation(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                            const uint32_t *in1, const fr_t *input, size_t n_input, fr_t *out, size_t *layer_len);
int  ora_circuit_add_mult_mle(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                              const uint32_t *in1, size_t layer_index, fr_t *add, fr_t *mul);
int  ora_gkr_prove(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                   const uint32_t *in1, const fr_t *layers, const size_t *layer_len, ora_gkr_proof_t *proof);
/* the same prover on sparse containers (gkr_sparse.c): reaches depth 20, bit-identical to ora_gkr_prove where that runs */
int  ora_gkr_prove_sparse(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                          const uint32_t *in1, const fr_t *layers, const size_t *layer_len, ora_gkr_proof_t *proof);
int  ora_gkr_verify(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                    const uint32_t *in1, const fr_t *input, size_t n_input, const ora_gkr_proof_t *proof);

/* ---- G1 + KZG ------------------------------------------------------------- */
void ora_g1_generator(g1_jac_t *o);
void ora_g1_identity(g1_jac_t *o);
void ora_g1_add(g1_jac_t *o, const g1_jac_t *a, const g1_jac_t *b);
void ora_g1_double(g1_jac_t *o, const g1_jac_t *a);
void ora_g1_neg(g1_jac_t *o, const g1_jac_t *a);
void ora_g1_mul_bigint(g1_jac_t *o, const g1_jac_t *base, const uint64_t *scalar, size_t n_limbs);
void ora_g1_to_affine(g1_affine_t *o, const g1_jac_t *a);
void ora_g1_from_affine(g1_jac_t *o, const g1_affine_t *a);
int  ora_g1_is_on_curve(const g1_affine_t *a);
void ora_g1_batch_to_affine(g1_affine_t *o, const g1_jac_t *a, size_t n);
/* kzg/src/utils.rs:19-40 + polynomial/src/utils.rs:141-157 */
void ora_kzg_eq_points(fr_t *out, const fr_t *tau, size_t n_vars);
/* trusted_setup.rs:25-35 */
void ora_kzg_multilinear_srs_g1(g1_jac_t *out, const fr_t *tau, size_t n_vars);
/* univariate_kzg.rs:18-35 */
void ora_kzg_univariate_srs_g1(g1_jac_t *out, const fr_t *tau, size_t max_degree);
/* multilinear_kzg.rs:33-48 / univariate_kzg.rs:37-58 : naive sum_i srs[i].mul_bigint(coeff[i]) */
int  ora_kzg_commitment(g1_jac_t *out, const fr_t *coeffs, size_t n_coeffs, const g1_jac_t *srs, size_t n_srs,
                        int require_equal_len);
/* same sum computed by a CPU bucket method -- context number only, not the reference algorithm */
/* MultilinearKZG::open (multilinear_kzg.rs:50-88), naive as the reference; proofs[n_vars] Jacobian */
int  ora_kzg_open(fr_t *evaluation, g1_jac_t *proofs, const fr_t *evals, size_t n, const fr_t *points, size_t n_points,
                  const g1_jac_t *srs, size_t n_srs);
/* UnivariateKZG::open (univariate_kzg.rs:60-81) */
int  ora_univariate_kzg_open(fr_t *evaluation, g1_jac_t *proof, const fr_t *coeffs, size_t n, const fr_t *z,
                             const g1_jac_t *srs, size_t n_srs);
void ora_msm_pippenger(g1_jac_t *out, const fr_t *scalars, const g1_affine_t *pts, size_t n);

/* ---- NTT / Domain / multiply (utils.rs:281-324, domain.rs, evaluation.rs) -- */
void ora_serial_fft(fr_t *list, size_t n, const fr_t *w, uint32_t size_log);
int  ora_domain_fft(fr_t *out, const fr_t *coeffs, size_t n_coeffs, size_t domain_size);
int  ora_domain_ifft(fr_t *out, const fr_t *evals, size_t n_evals, size_t domain_size);
/* UnivariateEval::multiply: out has na+nb-1 coefficients */
int  ora_univariate_multiply(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
/* DenseUnivariatePolynomial Mul (schoolbook, dense_univariate.rs:210-233) ; returns out length */
size_t ora_dense_mul(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
void ora_dense_evaluate(fr_t *o, const fr_t *coeffs, size_t n, const fr_t *x);
/* divide_with_q_and_r (dense_univariate.rs:88-124); q, r: na entries each */
int  ora_dense_divide(fr_t *q, size_t *nq, fr_t *r, size_t *nr, const fr_t *a, size_t na, const fr_t *b, size_t nb);

#ifdef __cplusplus
}
#endif
#endif
