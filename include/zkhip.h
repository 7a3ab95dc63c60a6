/*
 * zkhip.h -- C ABI of libzkhip.so, the MI355X (gfx950) proving hot path behind the
 * polynomial::Multilinear / sumcheck prover / kzg::commitment / Domain surfaces of
 * aagbotemi/zk-cryptography.
 *
 * The reference has no FFI of its own (pure Rust traits); each entry point below is
 * what a Rust shim implementing the cited trait method would bind.
 *
 * Conventions
 *   - Fr element  = uint64_t[4]  little-endian limbs, Montgomery form (R = 2^256 mod r):
 *                   the in-memory form of ark_ff Fp<MontBackend<FrConfig,4>,4> (BLS12-381 Fr).
 *   - Fq element  = uint64_t[6]  same, R = 2^384 mod p.
 *   - G1 affine   = uint64_t[12] (x[6], y[6]) + one uint8_t infinity flag per point in a
 *                   separate array (what CurveGroup::normalize_batch yields).
 *   - Pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensor storage), h_* are
 *     host pointers.  The library never takes ownership of caller buffers.
 *   - Every call is enqueued on the context's HIP stream; calls that return values to h_*
 *     buffers synchronise that stream before returning, the others do not.
 *   - Return value: 0 on success, negative zkhip_status otherwise.  Shape errors that the
 *     reference raises as assert!/panic! come back as ZKHIP_ERR_SHAPE / ZKHIP_ERR_INDEX.
 *   - One context per host thread (or external locking); contexts are independent.
 *   - While a split-phase session (zkhip_sc_begin / zkhip_mc_begin .. finish / abort) is alive its tables live in the
 *     context's workspace: every other entry point that needs that workspace returns ZKHIP_ERR_BUSY until the session
 *     ends (a second zkhip_sc_begin gets an allocation of its own instead).
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZKHIP_OK = 0,
    ZKHIP_ERR_HIP = -1,      /* a HIP runtime call failed (zkhip_last_hip_error) */
    ZKHIP_ERR_SHAPE = -2,    /* assert!/assert_eq! on sizes in the reference */
    ZKHIP_ERR_INDEX = -3,    /* out-of-bounds index panic in the reference */
    ZKHIP_ERR_ARG = -4,      /* null pointer / unsupported argument */
    ZKHIP_ERR_NOMEM = -5,
    ZKHIP_ERR_BUSY = -6,     /* the context's workspace backs a live split-phase session (zkhip_sc_* / zkhip_mc_*):
                                finish or abort it first, or use another context */
    ZKHIP_ERR_PEER = -7,     /* a sharded prover: ANOTHER rank failed (its record arrived poisoned); this rank's outputs are void */
    ZKHIP_ERR_TIMEOUT = -8   /* a device-side wait gave up (the GKR prover's outer-transcript hasher did not see a round's items:
                                its workgroup and the one that publishes them were not resident together) */
} zkhip_status;

typedef struct zkhip_ctx zkhip_ctx;

/* ---- context / memory ------------------------------------------------------------ */
int zkhip_version(void);
const char *zkhip_status_string(int status);
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream); NULL = the
 * device's default (null) stream. */
int zkhip_ctx_create(zkhip_ctx **out, int device, void *stream);
int zkhip_ctx_destroy(zkhip_ctx *ctx);
int zkhip_ctx_set_stream(zkhip_ctx *ctx, void *stream);
int zkhip_ctx_synchronize(zkhip_ctx *ctx);
int zkhip_last_hip_error(zkhip_ctx *ctx);
/* device memory for hosts that do not bring their own allocator */
int zkhip_malloc(zkhip_ctx *ctx, void **d_ptr, size_t bytes);
int zkhip_free(zkhip_ctx *ctx, void *d_ptr);
int zkhip_memcpy_h2d(zkhip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int zkhip_memcpy_d2h(zkhip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* stream-ordered timing of the dominant kernel (used by bench.py's roofline leg):
 * when enabled, every fold launch is bracketed by HIP events on the context's stream. */
int zkhip_profile_enable(zkhip_ctx *ctx, int enable);
int zkhip_profile_read(zkhip_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches, double *bytes);
/* every bracketed launch since zkhip_profile_enable(ctx, 1) in issue order: names[32 * i] (NUL-terminated), start / stop in
 * microseconds relative to the first launch's start -- the stream-ordered timeline of a call, gaps included, without a
 * profiler attached (tools/timeline.py) */
int zkhip_profile_timeline(zkhip_ctx *ctx, uint32_t max_records, char *names, double *start_us, double *stop_us,
                           uint32_t *count);

/* ---- host-side Fr helpers (what `Fr::from(..)`, `into_bigint()` and the field operators are to a Rust caller) --- */
/* Synthetic benchmark / test inputs: n field elements, uniform over Fr, from a splitmix64-seeded xoshiro256**
 * stream (four draws = 256 bits LE, top bit masked, redrawn while >= r), in Montgomery form.  Deterministic per seed. */
int zkhip_synthetic_fr(uint64_t seed, size_t n, uint64_t *h_out);
int zkhip_fr_from_i64(int64_t v, uint64_t *h_out);                       /* Fr::from(v): Montgomery limbs of v mod r */
int zkhip_fr_to_canonical(const uint64_t *h_in, uint64_t *h_out);         /* into_bigint(): canonical LE limbs */
int zkhip_fr_add(const uint64_t *h_a, const uint64_t *h_b, uint64_t *h_out);
int zkhip_fr_sub(const uint64_t *h_a, const uint64_t *h_b, uint64_t *h_out);
int zkhip_fr_mul(const uint64_t *h_a, const uint64_t *h_b, uint64_t *h_out);

/* ---- Multilinear (polynomial/src/multilinear/evaluation_form.rs) -------------------- */
/* MultilinearTrait::partial_evaluation (interface.rs:9-13, evaluation_form.rs:123-141):
 * d_out[n/2] = fold of variable var_index at point r.  r may live on the host (h_r) or on the
 * device (d_r, e.g. a challenge produced by the transcript); exactly one is non-NULL. */
int zkhip_mle_partial_evaluation(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_r,
                                 const uint64_t *d_r, uint32_t var_index, uint64_t *d_out);
/* Entries the call above writes: n/2, or 0 when 2^var_index >= n -- the reference's pair list is empty there
 * (polynomial/src/utils.rs:37-50) and it returns an EMPTY table with n_vars - 1; the call then succeeds without touching d_out.
 * (0 as well for the shapes the reference panics on.) */
size_t zkhip_mle_partial_evaluation_len(size_t n, uint32_t var_index);
/* MultilinearTrait::partial_evaluations (evaluation_form.rs:143-159): successive folds; d_out must hold
 * n >> n_pts elements.  h_pts: n_pts x 4.  A LAST fold whose 2^index >= the current length leaves the reference's empty
 * table (nothing is written); any fold after that is its panic -> ZKHIP_ERR_SHAPE. */
int zkhip_mle_partial_evaluations(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_pts,
                                  const uint32_t *h_var_indices, size_t n_pts, uint64_t *d_out);
/* MultilinearTrait::evaluation (evaluation_form.rs:162-175): h_out[4]. n_pts must equal log2(n). */
int zkhip_mle_evaluation(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_pts, size_t n_pts,
                         uint64_t *h_out);
/* split_poly_into_two_and_sum_each_part (:68-74) and sum_over_the_boolean_hypercube (:80-84) /
 * Sumcheck::poly_sum (sumcheck.rs:25-27): h_out[12] = (lower-half sum, upper-half sum, total). */
int zkhip_mle_half_sums(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, uint64_t *h_out);
/* add_distinct / mul_distinct (:28-52): d_out[na*nb] */
int zkhip_mle_add_distinct(zkhip_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb,
                           uint64_t *d_out);
int zkhip_mle_mul_distinct(zkhip_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb,
                           uint64_t *d_out);
/* Add / Sub / Mul<F> (:178-251): op 0 add, 1 sub (d_b: nb elements), 2 scale (h_scalar[4]; nb ignored).  The reference
 * walks the left operand and indexes the right one (rhs.evaluations[i], :185,215): a longer rhs is read up to n, a shorter
 * one is the index panic -> ZKHIP_ERR_INDEX. */
int zkhip_mle_elementwise(zkhip_ctx *ctx, int op, const uint64_t *d_a, const uint64_t *d_b,
                          const uint64_t *h_scalar, size_t n, size_t nb, uint64_t *d_out);
/* add_to_front (:86-96): d_out[n * 2 * 2^variable_length] = the table repeated; add_to_back (:98-110):
 * d_out[n * 2^variable_length] = every entry repeated 2^variable_length times (new variables the polynomial does not
 * depend on, before / after its own). */
int zkhip_mle_add_to_front(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, uint32_t variable_length, uint64_t *d_out);
int zkhip_mle_add_to_back(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, uint32_t variable_length, uint64_t *d_out);
/* Multilinear::to_bytes (:54-62): 32 big-endian canonical bytes per element into d_out_bytes[32 n] */
int zkhip_mle_to_bytes(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, uint8_t *d_out_bytes);

/* ---- FiatShamirTranscript (transcripts/fiat-shamir/src/fiat_shamir.rs:10-40) ----------------------------------
 * Every prover here keeps its transcript on the device; this entry runs the same device code on bytes of the caller's choice, so that
 * the hash itself can be held against an independent SHA-256.  A transcript that has just answered a challenge() holds a fresh
 * hasher that absorbed the 32-byte digest (:21-25): h_prefix32 = that digest (NULL: FiatShamirTranscript::new()); then
 * commit(h_bytes[0 .. n)) (:17-19) and challenge() -> h_digest32, the 32 bytes evaluate_challenge_into_field reduces (:27-29).
 * One wave: the message schedule on the sixteen lanes of a row, the state rounds on six (csrc/transcript.hpp). */
int zkhip_transcript_challenge(zkhip_ctx *ctx, const uint8_t *h_prefix32, const uint8_t *h_bytes, size_t n, uint8_t *h_digest32);

/* ---- layered circuit: the GKR prover's table builders (circuit/src/circuit.rs) ----------------------------
 * Gates arrive as host arrays (h_gate_type: 0 = Add, 1 = Mul; h_in0 / h_in1: input labels); values stay in HBM.
 *   zkhip_circuit_layer_eval   one step of Circuit::evaluation (:31-57): d_out[g] = d_in[in0[g]] (+|*) d_in[in1[g]];
 *                              a label >= n_in is the reference's index panic -> ZKHIP_ERR_INDEX
 *   zkhip_circuit_add_mult_mle Circuit::add_mult_mle (:59-97): the dense 0/1 wiring tables of layer `layer_index`,
 *                              d_add / d_mul of zkhip_gkr_mle_size(layer_index) elements each, a one at index
 *                              gate << 2(l+1) | in0 << (l+1) | in1 (circuit/src/utils.rs:12-25) */
size_t zkhip_gkr_mle_size(uint32_t layer_index);
int zkhip_circuit_layer_eval(zkhip_ctx *ctx, const uint64_t *d_in, size_t n_in, const uint8_t *h_gate_type,
                             const uint32_t *h_in0, const uint32_t *h_in1, size_t n_gates, uint64_t *d_out);
int zkhip_circuit_add_mult_mle(zkhip_ctx *ctx, const uint8_t *h_gate_type, const uint32_t *h_in0, const uint32_t *h_in1,
                               size_t n_gates, uint32_t layer_index, uint64_t *d_add, uint64_t *d_mul);

/* GKRProtocol::prove (gkr/src/protocol.rs:21-117 with gkr/src/utils.rs:8-56) in one call, every table in HBM.  The dense
 * wiring tables of Circuit::add_mult_mle (2^(3 l + 2) entries) and the dense (b, c) tables of a layer's sumcheck
 * (2^(2 l + 2) entries) are not built: the layer prover sums over the gates and works on tables as wide as the layer
 * (rounds over b, then over c), giving the proof of the dense prover bit for bit.  2 * n_layers <= 48
 * rounds per layer proof, i.e. circuits up to depth 24 / width 2^24.  The prover's own transcript (protocol.rs:25: every layer's
 * proof bytes, then alpha and beta) runs on the DEVICE beside the rounds: the call enqueues all layers behind each other and
 * waits once, for the whole proof (ZKHIP_GKR_HOST_TRANSCRIPT=1 or ZKHIP_PIPE=0: the transcript on the host, one wait per layer;
 * the same proof).
 *   circuit    : n_layers layers, layer l with h_n_gates[l] gates (layer 0 = output); gate arrays concatenated
 *   evaluation : h_layer_ptrs[k], k = 0..n_layers, DEVICE tables as Circuit::evaluation returns them (output first,
 *                input last), h_layer_len[k] entries each.  The reference's shape panics (Multilinear::new on a
 *                non-power-of-two layer, mismatching table sizes) -> ZKHIP_ERR_SHAPE.
 * Outputs (host), one ComposedSumcheckProof per layer k < n_layers, R = 2 * n_layers rounds reserved per proof:
 *   h_sums[k*4]; h_n_rounds[k]; h_round_poly_lens[k*R + r]; h_round_polys[(k*R + r)*7*8] (coeff[4], pow[4] pairs as
 *   zkhip_multi_composed_prove); h_wb[k*4], h_wc[k*4] (GKRProof::wb_s / wc_s); h_w0[8] = w_0_mle = [output, 0];
 *   h_challenges (nullable) [(k*R + r)*4]: the sumcheck challenges of every layer (b = first half, c = second half;
 *   SuccintGKRProtocol::prove opens the input layer at the last layer's b and c, succint_protocol.rs:133-152). */
int zkhip_gkr_prove(zkhip_ctx *ctx, uint32_t n_layers, const size_t *h_n_gates, const uint8_t *h_gate_type,
                    const uint32_t *h_in0, const uint32_t *h_in1, const uint64_t *const *h_layer_ptrs,
                    const size_t *h_layer_len, uint64_t *h_sums, uint32_t *h_n_rounds, uint32_t *h_round_poly_lens,
                    uint64_t *h_round_polys, uint64_t *h_wb, uint64_t *h_wc, uint64_t *h_w0, uint64_t *h_challenges);

/* Device-resident circuit: the gate arrays and their groupings by first / second input (what the layer prover reads) are
 * validated, built and uploaded once; a prover that proves many inputs on one circuit then pays no per-proof host work
 * (zkhip_gkr_prove spends about half of a depth-20 proof there).  Layer l holds h_n_gates[l] <= 2^l gates (2 at l = 0) whose
 * inputs label the 2^(l+1) values below it (circuit/src/circuit.rs:59-97); a label out of range is reported as ZKHIP_ERR_INDEX by
 * the prover when it reaches that layer (where the reference panics), after the shape checks of the layers before it.
 * zkhip_gkr_prove_circuit = zkhip_gkr_prove on that circuit (same outputs, same errors for the tables). */
typedef struct zkhip_circuit zkhip_circuit;
int zkhip_circuit_create(zkhip_ctx *ctx, uint32_t n_layers, const size_t *h_n_gates, const uint8_t *h_gate_type,
                         const uint32_t *h_in0, const uint32_t *h_in1, zkhip_circuit **out);
void zkhip_circuit_destroy(zkhip_circuit *circuit);
int zkhip_gkr_prove_circuit(zkhip_circuit *circuit, const uint64_t *const *h_layer_ptrs, const size_t *h_layer_len,
                            uint64_t *h_sums, uint32_t *h_n_rounds, uint32_t *h_round_poly_lens, uint64_t *h_round_polys,
                            uint64_t *h_wb, uint64_t *h_wc, uint64_t *h_w0, uint64_t *h_challenges);
/* n_proofs independent proofs of one circuit in ONE call (gkr/benches/gkr_benchmark.rs:11-27 proves input after input): proof b reads
 * h_layer_ptrs[b * (n_layers + 1) ..] (the layer lengths are the circuit's: h_layer_len[n_layers + 1], shared) and writes its outputs at
 * b times the per-proof sizes of zkhip_gkr_prove_circuit's arrays (h_sums 4 L, h_n_rounds L, h_round_poly_lens 2 L^2, h_round_polys
 * 2 L^2 * 7 * 8, h_wb / h_wc 4 L, h_w0 8, h_challenges 2 L^2 * 4, L = n_layers; h_challenges and h_status may be NULL).  The proofs run side
 * by side on up to max_lanes (0 = 8; at most 12) internal lanes -- streams, scratch and transcript state of their own, created on first
 * use and kept with the context; the lanes' streams are spread over the stream priorities so that each has a hardware queue to itself --
 * and every lane replays the circuit's launch chain as a HIP graph over its own copy of the layer values.  A proof is a chain of small
 * dependent kernels, so throughput comes from independent proofs: Circuit::random(8) 1.30 ms alone, 0.29 ms per proof eight at a time;
 * depth 20 8.4 ms alone, 2.8 ms per proof.  Every proof equals the one zkhip_gkr_prove_circuit makes.  Returns the first non-zero status
 * (h_status[b]: each proof's own). */
int zkhip_gkr_prove_batch(zkhip_circuit *circuit, uint32_t n_proofs, uint32_t max_lanes, const uint64_t *const *h_layer_ptrs,
                          const size_t *h_layer_len, uint64_t *h_sums, uint32_t *h_n_rounds, uint32_t *h_round_poly_lens,
                          uint64_t *h_round_polys, uint64_t *h_wb, uint64_t *h_wc, uint64_t *h_w0, uint64_t *h_challenges, int *h_status);
/* The tables of one layer's sumcheck in the linear-time form, for a caller that runs the sumcheck itself --
 * the sharded prover shards these tables over the ranks (zkhip_mc_begin_ex) instead of calling zkhip_gkr_prove_circuit.
 * Layer `layer` (0 = output layer) of a device-resident circuit, d_w = the layer's input values (w_len = 2^(layer+1)).
 *   phase 0 (the rounds over b):  d_out[0] = Ha0, d_out[1] = Ha1, d_out[2] = Hm  (w_len entries each, caller-allocated); the
 *            claim's terms are [Ha0, W] + Ha1 (additive table) and [Hm, W].  h_rb / h_rc: the points the wiring is bound at
 *            (n = max(layer, 1) field elements each; h_rc NULL for the first layer proof), h_alpha / h_beta their weights.
 *   phase 1 (the rounds over c):  reads the s = layer + 1 challenges of phase 0 where the context recorded them;
 *            d_out[0] = Aa, d_out[1] = W(u) + W, d_out[2] = Am, d_out[3] = W(u) W; h_wu[4] = W(u) (= w_b of the proof).
 * zkhip_gkr_layer_tables_sharded is what a rank of a sharded proof calls (gkr/src/protocol.rs:61-108 with the evaluation tables
 * of BASELINE configs[3] "sharded across 8"): it builds ONLY the rows j * world + rank of every table -- w_len / world entries per
 * output, the rank-interleaved shard the sumcheck sweeps -- and phase 0 additionally writes the shard of W itself to d_out[3]
 * (world > 1).  What the rows gather from by wire index (gate weights, W, eq(u)) stays whole on every rank.  world = 1, rank = 0 is
 * zkhip_gkr_layer_tables. */
int zkhip_gkr_layer_tables(zkhip_circuit *circuit, uint32_t layer, const uint64_t *d_w, size_t w_len, const uint64_t *h_rb,
                           const uint64_t *h_rc, const uint64_t *h_alpha, const uint64_t *h_beta, int phase,
                           uint64_t *const *d_out, uint64_t *h_wu);
int zkhip_gkr_layer_tables_sharded(zkhip_circuit *circuit, uint32_t layer, const uint64_t *d_w, size_t w_len, const uint64_t *h_rb,
                                   const uint64_t *h_rc, const uint64_t *h_alpha, const uint64_t *h_beta, int phase,
                                   uint32_t world, uint32_t rank, uint64_t *const *d_out, uint64_t *h_wu);


/* ---- basic sumcheck prover (sumcheck/src/sumcheck.rs:25-61) -------------------------- */
/* Block sums of a table: d_out[(2^log_blocks + 1) * 4] = the sums of its 2^log_blocks equal consecutive blocks
 * followed by the total (= Sumcheck::poly_sum, sumcheck.rs:25-27, which also goes to h_total[4] if non-NULL).
 * The prover consumes them (see zkhip_sumcheck_prove) so that poly_sum() + prove() stream the table once for
 * the sums instead of twice.  zkhip_sumcheck_plan_log_blocks(n) is the granularity the prover wants (0: none). */
int zkhip_sumcheck_plan_log_blocks(size_t n);
int zkhip_mle_block_sums(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, uint32_t log_blocks, uint64_t *d_out,
                         uint64_t *h_total);
/* The same block sums with the TOTAL deferred where it would cost a launch of its own (the fine granularity of 2^24-entry
 * tables, the overlapped plan): zkhip_sumcheck_prove with both claimed-sum arguments NULL absorbs the true sum out of its own sum tree, so a
 * poly_sum() whose caller proves next need not compute it; zkhip_mle_block_sums_total (two small launches + a copy) delivers it
 * to d_block_sums + 4 * 2^log_blocks and h_total[4] when the caller wants to read `self.sum` first. */
int zkhip_mle_block_sums_deferred(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, uint32_t log_blocks, uint64_t *d_out);
int zkhip_mle_block_sums_total(zkhip_ctx *ctx, uint64_t *d_block_sums, uint32_t log_blocks, uint64_t *h_total);
/* Sumcheck::prove (sumcheck.rs:29-61).  Inputs:
 *   h_claimed_sum[4]  `self.sum` as the caller holds it (what poly_sum() stored, or the default zero): the
 *                     transcript absorbs exactly this.  Alternatively d_claimed_sum[4]: the same value still on
 *                     the device (zkhip_mle_block_sums leaves the total at d_out + 4 * 2^log_blocks), which spares
 *                     poly_sum() a device->host round trip.  Both NULL = absorb the true sum, computed here.
 *   d_block_sums      optional device array from zkhip_mle_block_sums(.., log_blocks, ..); NULL = computed here.
 * Host outputs: h_sum[4] (the absorbed sum); h_round_polys[n_vars*2*4] (the Vec<Multilinear> of 2
 * evaluations each, :11-15); h_challenges[n_vars*4].  d_evals is not modified.
 * Internally k rounds of transcript run on 2^k block sums, then one k-variable fold pass follows
 * (csrc/multifold_kernels.hpp); the values are those of the round-by-round loop, bit for bit. */
int zkhip_sumcheck_prove(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_claimed_sum,
                         const uint64_t *d_claimed_sum, const uint64_t *d_block_sums, uint32_t log_blocks, uint64_t *h_sum,
                         uint64_t *h_round_polys, uint64_t *h_challenges);

/* Sumcheck::prove in flight: begin enqueues the whole proof (same inputs as zkhip_sumcheck_prove) and returns a ticket, end
 * waits for that proof and delivers the same outputs (all-NULL outputs abandon it).  Up to eight proofs may be in flight.  Each
 * runs with workspace and scratch of its own (and on the library's internal streams) behind what the caller's stream held at `begin`, so the streaming passes
 * of one proof (and the zkhip_mle_block_sums of the next table, on the caller's stream) overlap the transcript rounds of the
 * others; `end` orders the caller's stream behind the proof again.  The table, its block sums and a device-resident claimed sum
 * must stay untouched between begin and end.  While a proof is in flight zkhip_sumcheck_prove returns ZKHIP_ERR_BUSY (the
 * result slots are taken).
 * From the third proof in flight on, tables of >= 2^24 entries: `begin` enqueues the proof's first half only.  The big fold goes onto
 * the CALLER's stream -- so that one stream carries every streaming pass of every proof back to back -- and is enqueued by a later
 * `begin` (behind the block sums of the next one to three tables), by `end` of that proof or of a younger one, by
 * zkhip_ctx_synchronize, zkhip_ctx_set_stream (on the old stream) and zkhip_ctx_destroy.  A caller that begins proofs and then only
 * waits on its own stream therefore must call `end` (or zkhip_ctx_synchronize) for them to complete; a second half that cannot be
 * enqueued is reported by `end` of its proof. */
int zkhip_sumcheck_prove_begin(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_claimed_sum,
                               const uint64_t *d_claimed_sum, const uint64_t *d_block_sums, uint32_t log_blocks, uint32_t *ticket);
int zkhip_sumcheck_prove_end(zkhip_ctx *ctx, uint32_t ticket, uint64_t *h_sum, uint64_t *h_round_polys, uint64_t *h_challenges);

/* ---- the same prover, split per phase, for a table SHARDED over several GPUs ---------------------
 * The N = n_local * world entries are partitioned by their low index bits: rank g holds entry i = j*world + g
 * at local index j.  Rounds fold variable 0 (the most significant index bit), so every fold is local; per
 * round each rank contributes its two partial half sums (64 bytes), the host all-gathers them (RCCL over xGMI
 * through torch.distributed, or any other transport) and every rank absorbs the same sums into its own copy
 * of the transcript.  Once the whole remaining table fits one workgroup's LDS it is all-gathered and the
 * last rounds run replicated inside one kernel.  Modular addition is not an RCCL reduction, hence all-gather + local add.
 *   begin -> { local_half_sums -> [all-gather] -> absorb -> fold } x log2(n_local)
 *         -> local_value -> [all-gather] -> tail -> finish                                            */
typedef struct zkhip_sc_state zkhip_sc_state;
int zkhip_sc_begin(zkhip_ctx *ctx, const uint64_t *d_local_evals, size_t n_local, zkhip_sc_state **out);
int zkhip_sc_local_len(zkhip_sc_state *st, size_t *n_local_now);
int zkhip_sc_local_half_sums(zkhip_sc_state *st, uint64_t *d_out /* [2][4] */);
/* d_gathered[world][2][4]: every rank's (lower, upper) partial sums in rank order.  First round only:
 * h_claimed_sum as in zkhip_sumcheck_prove (NULL = the true sum). */
int zkhip_sc_absorb(zkhip_sc_state *st, const uint64_t *d_gathered, uint32_t world, const uint64_t *h_claimed_sum);
int zkhip_sc_fold(zkhip_sc_state *st);
/* copy of the current local table (n_local_now entries) -- gathered by the host once the whole remaining
 * table (n_local_now * world entries) fits zkhip_sc_tail_capacity() entries (twice what one workgroup's LDS holds:
 * zkhip_sc_tail runs the first round of a table of that size by itself, which costs less than another exchange) */
int zkhip_sc_local_table(zkhip_sc_state *st, uint64_t *d_out);
int zkhip_sc_tail_capacity(void);
/* d_values[m][4]: the whole remaining table in natural order (entry j*world + g = rank g's local entry j),
 * m a power of two <= zkhip_sc_tail_capacity(); runs ALL remaining log2(m) rounds replicated.  h_claimed_sum
 * only matters when no round has been absorbed yet. */
int zkhip_sc_tail(zkhip_sc_state *st, const uint64_t *d_values, uint32_t m, const uint64_t *h_claimed_sum);
/* Stage form of the same protocol (what zk_cryptography_amd.distributed uses): instead of one exchange per round,
 * one exchange per STAGE of k rounds -- the ranks all-gather their 2^k partial block sums (32 * 2^k bytes each),
 * every rank runs the k rounds of transcript on the summed block sums, then folds its shard by k variables in one
 * pass (csrc/multifold_kernels.hpp).  A 2^27-entry table over 8 GPUs needs 3 such exchanges plus the final gather.
 *   begin -> { stage_plan(k > 0) -> stage_block_sums -> [all-gather] -> stage_absorb -> stage_fold }*
 *         -> stage_plan(k == 0) -> local_table -> [all-gather + interleave] -> tail -> finish */
int zkhip_sc_stage_plan(zkhip_sc_state *st, uint32_t world, uint32_t *k_out);
int zkhip_sc_stage_block_sums(zkhip_sc_state *st, uint64_t *d_out /* [2^k][4] */);
/* d_gathered[world][2^k][4] in rank order; h_claimed_sum as in zkhip_sc_absorb (first stage only) */
int zkhip_sc_stage_absorb(zkhip_sc_state *st, const uint64_t *d_gathered, uint32_t world, const uint64_t *h_claimed_sum);
int zkhip_sc_stage_fold(zkhip_sc_state *st);
/* Overlapped stage: the single-GPU plan of 2^19..2^24-entry tables in exchange form, for the shard a rank holds.
 *   begin -> overlap_plan(k1 > 0) -> overlap_sums -> [all-gather 2^k1 sums] -> overlap_rounds1 -> [all-gather mid_entries sums,
 *            while the k1-variable fold of the shard runs on the context's fold stream] -> overlap_rounds2
 *         -> local_table (256 entries) -> [all-gather + interleave] -> tail -> finish
 * Rounds 1..k1 run on the gathered coarse block sums, rounds k1+1..k1+k2 on the gathered fine sums folded by the first k1
 * challenges, beside the big fold; three exchanges, as in the stage form.  *k1 == 0: the plan does not apply (shard outside
 * 2^19..2^24 entries, rounds already absorbed, or 256 * world entries exceed the tail) -- use the stage form. */
int zkhip_sc_overlap_plan(zkhip_sc_state *st, uint32_t world, uint32_t *k1, uint32_t *k2, uint32_t *mid_entries);
int zkhip_sc_overlap_sums(zkhip_sc_state *st, uint64_t *d_out /* [2^k1][4] */);
/* d_gathered[world][2^k1][4] in rank order; d_mid[mid_entries][4] receives what this rank contributes to the second exchange */
int zkhip_sc_overlap_rounds1(zkhip_sc_state *st, const uint64_t *d_gathered, uint32_t world, const uint64_t *h_claimed_sum,
                             uint64_t *d_mid);
/* d_gathered[world][mid_entries][4] in rank order */
int zkhip_sc_overlap_rounds2(zkhip_sc_state *st, const uint64_t *d_gathered, uint32_t world);
/* copies out the proof (as zkhip_sumcheck_prove; *n_rounds rounds were recorded) and releases the state */
int zkhip_sc_finish(zkhip_sc_state *st, uint64_t *h_sum, uint64_t *h_round_polys, uint64_t *h_challenges,
                    uint32_t *n_rounds);
/* releases the state without reading anything back (error paths: a failed collective, an exception in the host loop) */
int zkhip_sc_abort(zkhip_sc_state *st);

/* ---- composed sumcheck provers (sumcheck/src/composed/) --------------------------------------- */
/* A product term is K tables (ComposedMultilinear, polynomial/src/composed/composed_multilinear.rs:8-18) of n
 * entries each, given as a HOST array of K DEVICE pointers.  K <= 5; at most 4 terms. */
/* ComposedSumcheck::calculate_poly_sum (composed_sumcheck.rs:28-30): sum_x prod_k f_k(x) -> h_sum[4] */
int zkhip_composed_sum(zkhip_ctx *ctx, const uint64_t *const *h_table_ptrs, uint32_t k, size_t n, uint64_t *h_sum);
/* ComposedMultilinearTrait::element_wise_product (op 0) / element_wise_add (op 1) as materialised vectors
 * (polynomial/src/composed/composed_multilinear.rs:105-119; trait at polynomial/src/interface.rs:15-19; caller
 * sumcheck/src/utils.rs:46): d_out[i] = prod_k / sum_k table_k[i], i < n = the FIRST table's length (the reference indexes
 * every table up to polys[0].len()).  Any k >= 1 (not limited to 5); k == 0 is the `self.polys[0]` index panic ->
 * ZKHIP_ERR_INDEX. */
int zkhip_composed_element_wise(zkhip_ctx *ctx, int op, const uint64_t *const *h_table_ptrs, uint32_t k, size_t n,
                                uint64_t *d_out);
/* ComposedSumcheck::prove (composed_sumcheck.rs:32-67): h_round_polys[n_vars*(k+1)*4] (evaluations at
 * 0..=k per round), h_challenges[n_vars*4]. */
int zkhip_composed_prove(zkhip_ctx *ctx, const uint64_t *const *h_table_ptrs, uint32_t k, size_t n,
                         uint64_t *h_round_polys, uint64_t *h_challenges);
/* MultiComposedSumcheckProver::calculate_poly_sum (multi_composed_sumcheck.rs:36-45) */
int zkhip_multi_composed_sum(zkhip_ctx *ctx, const uint64_t *const *h_table_ptrs, const uint32_t *h_term_sizes,
                             uint32_t n_terms, size_t n, uint64_t *h_sum);
/* MultiComposedSumcheckProver::prove (partial = 0, :47-54: every table's bytes are absorbed first) and
 * ::prove_partial (partial = 1, :56-62), both through prove_internal (:64-121).
 *   h_table_ptrs: sum(h_term_sizes) device pointers, term after term;  h_sum[4]: the claimed sum.
 * Outputs: h_round_poly_lens[n_vars] = #monomials of each round's SparseUnivariatePolynomial;
 *          h_round_polys[n_vars*7*8] = per round up to 7 monomials (coeff[4], pow[4]) in Montgomery form;
 *          h_challenges[n_vars*4]. */
int zkhip_multi_composed_prove(zkhip_ctx *ctx, const uint64_t *const *h_table_ptrs, const uint32_t *h_term_sizes,
                               uint32_t n_terms, size_t n, const uint64_t *h_sum, int partial,
                               uint32_t *h_round_poly_lens, uint64_t *h_round_polys, uint64_t *h_challenges);

/* The same provers over tables SHARDED across `world` ranks (rank g holds entry j*world + g of every table at
 * local index j; the rounds fold the most significant variable, so folds are local).  ComposedSumcheck::prove (multi = 0,
 * one term) and MultiComposedSumcheckProver::prove_partial (multi = 1, h_sum = the claimed sum of the WHOLE tables).  Per
 * round every rank contributes one record of its partial sums -- term p's evaluations at t = 0..K_p, term after term,
 * zkhip_mc_record_len field elements (<= 24, i.e. <= 768 bytes) -- the host all-gathers the records (RCCL over xGMI) and
 * every rank closes the round on the gathered records (sum, interpolation, transcript, challenge), replicated.  A product of
 * tables does not commute with block sums, so there is no stage form here: one exchange per round.  Once the remaining
 * tables of all ranks fit one workgroup's LDS together they are gathered and the last rounds run replicated in one launch.
 *   begin -> { round_sums -> [all-gather] -> absorb } while local_len * world > tail_capacity
 *         -> local_tables -> [all-gather + interleave] -> tail -> finish
 * One session per context at a time (its buffers live in the context's workspace). */
typedef struct zkhip_mc_state zkhip_mc_state;
int zkhip_mc_begin(zkhip_ctx *ctx, const uint64_t *const *h_local_table_ptrs, const uint32_t *h_term_sizes, uint32_t n_terms,
                   size_t n_local, uint32_t world, int multi, const uint64_t *h_sum, zkhip_mc_state **out);
/* The general form (what the sharded GKR prover uses): d_local_lin (nullable) names per term an ADDITIVE
 * table (term = product of its tables + that table; terms of <= 2 tables only); cont = 1 continues the transcript and the
 * array of recorded rounds of the previous session of this context from round out_base on (h_sum is then not absorbed and may
 * be NULL) -- a sumcheck over (b, c) runs as a session over b followed by a session over c, and the second session's finish
 * delivers the rounds of both. */
int zkhip_mc_begin_ex(zkhip_ctx *ctx, const uint64_t *const *h_local_table_ptrs, const uint32_t *h_term_sizes, uint32_t n_terms,
                      const uint64_t *const *h_local_lin_ptrs, size_t n_local, uint32_t world, int multi, const uint64_t *h_sum,
                      int cont, uint32_t out_base, zkhip_mc_state **out);
int zkhip_mc_record_len(zkhip_mc_state *st, uint32_t *rec, uint32_t *n_tables);
/* entries per local table the next round's sums run over (n_local, then n_local/2, ...) */
int zkhip_mc_local_len(zkhip_mc_state *st, size_t *n_now);
int zkhip_mc_tail_capacity(zkhip_mc_state *st, uint32_t *entries_per_table);
/* folds the local tables at the previous challenge (rounds > 0) and writes this rank's record d_out[rec][4] */
int zkhip_mc_round_sums(zkhip_mc_state *st, uint64_t *d_out);
/* d_gathered[world][rec][4]: every rank's record in rank order */
int zkhip_mc_absorb(zkhip_mc_state *st, const uint64_t *d_gathered, uint32_t world);
/* the local tables as the next round would see them, d_out[n_tables][local_len][4] */
/* Two rounds per exchange for claims whose terms are products of TWO tables (csrc/composed_stage.hpp): a product does not commute
 * with block sums but is bilinear in them, so the next two round polynomials are functions of the 16 cross-block sums
 * C[a][b] = sum_j A[a][j] B[b][j] (a, b = the two leading index bits) and the 4 block sums of an additive table.
 * zkhip_mc_stage_record_len: the record length in field elements (n_terms * 20), or 0 when the next step cannot be a stage;
 * zkhip_mc_stage_sums: this rank's record; zkhip_mc_stage_absorb: the gathered records of all ranks -> two transcript rounds
 * (round polynomials and challenges recorded as by zkhip_mc_absorb) and the fold of every local table by both challenges.
 * Bit-identical to two zkhip_mc_round_sums / zkhip_mc_absorb steps (multi_composed_sumcheck.rs:76-104). */
int zkhip_mc_stage_record_len(zkhip_mc_state *s, uint32_t *vals);
int zkhip_mc_stage_sums(zkhip_mc_state *s, uint64_t *d_out);
int zkhip_mc_stage_absorb(zkhip_mc_state *s, const uint64_t *d_gathered, uint32_t world);
int zkhip_mc_local_tables(zkhip_mc_state *st, uint64_t *d_out);
/* d_tables[n_tables][m][4]: the whole remaining tables in natural order (entry j*world + g = rank g's local entry j),
 * m = local_len * world <= tail_capacity; runs ALL remaining log2(m) rounds */
int zkhip_mc_tail(zkhip_mc_state *st, const uint64_t *d_tables, uint32_t m);
/* copies out the proof in the layouts of zkhip_composed_prove / zkhip_multi_composed_prove (log2(n_local * world) rounds)
 * and releases the state; all-NULL outputs just release it */
int zkhip_mc_finish(zkhip_mc_state *st, uint32_t *h_round_poly_lens, uint64_t *h_round_polys, uint64_t *h_challenges);
int zkhip_mc_abort(zkhip_mc_state *st);

/* ---- the sharded provers behind ONE call each (BASELINE configs[3] / [4]) -------------------------------------
 * The exchange protocols -- stage plans, exchange order, tail gather, the per-layer GKR loop, the commit merge -- run INSIDE the
 * library (csrc/shard_protocol.hpp): a host binds one entry point per prover and supplies a communicator.  The loops they stand
 * for: sumcheck/src/sumcheck.rs:29-61, sumcheck/src/composed/composed_sumcheck.rs:32-67,
 * sumcheck/src/composed/multi_composed_sumcheck.rs:64-121, gkr/src/protocol.rs:61-108, kzg/src/multilinear_kzg.rs:33-48.
 *
 * zkhip_comm = this rank's end of the exchange, bound to a context (one process per GPU; rank g holds entry j * world + g of every
 * sharded table / SRS at local index j).  Every exchange is an ALL-GATHER of a few hundred bytes to 64 KiB per rank, issued on the
 * context's stream: it is ordered behind the kernels that produced its payload and the kernels that consume it are enqueued behind
 * it -- no host wait per exchange.  Two transports:
 *   zkhip_comm_create       a caller-supplied all-gather.  fn(user, d_send, d_recv, bytes_per_rank, stream) must deliver every rank's
 *                           d_send[bytes_per_rank] into d_recv[rank * bytes_per_rank ..] on all ranks, ordered on `stream` (a
 *                           hipStream_t) or completed before it returns; 0 = success.  A Rust host plugs in rccl-sys
 *                           (ncclAllGather(d_send, d_recv, bytes, ncclUint8, comm, stream)), the tests plug in gloo.  world == 1:
 *                           fn may be NULL (nothing is exchanged).
 *   zkhip_comm_create_rccl  the library opens librccl.so itself (dlopen at run time: libzkhip does not link it; ZKHIP_RCCL_LIB
 *                           overrides the name) and creates an RCCL communicator from a 128-byte unique id: rank 0 calls
 *                           zkhip_rccl_unique_id, the host distributes the id to all ranks by any means, every rank calls
 *                           zkhip_comm_create_rccl (collective, like ncclCommInitRank).
 * world must be a power of two (the tables have 2^n entries).  A comm is used by one host thread at a time, like its context.
 * A FAILING RANK DOES NOT HANG ITS PEERS (csrc/shard_protocol.hpp).  The sharded entry points are collective calls; a rank whose step
 * fails for a reason of its own -- out of memory, a busy workspace, a HIP error, at begin or between two exchanges -- stays in the
 * protocol: it skips its compute and enters every remaining exchange with a POISON record (first element all ones, which no field
 * element is; the buffers for that are set aside when the comm is created), then returns its own error.  Behind every gather one tiny
 * kernel looks at the first element of each rank's record and raises a sticky host-mapped flag; the healthy ranks read it where they
 * wait for the GPU anyway -- the end of the proof, the end of each GKR layer -- and return ZKHIP_ERR_PEER (their outputs are void).
 * zkhip_kzg_commit_sharded carries a status word in its one 128-byte record.  A call that collects nothing (zkhip_mc_prove_sharded
 * with all-NULL outputs) leaves the flag for the next collecting call on the comm.  Not covered: a process that dies, and argument /
 * shape errors, which depend only on values every rank holds and return at once on all of them.
 * zkhip_comm_inject_failure: test hook -- the next protocol run on this comm fails with `status` in front of its exchange number
 * exchange_index (0-based), as if a compute step had; exchange_index < 0 disarms.
 * zkhip_rccl_version: ncclGetVersion of the RCCL the library resolved (major * 10000 + minor * 100 + patch), ZKHIP_ERR_HIP if none. */
typedef int (*zkhip_all_gather_fn)(void *user, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
typedef struct zkhip_comm zkhip_comm;
int zkhip_comm_create(zkhip_ctx *ctx, uint32_t rank, uint32_t world, zkhip_all_gather_fn fn, void *user, zkhip_comm **out);
int zkhip_rccl_unique_id(uint8_t *h_id128);
int zkhip_rccl_version(int *version);
int zkhip_comm_create_rccl(zkhip_ctx *ctx, const uint8_t *h_id128, uint32_t rank, uint32_t world, zkhip_comm **out);
int zkhip_comm_destroy(zkhip_comm *comm);
/* one exchange as the provers issue them (d_recv: world * bytes_per_rank); cumulative counters of this comm */
int zkhip_comm_all_gather(zkhip_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);
int zkhip_comm_stats(zkhip_comm *comm, uint64_t *exchanges, uint64_t *bytes_sent);
int zkhip_comm_inject_failure(zkhip_comm *comm, int exchange_index, int status);
/* the in-library cost of an exchange: `iters` all-gathers of bytes_per_rank back to back (one wait at the end), then `iters` with
 * the host waiting for each -- microseconds per exchange (what bench.py reports as `exchange`) */
int zkhip_comm_measure(zkhip_comm *comm, size_t bytes_per_rank, uint32_t iters, double *us_back_to_back, double *us_host_wait);

/* Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61) of the table whose rank-interleaved shard d_local_evals[n_local] this rank
 * holds.  Outputs as zkhip_sumcheck_prove, identical on every rank (the transcript is replicated): log2(n_local * world) rounds.
 * h_claimed_sum as zkhip_sumcheck_prove (NULL: the true sum of the WHOLE table).  *exchanges (nullable): all-gathers issued
 * (3 for shards of 2^19..2^24 entries: coarse sums | fine sums beside the shard's fold | the gathered tail).
 * zkhip_sc_prove_sharded: the same on a session from zkhip_sc_begin, which it finishes (releases) whatever it returns. */
int zkhip_sumcheck_prove_sharded(zkhip_comm *comm, const uint64_t *d_local_evals, size_t n_local, const uint64_t *h_claimed_sum,
                                 uint64_t *h_sum, uint64_t *h_round_polys, uint64_t *h_challenges, uint32_t *exchanges);
int zkhip_sc_prove_sharded(zkhip_sc_state *st, zkhip_comm *comm, const uint64_t *h_claimed_sum, uint64_t *h_sum,
                           uint64_t *h_round_polys, uint64_t *h_challenges, uint32_t *exchanges);
/* ComposedSumcheck::prove (composed_sumcheck.rs:32-67) and MultiComposedSumcheckProver::prove_partial
 * (multi_composed_sumcheck.rs:56-121; h_sum = the claimed sum of the WHOLE tables) over rank-interleaved shards of every table;
 * outputs in the layouts of zkhip_composed_prove / zkhip_multi_composed_prove, identical on every rank.
 * use_stages: 1 = two rounds per exchange where every term is a product of two tables, 0 = one exchange per round, < 0 = the
 * default (stages when world > 1: they save exchanges, not work).
 * zkhip_mc_prove_sharded: the same on a session from zkhip_mc_begin / _begin_ex, which it finishes whatever it returns; all-NULL
 * outputs run the rounds and release the session without reading anything back (a session that continues it delivers both). */
int zkhip_composed_prove_sharded(zkhip_comm *comm, const uint64_t *const *h_local_table_ptrs, uint32_t k, size_t n_local,
                                 int use_stages, uint64_t *h_round_polys, uint64_t *h_challenges, uint32_t *exchanges);
int zkhip_multi_composed_prove_sharded(zkhip_comm *comm, const uint64_t *const *h_local_table_ptrs, const uint32_t *h_term_sizes,
                                       uint32_t n_terms, size_t n_local, const uint64_t *h_sum, int use_stages,
                                       uint32_t *h_round_poly_lens, uint64_t *h_round_polys, uint64_t *h_challenges,
                                       uint32_t *exchanges);
int zkhip_mc_prove_sharded(zkhip_mc_state *st, zkhip_comm *comm, int use_stages, uint32_t *h_round_poly_lens,
                           uint64_t *h_round_polys, uint64_t *h_challenges, uint32_t *exchanges);
/* GKRProtocol::prove (gkr/src/protocol.rs:21-117) with every layer's sumcheck tables SHARDED over the ranks (BASELINE configs[3]:
 * "evals sharded across 8"): rank g builds only rows j * world + g of a layer's seven linear-size tables and the two sessions per
 * layer (rounds over b, rounds over c) run on those shards, two rounds per exchange; layers narrower than 2 * world values run whole
 * on every rank.  The layer VALUES (h_layer_ptrs, as zkhip_gkr_prove_circuit) stay whole on every rank -- random wiring reads any of
 * them.  Inputs and outputs as zkhip_gkr_prove_circuit, the proof identical on every rank and to the unsharded prover's, bit for
 * bit.  One host synchronisation per layer (the outer transcript), none per exchange. */
int zkhip_gkr_prove_sharded(zkhip_circuit *circuit, zkhip_comm *comm, const uint64_t *const *h_layer_ptrs, const size_t *h_layer_len,
                            int use_stages, uint64_t *h_sums, uint32_t *h_n_rounds, uint32_t *h_round_poly_lens,
                            uint64_t *h_round_polys, uint64_t *h_wb, uint64_t *h_wc, uint64_t *h_w0, uint64_t *h_challenges,
                            uint32_t *exchanges);
/* MultilinearKZG::commitment / UnivariateKZG::commitment (kzg/src/multilinear_kzg.rs:33-48, univariate_kzg.rs:37-58) over
 * (scalars, SRS) sharded the same way: a full sub-MSM on this rank's shard (arguments as zkhip_kzg_commit; d_table non-NULL: the
 * shard's shifted-SRS table, as zkhip_kzg_commit_table), ONE all-gather of the partial commitments (128 bytes per rank), their group
 * sum on every rank.  require_equal_len applies to the local shard. */
int zkhip_kzg_commit_sharded(zkhip_comm *comm, const uint64_t *d_points_xy, const void *d_table, const uint8_t *d_points_inf,
                             size_t n_points, const uint64_t *d_scalars, size_t n_scalars, int require_equal_len,
                             uint64_t *h_out_xy, uint8_t *h_out_inf);

/* ---- KZG commit = multi-scalar multiplication over G1 -------------------------------------- */
/* MultilinearKZG::commitment (kzg/src/multilinear_kzg.rs:33-48; require_equal_len = 1 reproduces its
 * assert_eq!(srs.len(), evaluations.len())) and UnivariateKZG::commitment (kzg/src/univariate_kzg.rs:37-58;
 * require_equal_len = 0: the first n_scalars SRS points are used, and n_scalars > n_points is the
 * out-of-bounds panic at :53 -> ZKHIP_ERR_INDEX).
 *   d_points_xy  n_points x 12 : affine SRS points (x[6], y[6]), Montgomery Fq limbs, resident in HBM
 *   d_points_inf n_points bytes: 1 = point at infinity (NULL = none)
 *   d_scalars    n_scalars x 4 : polynomial evaluations / coefficients, Montgomery Fr limbs
 * Output (host): h_out_xy[12] affine commitment, *h_out_inf = 1 if it is the identity.  Parity with the
 * reference is on these affine coordinates (Jacobian X,Y,Z are algorithm dependent). */
int zkhip_kzg_commit(zkhip_ctx *ctx, const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                     const uint64_t *d_scalars, size_t n_scalars, int require_equal_len, uint64_t *h_out_xy,
                     uint8_t *h_out_inf);
/* Shifted-SRS table: d_table[w * n + i] = 2^(first bit of window w) * point i for the W digit windows of a 256-bit scalar, in the
 * kernels' internal layout (zkhip_srs_table_bytes(n) = W * 128 * n bytes; at 2^20 points W = 13 windows of 20 / 19 bits: 1.6 GiB -- HBM
 * is what this part has; smaller SRS get windows WIDER than their size, 2^12 points: 18 windows of 15 / 14 bits; commits of at most 2^12 scalars against a
 * table take a short path without sort or buckets -- one plain sum per digit bit, two launches: 0.42 ms at 2^8 against 0.65).  With it the digits
 * of all windows fall into ONE bucket set: at 2^20 13 n bucket additions instead of 16 n and a single bucket reduction.  The table
 * depends on the SRS only (built once, ~70 ms at 2^20); zkhip_kzg_commit_table then has the semantics of zkhip_kzg_commit (same group
 * element, same errors).  n * W must stay below 2^31.  A table begins with a 128-byte HEADER (magic, kind, n_points, window widths,
 * entries; included in zkhip_srs_table_bytes / zkhip_srs_level_tables_bytes): its layout is a function of n_points and of the tuning
 * variable ZKHIP_LEVEL_TABLE_DELTA (read once per process), and every entry point that takes a table compares the header with the
 * geometry it is about to address the table with -- a table built for another size, of the other kind, by a process that ran with
 * another value, or a buffer that is no table -> ZKHIP_ERR_ARG instead of a wrong commitment.  (One 128-byte read the first time the
 * process sees a (table, size) pair, remembered until a table is built at that address again or the owner RELEASES it:
 * zkhip_table_release(table) before the buffer is freed or reused -- the caller's allocator may hand the same address to something
 * that is no table, and a remembered address is not read again; zkhip_free does it for a buffer of zkhip_malloc's.) */
size_t zkhip_srs_table_bytes(size_t n_points);
int zkhip_table_release(zkhip_ctx *ctx, const void *d_table);
/* Content check for a host-side cache of what is derived from an SRS (the table above, the folded levels): the first two and the last
 * two points with their infinity flags, h_out[52] = 4 x (12 coordinate words, flag) -- one small launch, one copy, ~20 us.  A wrapper
 * that keeps such caches compares it before every use to catch writes into the SRS buffers that its own bookkeeping cannot see. */
int zkhip_srs_fingerprint(zkhip_ctx *ctx, const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                          uint64_t *h_out);
int zkhip_srs_precompute(zkhip_ctx *ctx, const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                         void *d_table);
int zkhip_kzg_commit_table(zkhip_ctx *ctx, const void *d_table, const uint8_t *d_points_inf, size_t n_points,
                           const uint64_t *d_scalars, size_t n_scalars, int require_equal_len, uint64_t *h_out_xy,
                           uint8_t *h_out_inf);
/* The same commitments, in flight: begin enqueues everything (on a stream of the context's own, ordered behind the caller's
 * stream) and returns a ticket; end waits for that commit, runs its host epilogue and delivers the result (all-NULL outputs
 * abandon it).  Up to THREE commits may be in flight: a commit is a throughput-bound bucket accumulation followed by
 * latency-bound reductions and a host epilogue, and back to back the latter hide behind the next commit's accumulation
 * (the reference commits one polynomial at a time; a prover with several polynomials to commit -- plonk's wires, every round
 * of an opening -- issues them this way).  Exactly one of d_points_xy / d_table (zkhip_srs_precompute) is given.  While a
 * commit is in flight the context's workspace is lent (other entry points that need it return ZKHIP_ERR_BUSY), and so is a
 * fourth begin, or a later one larger than the first. */
int zkhip_kzg_commit_begin(zkhip_ctx *ctx, const uint64_t *d_points_xy, const void *d_table, const uint8_t *d_points_inf,
                           size_t n_points, const uint64_t *d_scalars, size_t n_scalars, int require_equal_len,
                           uint32_t *ticket);
int zkhip_kzg_commit_end(zkhip_ctx *ctx, uint32_t ticket, uint64_t *h_out_xy, uint8_t *h_out_inf);
/* Several independent commitments in one pass of every kernel: problem j commits d_scalars[h_offsets[j] .. h_offsets[j+1])
 * against d_points_xy[same range] (n_problems <= 64; no reference counterpart -- the reference commits one polynomial at
 * a time; MultilinearKZG::open commits ALL its rounds this way, and a caller that commits many polynomials of any sizes
 * should too: a single small commit is latency bound at ~1 ms, and every problem of a batch gets window widths of its own, so
 * problems of 1 and of 2^19 entries share a pass without slowing each other).  Outputs: h_out_xy[12 j], h_out_inf[j]. */
int zkhip_kzg_commit_batch(zkhip_ctx *ctx, const uint64_t *d_points_xy, const uint8_t *d_points_inf,
                           const uint64_t *d_scalars, const size_t *h_offsets, uint32_t n_problems, uint64_t *h_out_xy,
                           uint8_t *h_out_inf);
/* Diagnostics (host only, no GPU): the geometry the batched commit above would use for these problem sizes -- how every problem's 256
 * scalar bits are cut into digit windows and how many buckets, sort partitions and reduction workgroups the pass has.
 *   h_win_first[n_problems + 1]: problem j owns the windows [h_win_first[j], h_win_first[j+1]);  h_win_bits[<= 2048]: their widths;
 *   h_totals[8] = { windows, bucket sets, buckets, sort partitions, (set, term) points, row/column workgroups, term workgroups,
 *                   heavy-bucket threshold }.
 * Returns ZKHIP_ERR_SHAPE when no window widths fit the sort's partitions (more than 64 problems' worth). */
int zkhip_msm_geometry_info(const size_t *h_offsets, uint32_t n_problems, uint16_t *h_win_first, uint8_t *h_win_bits,
                            uint32_t *h_totals);
/* SRS generation on the device (G1 side; the G2 powers are only used by the pairing verifier, out of scope).
 *   multilinear: TrustedSetup::generate_powers_of_tau_in_g1 (kzg/src/trusted_setup.rs:25-35):
 *                point i = G * prod_j (bit_j(i) ? tau_j : 1 - tau_j), hypercube bits MSB first; 2^n_vars points.
 *   univariate:  UnivariateKZG::generate_srs (kzg/src/univariate_kzg.rs:18-35): point i = G * tau^i, i = 0..=max_degree.
 * Outputs are DEVICE arrays in the layout zkhip_kzg_commit consumes: d_out_xy[n*12], d_out_inf[n]. */
int zkhip_srs_multilinear_g1(zkhip_ctx *ctx, const uint64_t *h_tau, uint32_t n_vars, uint64_t *d_out_xy,
                             uint8_t *d_out_inf);
int zkhip_srs_univariate_g1(zkhip_ctx *ctx, const uint64_t *h_tau, size_t max_degree, uint64_t *d_out_xy,
                            uint8_t *d_out_inf);
/* MultilinearKZGInterface::open (kzg/src/multilinear_kzg.rs:50-88): evaluation of the table at h_points plus one
 * G1 proof per variable.  Round i takes the quotient q_i = f_i(1, .) - f_i(0, .) (kzg/src/utils.rs:12-17), blows it up
 * to all n variables (add_to_front, evaluation_form.rs:86-96: the quotient repeated 2^(i+1) times) and commits it
 * against the whole SRS, then continues with the remainder f_{i+1} = f_i(z_i, .) (utils.rs:5-10).  A commitment to a
 * repeated table is sum_j q_i[j] * S_i[j] with S_i[j] = sum_rep SRS[rep * |q_i| + j]; the S_i ("folded SRS", n - 1
 * points in all) depend on the SRS only:
 *   zkhip_srs_fold_levels : d_out_xy[(n-1)*12], d_out_inf[n-1] <- S_0 (n/2 points), S_1 (n/4), ..., S_{nv-1} (1);
 *   zkhip_kzg_open        : d_folded_* = that array, or NULL to derive it inside the call;
 *   zkhip_srs_level_tables: shifted tables 2^(c w) * S_i[j] of the folded levels of at most 2^19 points, each level with a window
 *                           width c of its own (~log2 of its size), zkhip_srs_level_tables_bytes(n) bytes in all (1.9 GiB at 2^20):
 *                           built once per SRS like zkhip_srs_precompute's table; with them every round's commitment needs
 *                           ceil(256 / c) instead of ~ceil(256 / (c - 4)) bucket additions per point, ONE bucket set and a
 *                           host epilogue of ~20 instead of 255 doublings;
 *   zkhip_kzg_open_tables : zkhip_kzg_open with those tables (d_level_tables; NULL = zkhip_kzg_open; needs d_folded_inf).  Openings of at
 *                           most 2^12 entries take every round's quotient commit on the short path of zkhip_kzg_commit_table (one plain
 *                           sum per digit bit, two launches for all rounds); zkhip_srs_level_tables builds such an SRS's tables in a few ms.
 * Outputs (host): h_evaluation[4]; h_proofs_xy[n_vars*12], h_proofs_inf[n_vars] (affine, as zkhip_kzg_commit).
 * Shape errors as in the reference: n_points != n (multilinear_kzg.rs:36-41), n_eval_points != n_vars
 * (evaluation_form.rs:163-167), n_vars < 2 (`variable_index - 1` underflows at :73) -> ZKHIP_ERR_SHAPE. */
int zkhip_srs_fold_levels(zkhip_ctx *ctx, const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                          uint64_t *d_out_xy, uint8_t *d_out_inf);
int zkhip_kzg_open(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_points, size_t n_eval_points,
                   const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                   const uint64_t *d_folded_xy, const uint8_t *d_folded_inf, uint64_t *h_evaluation,
                   uint64_t *h_proofs_xy, uint8_t *h_proofs_inf);
size_t zkhip_srs_level_tables_bytes(size_t n_points);
int zkhip_srs_level_tables(zkhip_ctx *ctx, const uint64_t *d_folded_xy, const uint8_t *d_folded_inf, size_t n_points,
                           void *d_tables);
int zkhip_kzg_open_tables(zkhip_ctx *ctx, const uint64_t *d_evals, size_t n, const uint64_t *h_points, size_t n_eval_points,
                          const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                          const uint64_t *d_folded_xy, const uint8_t *d_folded_inf, const void *d_level_tables,
                          uint64_t *h_evaluation, uint64_t *h_proofs_xy, uint8_t *h_proofs_inf);
/* UnivariateKZGInterface::open (kzg/src/univariate_kzg.rs:60-81): evaluation = poly(z) (dense_univariate.rs:184-196),
 * proof = commitment to the quotient of (poly - z) / (x - z) (divide_with_q_and_r, dense_univariate.rs:88-124) against
 * the first n_coeffs - 1 SRS points.  Evaluation and quotient come from one Horner suffix scan on the device.
 * n_coeffs - 1 > n_points is the index panic at :75 -> ZKHIP_ERR_INDEX (checked on the coefficient count as given;
 * the reference would first drop zero leading coefficients).  Outputs (host): h_evaluation[4], h_proof_xy[12], *h_proof_inf. */
int zkhip_univariate_kzg_open(zkhip_ctx *ctx, const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *h_z,
                              const uint64_t *d_points_xy, const uint8_t *d_points_inf, size_t n_points,
                              uint64_t *h_evaluation, uint64_t *h_proof_xy, uint8_t *h_proof_inf);
/* DenseUnivariatePolynomial::evaluate (dense_univariate.rs:184-196: sum c_i z^i; here the Horner suffix scan's V_0) and
 * ::degree (:199-207: index of the last non-zero coefficient, 0 for none or an empty vector) of device coefficients. */
int zkhip_dense_evaluate(zkhip_ctx *ctx, const uint64_t *d_coeffs, size_t n_coeffs, const uint64_t *h_z, uint64_t *h_out);
int zkhip_dense_degree(zkhip_ctx *ctx, const uint64_t *d_coeffs, size_t n_coeffs, size_t *h_degree);
/* Sum of n affine points given on the host (combining per-GPU partial commitments after an all-gather). */
int zkhip_g1_sum_affine(const uint64_t *h_points_xy, const uint8_t *h_points_inf, size_t n, uint64_t *h_out_xy,
                        uint8_t *h_out_inf);

/* ---- NTT / Domain / polynomial product (polynomial/src/univariate/) ------------------------- */
/* Domain::new (domain.rs:31-48) for a power-of-two size: generator = F::get_root_of_unity(size), its inverse,
 * and size^-1, all Montgomery Fr (host outputs, 4 limbs each).  size > 2^32 -> ZKHIP_ERR_SHAPE (unwrap panic). */
int zkhip_domain_params(uint64_t size, uint64_t *h_generator, uint64_t *h_generator_inv, uint64_t *h_size_inv);
/* Domain::fft_internal / ifft_internal (domain.rs:120-133) = serial_fft (polynomial/src/utils.rs:281-315) with
 * omega, resp. omega^-1 followed by the scaling with size^-1; in place on d_data[2^log_n], natural order in/out. */
int zkhip_ntt(zkhip_ctx *ctx, uint64_t *d_data, uint32_t log_n, int inverse);
/* Domain::fft / ifft as the reference calls them (domain.rs:108-118: clone the input, resize it to the domain size with
 * zeros, transform): d_src holds n_src <= 2^log_n values and is not modified, d_dst receives the 2^log_n results.  The zero
 * padding happens inside the first pass (no staging copy). */
int zkhip_domain_transform(zkhip_ctx *ctx, const uint64_t *d_src, size_t n_src, uint64_t *d_dst, uint32_t log_n, int inverse);
int zkhip_pointwise_mul(zkhip_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, size_t n, uint64_t *d_out);
/* UnivariateEval::multiply (evaluation.rs:59-86): d_out[na + nb - 1] = coefficients of a * b via three transforms. */
int zkhip_univariate_multiply(zkhip_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb,
                              uint64_t *d_out);

#ifdef __cplusplus
}
#endif
#endif
