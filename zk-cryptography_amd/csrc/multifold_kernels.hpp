// multifold_kernels.hpp -- the basic sumcheck prover restructured around k-variable folds (gfx950).
//
// Same observable behaviour as Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61): identical sum, round
// polynomials and challenges.  The restructuring uses two exact identities of the multilinear fold
// (polynomial/src/multilinear/evaluation_form.rs:123-141) when variable 0 (the most significant index bit) is
// folded first, as every sumcheck round does (sumcheck.rs:50):
//   (1) block sums commute with the fold: if B[b] = sum_j T[b*m + j] (2^k blocks of m entries), then the block
//       sums of fold(T, r) are fold(B, r).  The round polynomial (lower-half sum, upper-half sum) of each of the
//       next k rounds is therefore the round polynomial of the 2^k-entry table B: k rounds of transcript run on
//       B alone, inside one workgroup, BEFORE the big table is touched again;
//   (2) k folds collapse into one pass: T_k[j] = sum_b eq_b(r_1..r_k) T[b*m + j], eq_b = prod_i (b_i ? r_i : 1-r_i).
// A 2^24-entry prover thus streams the table twice (block sums, then one 8-variable fold that also emits the
// block sums of its output) instead of ~5.3 times, and issues ~8 kernels instead of ~30.
// Field arithmetic is exact, so the values equal the round-by-round ones bit for bit.
//
// The k-variable fold accumulates sum_b w_b * T_b as an unreduced 17-limb integer (15 column accumulators of
// 96 bits: 64 mads + 64 carry adds per term, half of a Montgomery product) and reduces once per output with
// a 9-word REDC; the weights carry an extra factor 2^32 that the ninth word removes.
#pragma once
#include "sumcheck_kernels.hpp"
#include "wide_acc.hpp"

namespace zk {

constexpr int MF_MAX_LOGK = 8;   // variables per fold in the single-GPU plan
constexpr int MF_CAP_LOGK = 9;   // what the kernels can take; the sharded plan uses 9 where it saves an exchange (2^27 over 8 GPUs)

// per-workgroup sum of a contiguous chunk of `chunk` entries (chunk a power of two, >= MLE_BLOCK)
static __global__ __launch_bounds__(MLE_BLOCK) void chunk_sums_kernel(const uint64_t* __restrict__ in, uint32_t chunk,
                                                                      uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const uint64_t* base = in + 4 * (size_t)blockIdx.x * chunk;
    Fr s = Fr::zero();
    for (uint32_t j = threadIdx.x; j < chunk; j += 4 * MLE_BLOCK) {
        Fr v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (j + u * MLE_BLOCK < chunk) ? load_fr(base, j + u * MLE_BLOCK) : Fr::zero();
#pragma unroll
        for (int u = 0; u < 4; ++u) s = s + v[u];
    }
    s = block_reduce_fr(s, red);
    if (threadIdx.x == 0) store_fr(partials, blockIdx.x, s);
}

// out[b] = sum of the group-th consecutive partials, b < n_blocks; out[n_blocks] = total
static __global__ __launch_bounds__(MLE_BLOCK) void group_sums_kernel(const uint64_t* __restrict__ partials, uint32_t group,
                                                                      uint32_t n_blocks, uint64_t* __restrict__ out) {
    __shared__ Fr red[MLE_BLOCK / 64];
    Fr tot = Fr::zero();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += MLE_BLOCK) {
        Fr s = Fr::zero();
        for (uint32_t g = 0; g < group; ++g) s = s + load_fr(partials, (size_t)b * group + g);
        store_fr(out, b, s);
        tot = tot + s;
    }
    tot = block_reduce_fr(tot, red);
    if (threadIdx.x == 0) store_fr(out, n_blocks, tot);
}

// ---- fine block sums for the overlapped plan -------------------------------------------------------------
// sums[c] = sum of the c-th run of FINE_CHUNK consecutive entries.  A wave owns two adjacent runs (8 loads of 2 KiB in flight per
// wave); no loop, the grid covers the table (n / (8 FINE_CHUNK) workgroups).
// Probes (tools/ubench_fine.hip and in-situ A/Bs, 2^24 entries, same box; profiles/r04/NOTES.md): the loads alone take
// 82.2-82.5 us non-temporal, 85.9 plain; this kernel 90.3-91.0.  Its two runs half a table apart + non-temporal loads: 87.1-87.7 in
// isolation, but in the prover the k-variable fold behind it then ran 95 instead of 84 us (it had been finding the tail of THIS pass in
// the 256 MiB Infinity Cache) and the step did not move.  Four runs per wave, waves permuted by strides of 2^3..2^12 runs: 88-96 us.
// Successive passes in alternating directions (each starting in the half the previous one left in that cache): step 0.3277 / 0.3307
// against 0.3297 / 0.3321 ms -- inside the noise.  All dropped.  The ceiling of a read-only stream on this part is the same from
// anywhere: a table that FITS the Infinity Cache (64 / 128 / 256 MiB swept twice) is read at 6.4-7.2 TB/s.
constexpr int FINE_CHUNK = 256;
static __global__ __launch_bounds__(MLE_BLOCK) void fine_sums_kernel(const uint64_t* __restrict__ in, size_t n_chunks,
                                                                     uint64_t* __restrict__ sums) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t c0 = ((size_t)blockIdx.x * (MLE_BLOCK / 64) + wave) * 2;
    if (c0 >= n_chunks) return;
    const bool two = c0 + 1 < n_chunks;
    const uint64_t* base = in + 4 * (c0 * FINE_CHUNK);
    Fr v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = load_fr(base, lane + 64 * u);
#pragma unroll
    for (int u = 4; u < 8; ++u) v[u] = two ? load_fr(base, lane + 64 * u) : Fr::zero();
    Fr a = (v[0] + v[1]) + (v[2] + v[3]);
    Fr b = (v[4] + v[5]) + (v[6] + v[7]);
    wave_reduce_fr2(a, b);              // DPP / permlane moves: no LDS traffic beside the loads
    if (lane == 0) {
        store_fr(sums, c0, a);
        if (two) store_fr(sums, c0 + 1, b);
    }
}

// One workgroup per output b: the sum of in[b*group .. (b+1)*group) (the coarse block sums from the fine ones), written as
// a Montgomery residue to out_mont[b] and / or as a canonical integer -- what the serial kernel's sum tree holds -- to
// out_canon[b] (either may be null)
static __global__ __launch_bounds__(MLE_BLOCK) void group_sums_wg_kernel(const uint64_t* __restrict__ in, uint32_t group,
                                                                         uint64_t* __restrict__ out_mont, uint64_t* __restrict__ out_canon) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const uint64_t* base = in + 4 * (size_t)blockIdx.x * group;
    Fr s = Fr::zero();
    for (uint32_t j = threadIdx.x; j < group; j += 4 * MLE_BLOCK) {
        Fr v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (j + u * MLE_BLOCK < group) ? load_fr(base, j + u * MLE_BLOCK) : Fr::zero();
#pragma unroll
        for (int u = 0; u < 4; ++u) s = s + v[u];
    }
    s = block_reduce_fr(s, red);
    if (threadIdx.x == 0) {
        if (out_mont) store_fr(out_mont, blockIdx.x, s);
        if (out_canon) store_fr(out_canon, blockIdx.x, fr_from_mont_outlined(s));
    }
}

constexpr int TREE_MAX_LOG = 10;   // the serial kernel keeps tables of up to 2^10 entries (and their sum trees) in LDS

struct SmallArgs {
    const uint64_t* src;    // mode 0: the table itself (2^log_n entries); mode 1: partial sums, `group` per entry
    uint32_t group;         // mode 1: partials per table entry (0 = mode 0)
    uint32_t stride;        // mode 1: 0 = entry j owns partials j*group .. +group; else partial g of entry j is src[g*stride + j]
    uint32_t canon;         // the source values are canonical integers already (blockfold_kernel / group_sums_wg_kernel<true>)
    uint32_t log_n;         // working table has 2^log_n entries (<= 2^TREE_MAX_LOG)
    uint32_t n_rounds;      // rounds to run, <= log_n
    uint32_t round0;        // index of the first round run here
    uint32_t first;         // 0 continue; 1 start the transcript, sum = lo + hi; 2 sum = claimed; 3 sum = *d_claimed
    FrArg claimed;
    const uint64_t* d_claimed;
    uint64_t* weights_out;  // nullable: 2^n_rounds fold weights eq_b(r) * 2^32 (Montgomery form)
    uint64_t* final_out;    // nullable: the table left after n_rounds folds
    // The LAST kernel of a proof delivers it: host_delta != 0 is the distance (in u64) from the device span [state .. round polynomials] to
    // its pinned host mirror -- the rounds of earlier kernels are copied there at entry, this kernel's own outputs are stored twice -- so
    // no hipMemcpyAsync (8 us with its launch gap) follows the last round.  The host waits for the kernel's completion event.
    long long host_delta;
};

constexpr int SMALL_BLOCK = 512;   // 8 waves: the transcript wave, two schedule waves, five waves for the trees and weights

// n_rounds sumcheck rounds (half sums -> transcript -> challenge -> fold, sumcheck.rs:40-51) of a small table.
//
// The table sits in LDS together with its SUM TREE (node q = 2^l + b holds the sum of block b when the table is
// cut into 2^l blocks; the leaves are the table).  Folding the table at variable 0 folds every level the same way:
//     new[q] = old[q + p] + r * (old[q + 2p] - old[q + p]),  p = largest power of two <= q,
// so after a fold the next round polynomial (lo, hi) = new[2], new[3] is available WITHOUT a reduction, and every
// node is independent work.
// The tree is kept in CANONICAL (non-Montgomery) form: the transcript absorbs canonical bytes, so (lo, hi) need no
// conversion, and a product of a Montgomery-form challenge with a canonical difference is again canonical.
// Wave 0 runs nothing but the transcript chain plus ONE product per round (challenge_canonical * E, where
// E = to_mont(hi' - lo') was prepared a round earlier); waves 1 and 2 prepare the message schedules of the next round's
// two SHA-256 blocks and nothing else (with tree work on top they were the last to reach the barrier); waves 3-7 convert
// the challenge, fold the rest of the tree and the k-variable fold weights, and write the outputs (Montgomery form).
// One barrier per round; the trees, E and the challenge slot are double-buffered.  (Stamps of a 2^24 prove,
// tools/diag_small.py: a round = 2 x 2.1 us of state rounds + 0.9 us for the product + 0.3 us to publish.)
static __global__ __launch_bounds__(SMALL_BLOCK) void sumcheck_small_kernel(SmallArgs a, SumcheckDev* st,
                                                                            uint64_t* __restrict__ round_polys,
                                                                            uint64_t* __restrict__ challenges) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    __builtin_amdgcn_s_setprio(3);                      // the chip's critical path: win every issue arbitration
    ZK_STAMP_AT(0, 40 + a.round0, 0);                   // diagnostics: kernel entry
    const uint32_t n = 1u << a.log_n;
    Fr* tree0 = reinterpret_cast<Fr*>(zk_dyn_lds);      // 2n nodes, canonical values
    Fr* tree1 = tree0 + 2 * n;                          // n nodes: the first tree built here is already a folded one
    // weights (Montgomery), ping / pong: 2^(it+1) of them after round it, so the array that receives the last round's
    // holds 2^n_rounds and the other one half of that; no region at all when none are asked for
    const uint32_t w_all = a.weights_out ? (1u << a.n_rounds) : 0u;
    Fr* w0 = tree1 + n;
    Fr* w1 = w0 + ((a.n_rounds & 1) ? w_all / 2 : w_all);
    const bool grouped = a.group != 0 && a.stride == 0;
    Fr* scratch = w0 + w_all + w_all / 2 + 1;           // SMALL_BLOCK entries, only in the grouped mode
    Fr* e_sh = scratch + (grouped ? SMALL_BLOCK : 0);   // 2 x 2: to_mont(level-2 differences) of tree0 / tree1
    Fr* r_sh = e_sh + 4;                                // 2: canonical challenge, double-buffered
    if (a.host_delta && a.round0) {   // what earlier kernels of this proof recorded: sum, round polynomials and challenges of rounds < round0
        for (uint32_t i = threadIdx.x; i < 12 * a.round0 + 4; i += SMALL_BLOCK) {
            const uint64_t* p = i < 4 ? st->sum + i : i < 4 + 8 * a.round0 ? round_polys + (i - 4) : challenges + (i - 4 - 8 * a.round0);
            *const_cast<uint64_t*>(p + a.host_delta) = *p;
        }
    }
    // ---- leaves (sums are taken as they come -- the conversion to canonical integers is linear -- then converted once)
    if (a.group == 0) {
        for (uint32_t j = threadIdx.x; j < n; j += SMALL_BLOCK) {
            const Fr v = load_fr(a.src, j);
            tree0[n + j] = a.canon ? v : fr_from_mont_outlined(v);
        }
    } else if (a.stride != 0) {   // partial tables (blockfold_kernel's term ranges, or all-gathered per-rank block sums in rank order)
        for (uint32_t j = threadIdx.x; j < n; j += SMALL_BLOCK) {
            Fr s = load_fr(a.src, j);
            for (uint32_t g = 1; g < a.group; g += 8) {      // up to 8 loads in flight (the serial kernel's prologue is latency)
                Fr v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (g + u < a.group) v[u] = load_fr(a.src, (size_t)(g + u) * a.stride + j);
#pragma unroll
                for (int u = 0; u < 8; ++u) if (g + u < a.group) s = s + v[u];
            }
            tree0[n + j] = a.canon ? s : fr_from_mont_outlined(s);
        }
    } else {
        const uint32_t total = n * a.group;
        const uint32_t run = total > (uint32_t)SMALL_BLOCK ? total / SMALL_BLOCK : 1;   // consecutive partials per thread
        if (threadIdx.x * run < total) {
            Fr s = Fr::zero();
            for (uint32_t g = 0; g < run; ++g) s = s + load_fr(a.src, (size_t)threadIdx.x * run + g);
            scratch[threadIdx.x] = s;
        }
        __syncthreads();
        const uint32_t tpe = a.group / run;   // scratch slots per entry (run <= group because n <= SMALL_BLOCK in this mode)
        for (uint32_t j = threadIdx.x; j < n; j += SMALL_BLOCK) {
            Fr s = scratch[j * tpe];
            for (uint32_t g = 1; g < tpe; ++g) s = s + scratch[j * tpe + g];
            tree0[n + j] = a.canon ? s : fr_from_mont_outlined(s);
        }
    }
    if (threadIdx.x == 0 && a.weights_out) {
        Fr one32;   // Montgomery form of 2^32: the k-variable fold reduces with 9 words instead of 8
        constexpr uint32_t c[8] = {0xcaaf6b13u, 0x355094eau, 0x69a568efu, 0xf6b10cb3u, 0x40cc3869u, 0xe2c926a6u, 0xed269aadu, 0x736a6d3bu};
#pragma unroll
        for (int i = 0; i < 8; ++i) one32.l[i] = c[i];
        w0[0] = one32;
    }
    __syncthreads();
    // ---- inner levels, bottom up, three levels per barrier: a lane sums the subtree under 8 nodes of the level below
    for (uint32_t lvl = a.log_n; lvl > 1;) {
        const uint32_t step = lvl - 1 < 3 ? lvl - 1 : 3;
        const uint32_t groups = 1u << (lvl - step);
        for (uint32_t g = threadIdx.x; g < groups; g += SMALL_BLOCK) {
            Fr v[8];
            const uint32_t cnt0 = 1u << step;
#pragma unroll
            for (int i = 0; i < 8; ++i) if ((uint32_t)i < cnt0) v[i] = tree0[(1u << lvl) + g * cnt0 + i];
#pragma unroll
            for (int sft = 1; sft <= 3; ++sft) {
                if ((uint32_t)sft <= step) {
                    const uint32_t cnt = cnt0 >> sft;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if ((uint32_t)i < cnt) {
                            v[i] = v[2 * i] + v[2 * i + 1];
                            tree0[(1u << (lvl - sft)) + g * cnt + i] = v[i];
                        }
                    }
                }
            }
        }
        __syncthreads();
        lvl -= step;
    }
    // ---- rounds.  Every round hashes the same two-block message from the initial hash value:
    //   block 1 = prefix (the previous challenge's digest; in the first round of a proof the claimed sum) || lo
    //   block 2 = hi || 0x80 padding || length (96 bytes)
    // (commit(lo), commit(hi), challenge() of fiat_shamir.rs:17-25 after a challenge() has re-seeded the hasher).
    // Wave 0 runs the rounds of the two compressions; wave 1 prepares the schedule of block 1, wave 2 that of block 2
    // (sha256_schedule_to_lds), both from their own copy of lo / hi, one round ahead of the hash.
    const uint32_t wave = threadIdx.x >> 6;
    const bool wave0 = wave == 0;
    uint32_t* kw1 = reinterpret_cast<uint32_t*>(r_sh + 2);      // 64 words
    uint32_t* kw2 = kw1 + 64;                                   // 64 words
    uint32_t* dig_sh = kw2 + 64;                                // 2 x 8: raw digest of the last challenge, double-buffered
    uint32_t* flags = dig_sh + 16;                              // [0]: kw1 progress, [1]: kw2 progress
    if (threadIdx.x >= 64 && threadIdx.x < 66 && a.log_n >= 2)
        e_sh[threadIdx.x - 64] = (tree0[6 + threadIdx.x - 64] - tree0[4 + threadIdx.x - 64]).to_mont();
    if (threadIdx.x == 0) {
        flags[0] = 0; flags[1] = 0;
        // prefix of the first round run here
        uint32_t pre[8];
        if (a.first) {
            const Fr sum_c = (a.first == 2) ? fr_from_mont_outlined(fr_from_arg(a.claimed))
                           : (a.first == 3) ? fr_from_mont_outlined(load_fr(a.d_claimed, 0)) : tree0[2] + tree0[3];
#pragma unroll
            for (int i = 0; i < 8; ++i) pre[i] = sum_c.l[7 - i];
        } else {   // a kernel always stops after a challenge(): initial hash value, the digest pending
#pragma unroll
            for (int i = 0; i < 8; ++i) pre[i] = st->transcript.buf[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) dig_sh[8 + i] = pre[i];     // slot (it - 1) & 1 for it = 0
    }
    __syncthreads();
    Fr lo = tree0[2], hi = tree0[3];     // canonical
    uint32_t depth = a.log_n, round = a.round0, n_w = 1;
    ZK_STAMP_AT(0, 40 + a.round0, 1);                   // prologue done
    // schedules of round 0 (waves 1 and 2)
    // (a word per lane: lane i of every row holds word i of the block -- sha256_schedule_rows_to_lds)
    const uint32_t word = threadIdx.x & 15;
    auto schedule_block1 = [&](const uint32_t* prefix, const Fr& lo_c, uint32_t it) {
        const uint32_t w = word < 8 ? prefix[word] : fr_limb_by_lane(lo_c, 15 - word);
        sha256_schedule_rows_to_lds(w, kw1, flags, 4 * it, 1, threadIdx.x < 64 + 16);
    };
    auto schedule_block2 = [&](const Fr& hi_c, uint32_t it) {
        const uint32_t w = word < 8 ? fr_limb_by_lane(hi_c, 7 - word) : (word == 8 ? 0x80000000u : word == 15 ? 96u * 8u : 0u);
        sha256_schedule_rows_to_lds(w, kw2, flags + 1, 4 * it, 0, threadIdx.x < 128 + 16);
    };
    if (a.n_rounds) {
        if (wave == 1) schedule_block1(dig_sh + 8, lo, 0);
        if (wave == 2) schedule_block2(hi, 0);
    }
    uint32_t digest[8];
    // Waves 3, 5, 6, 7 fold the trees and the weights.  Wave 4 sits on the transcript wave's SIMD (waves go round the four SIMDs),
    // where every instruction it issues is one the hash waits for: it only records the round's outputs, which nobody in this
    // kernel reads, and at the lowest priority.
    constexpr uint32_t FIRST_HELPER = 192, N_HELPERS = SMALL_BLOCK - FIRST_HELPER - 64, OUT_WAVE = 4;
    if (wave == OUT_WAVE) __builtin_amdgcn_s_setprio(0);
    for (uint32_t it = 0; it < a.n_rounds; ++it) {
        Fr* told = (it & 1) ? tree1 : tree0;
        Fr* tnew = (it & 1) ? tree0 : tree1;
        Fr* wold = (it & 1) ? w1 : w0;
        Fr* wnew = (it & 1) ? w0 : w1;
        Fr* eold = e_sh + 2 * (it & 1);
        Fr* enew = e_sh + 2 * ((it & 1) ^ 1);
        const bool absorb_sum = a.first && it == 0;
        if (wave0) {
            ZK_STAMP_AT(0, round, 0);
            uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
            uint32_t blk[16];
            const uint32_t* pre = dig_sh + 8 * ((it + 1) & 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) { blk[i] = pre[i]; blk[8 + i] = lo.l[7 - i]; }
            ShaSplit sp;                                              // the state rounds on six lanes (transcript.hpp)
            sp.init();
            uint32_t hs[4];
            sp.split(h, hs);
            sha256_compress_kw_split(sp, hs, blk, kw1, flags, 4 * it);          // uni_poly.to_bytes()  sumcheck.rs:42
            ZK_STAMP_AT(0, round, 1);
            sha256_compress_kw_split(sp, hs, nullptr, kw2, flags + 1, 4 * it);  // challenge()  :46
            sp.join(hs, h);
            ZK_STAMP_AT(0, round, 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) digest[i] = h[i];
            Fr c;                                                    // from_be_bytes_mod_order, still canonical
#pragma unroll
            for (int i = 0; i < 8; ++i) c.l[i] = h[7 - i];
            c.reduce_once();
            c.reduce_once();
            if (threadIdx.x == 0) {
                r_sh[it & 1] = c;
#pragma unroll
                for (int i = 0; i < 8; ++i) dig_sh[8 * (it & 1) + i] = h[i];
            }
        }
        ZK_STAMP_AT(0, round, 3);
        ZK_STAMP_AT(SMALL_BLOCK - 1, round, 6);      // the last helper arrives at the barrier
        ZK_STAMP_AT(64, round, 7);                   // the first schedule wave arrives
        __syncthreads();   // challenge and digest published; tree `told` and `eold` complete
        ZK_STAMP_AT(0, round, 4);
        const Fr c = r_sh[it & 1];
        if (wave0) {
            if (depth >= 2) {   // next round polynomial straight from level 2 of the old tree: lo' = t[4] + c * (t[6] - t[4])
                const uint32_t q = threadIdx.x & 1;
                Fr v = told[4 + q] + fr_mul_outlined(c, eold[q]);
                lo = shfl_fr(v, 0);
                hi = shfl_fr(v, 1);
            }
            ZK_STAMP_AT(0, round, 5);
        } else if (wave <= 2) {
            // schedules of the NEXT round (its hash is already waiting for them)
            if (it + 1 < a.n_rounds && depth >= 2) {
                const uint32_t q = wave - 1;                         // wave 1: lo', wave 2: hi'
                const Fr v = told[4 + q] + fr_mul_outlined(c, eold[q]);
                if (wave == 1) schedule_block1(dig_sh + 8 * (it & 1), v, it + 1);
                else schedule_block2(v, it + 1);
            }
        } else {
            const Fr r = fr_to_mont_outlined(c);       // every helper wave converts for itself (no extra sync)
            const uint32_t helper = wave == OUT_WAVE ? N_HELPERS : wave < OUT_WAVE ? threadIdx.x - FIRST_HELPER : threadIdx.x - FIRST_HELPER - 64;
            if (threadIdx.x == 64 * OUT_WAVE) {   // outputs of this round, in Montgomery form as the reference holds them
                Fr lo_m = told[2].to_mont(), hi_m = told[3].to_mont();
                if (absorb_sum) {
                    Fr sum_m = (a.first == 2) ? fr_from_arg(a.claimed) : (a.first == 3) ? load_fr(a.d_claimed, 0) : lo_m + hi_m;
                    store_fr(st->sum, 0, sum_m);
                    if (a.host_delta) store_fr(st->sum + a.host_delta, 0, sum_m);
                }
                store_fr(round_polys, 2 * (size_t)round, lo_m);
                store_fr(round_polys, 2 * (size_t)round + 1, hi_m);
                store_fr(challenges, round, r);
                if (a.host_delta) {
                    store_fr(round_polys + a.host_delta, 2 * (size_t)round, lo_m);
                    store_fr(round_polys + a.host_delta, 2 * (size_t)round + 1, hi_m);
                    store_fr(challenges + a.host_delta, round, r);
                }
            }
            const uint32_t nodes = 1u << (depth - 1);   // the new tree has nodes 1 .. 2*nodes - 1
            if (helper >= N_HELPERS - 2 && helper < N_HELPERS && depth >= 3) {
                // the last two lanes of the last helper wave: new level 2 pair (q, q+2) and the product wave 0 will need next round
                const uint32_t q = 4 + (helper - (N_HELPERS - 2));
                Fr va = told[q + 4] + r * (told[q + 8] - told[q + 4]);           // new[q],   p = 4
                Fr vb = told[q + 6] + r * (told[q + 10] - told[q + 6]);          // new[q+2], p = 4
                tnew[q] = va;
                tnew[q + 2] = vb;
                enew[helper - (N_HELPERS - 2)] = (vb - va).to_mont();
            }
            for (uint32_t q = 1 + helper; q < 2 * nodes && helper < N_HELPERS; q += N_HELPERS) {
                if (depth >= 3 && q >= 4 && q < 8) continue;                     // done above
                const uint32_t p = 1u << (31 - __builtin_clz(q));
                tnew[q] = told[q + p] + r * (told[q + 2 * p] - told[q + p]);
            }
            if (a.weights_out) {   // eq weights: the index gains the new variable as its least significant bit
                for (uint32_t b = helper; b < n_w && helper < N_HELPERS; b += N_HELPERS) {
                    Fr w1v = wold[b] * r;
                    wnew[2 * b + 1] = w1v;
                    wnew[2 * b] = wold[b] - w1v;
                }
            }
        }
        n_w <<= 1;
        --depth;
        ++round;
    }
    ZK_STAMP_AT(0, 40 + a.round0, 2);                   // rounds done
    __syncthreads();
    if (threadIdx.x == 0 && a.n_rounds) {   // what FiatShamirTranscript holds after a challenge(): fresh hasher + the digest
        Transcript tr;
        tr.init();
        tr.commit_words8(digest);
        tr.store(&st->transcript);
    }

    if (a.weights_out) {
        Fr* w = (a.n_rounds & 1) ? w1 : w0;
        for (uint32_t b = threadIdx.x; b < n_w; b += SMALL_BLOCK) store_fr(a.weights_out, b, w[b]);
    }
    if (a.final_out) {
        Fr* t = (a.n_rounds & 1) ? tree1 : tree0;
        const uint32_t cnt = 1u << depth;
        for (uint32_t j = threadIdx.x; j < cnt; j += SMALL_BLOCK) store_fr(a.final_out, j, t[cnt + j].to_mont());
    }
}
// dynamic LDS of the kernel above: trees (3 x 2^log_n), weights (1.5 x 2^weight_rounds; weight_rounds < 0: none), the
// grouped mode's scratch, the small shared slots
__host__ __device__ constexpr size_t small_lds_bytes(uint32_t log_n, int weight_rounds, bool grouped) {
    return ((size_t)3 * ((size_t)1 << log_n) + (weight_rounds >= 0 ? 3 * ((size_t)1 << weight_rounds) / 2 : 0u) + 1 +
            (grouped ? SMALL_BLOCK : 0) + 4 + 2) * 32 + (64 + 64 + 16 + 2) * 4;
}

// Fold weights of k known points (MultilinearTrait::evaluation folds variable 0 repeatedly, evaluation_form.rs:162-175):
// w[b] = 2^32 * prod_i (b_i ? r_i : 1 - r_i), b_1 = most significant bit of b.  One lane per weight.
static __global__ __launch_bounds__(MLE_BLOCK) void eq_weights_kernel(PtsArg pts, uint32_t first, uint32_t k,
                                                                      uint64_t* __restrict__ out) {
    const uint32_t b = blockIdx.x * MLE_BLOCK + threadIdx.x;
    if (b >= (1u << k)) return;
    Fr w;
    constexpr uint32_t c[8] = {0xcaaf6b13u, 0x355094eau, 0x69a568efu, 0xf6b10cb3u, 0x40cc3869u, 0xe2c926a6u, 0xed269aadu, 0x736a6d3bu};
#pragma unroll
    for (int i = 0; i < 8; ++i) w.l[i] = c[i];
    const Fr one = Fr::one();
    for (uint32_t i = 0; i < k; ++i) {
        Fr r = fr_from_pts(pts, first + i);
        if (!((b >> (k - 1 - i)) & 1)) r = one - r;
        w = w * r;
    }
    store_fr(out, b, w);
}

// The weight tables of an evaluation in one pass (zkhip_mle_evaluation), one lane per entry:
//   w1[b], b < 2^k1       the fold weights of the first k1 points (with the factor 2^32, as above);
//   wa[x], x < 2^(k2 - s) and wb[y], y < 2^s: the plain eq tables (Montgomery form, no factor) of the next k2 - s and the last s
//   points -- the eq table of all k2 remaining points is their outer product, eq[x * 2^s + y] = wa[x] * wb[y], formed where it is
//   used (one product per output): at most max(k1, k2 - s, s) products one after another here instead of k2.
static __global__ __launch_bounds__(MLE_BLOCK) void eval_weights_kernel(PtsArg pts, uint32_t k1, uint32_t k2, uint32_t s, uint64_t* __restrict__ w1,
                                                                        uint64_t* __restrict__ wa, uint64_t* __restrict__ wb) {
    size_t b = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x;
    const size_t n1 = (size_t)1 << k1, na = (size_t)1 << (k2 - s), nb = (size_t)1 << s;
    uint32_t k, first;
    uint64_t* out;
    Fr w = Fr::one();
    if (b < n1) {
        k = k1; first = 0; out = w1;
        constexpr uint32_t c[8] = {0xcaaf6b13u, 0x355094eau, 0x69a568efu, 0xf6b10cb3u, 0x40cc3869u, 0xe2c926a6u, 0xed269aadu, 0x736a6d3bu};
#pragma unroll
        for (int i = 0; i < 8; ++i) w.l[i] = c[i];
    } else if (b < n1 + na) {
        b -= n1; k = k2 - s; first = k1; out = wa;
    } else if (b < n1 + na + nb) {
        b -= n1 + na; k = s; first = k1 + k2 - s; out = wb;
    } else {
        return;
    }
    const Fr one = Fr::one();
    for (uint32_t i = 0; i < k; ++i) {
        Fr r = fr_from_pts(pts, first + i);
        if (!((b >> (k - 1 - i)) & 1)) r = one - r;
        w = w * r;
    }
    store_fr(out, b, w);
}
// out[0] = sum of n records (the tiles' shares of an evaluation)
static __global__ __launch_bounds__(MLE_BLOCK) void sum_records_kernel(const uint64_t* __restrict__ records, uint32_t n, uint64_t* __restrict__ out) {
    __shared__ Fr red[MLE_BLOCK / 64];
    Fr s = Fr::zero();
    for (uint32_t i = threadIdx.x; i < n; i += 4 * MLE_BLOCK) {
        Fr v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + u * MLE_BLOCK < n) ? load_fr(records, i + u * MLE_BLOCK) : Fr::zero();
#pragma unroll
        for (int u = 0; u < 4; ++u) s = s + v[u];
    }
    s = block_reduce_fr(s, red);
    if (threadIdx.x == 0) store_fr(out, 0, s);
}

// ---- the k-variable fold -------------------------------------------------------------------------------
// out[j] = sum_{b < 2^k} w[b] * in[b*m + j], j < m.
// A wave covers G consecutive outputs x (64/G) term groups; a workgroup's S = blockDim/64 waves split the 2^k terms
// further, so one output is shared by S * 64/G lanes.  Every lane reduces its own unreduced partial sum (the
// weights' 2^32 factor makes each 9-word REDC a proper Montgomery residue, so partial results simply add); the
// groups of a wave combine by shuffles, the waves through LDS.  G = 64 streams big tables (2 KiB per wave-load);
// G = 16 keeps the chip busy when only a few hundred outputs are left.
// Also writes the workgroup's sum of outputs to partials[blockIdx.x] (block sums of the output table).
template <int G, int DEPTH = 4, bool NT = false>
static __global__ __launch_bounds__(1024) void multifold_kernel(const uint64_t* __restrict__ in, size_t m, uint32_t k,
                                                                const uint64_t* __restrict__ weights,
                                                                uint64_t* __restrict__ out,
                                                                uint64_t* __restrict__ partials) {
    constexpr int TG = 64 / G;                                // term groups inside a wave
    // dynamic LDS: 2^k weights, then the (waves - 1) x G partial outputs -- (32 << k) + 32 (waves - 1) G bytes.  LDS decides
    // how many workgroups share a CU, so it is sized exactly (a fixed 16 KiB for the weights cost the streaming fold 30 %).
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    Fr* w_lds = reinterpret_cast<Fr*>(zk_dyn_lds);
    Fr* part = w_lds + (1u << k);
    const uint32_t n_terms = 1u << k;
    for (uint32_t b = threadIdx.x; b < n_terms; b += blockDim.x) w_lds[b] = load_fr(weights, b);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t slice = wave * TG + lane / G;              // term slice of this lane
    const uint32_t per = n_terms / (n_waves * TG);            // host guarantees >= 1
    const uint32_t oj = lane % G;
    const size_t j = (size_t)blockIdx.x * G + oj;
    WideAcc acc;
    acc.clear();
    const uint32_t b0 = slice * per;
    const uint64_t* p = in + 4 * ((size_t)b0 * m + j);
    const size_t row = 4 * m;
    for (uint32_t t = 0; t < per; t += DEPTH) {
        Fr v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) if (t + u < per) v[u] = NT ? load_fr_nt(p + (size_t)(t + u) * row) : load_fr(p + (size_t)(t + u) * row, 0);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) if (t + u < per) acc.mac(w_lds[b0 + t + u], v[u]);
    }
    Fr o = wide_reduce(acc.lo, acc.hi);
#pragma unroll
    for (int d = 32; d >= G; d >>= 1) o = o + shfl_down_fr(o, d);   // lanes < G now hold the wave's sums
    if (wave != 0 && lane < G) part[(wave - 1) * G + lane] = o;
    __syncthreads();
    if (wave == 0) {
        if (lane < G) {
            for (uint32_t w2 = 1; w2 < n_waves; ++w2) o = o + part[(w2 - 1) * G + lane];
            store_fr(out, j, o);
        } else {
            o = Fr::zero();
        }
        Fr s = wave_reduce_fr(o);
        if (lane == 0) store_fr(partials, blockIdx.x, s);
    }
}

// out[j] = to_mont(sum_y in[y*m + j]): the partial tables of blockfold_kernel added up and brought back to Montgomery form
// (the sharded prover's local 256-entry table before it is gathered)
static __global__ __launch_bounds__(MLE_BLOCK) void slice_sums_kernel(const uint64_t* __restrict__ in, uint32_t ny, uint32_t m,
                                                               uint64_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * MLE_BLOCK + threadIdx.x;
    if (j >= m) return;
    Fr s = load_fr(in, j);
    for (uint32_t y = 1; y < ny; ++y) s = s + load_fr(in, (size_t)y * m + j);
    store_fr(out, j, fr_to_mont_outlined(s));
}

// ---- k-variable fold of a SMALL table (<= 2^18 entries), spread over the chip -----------------------------------------
// partial[y*m + c] = sum over the term range y of w[b] * in[b*m + c], as CANONICAL integers (the serial kernel's form; the
// conversion is linear, so partial tables still add up).  multifold_kernel<16> gives such a table to m/16 workgroups (16 at
// m = 256: ~16 us of one-wave-per-SIMD latency); here a workgroup of 1024 lanes takes OW outputs x (1024/OW) term slices
// of <= 4 terms each, sums the slices in LDS, and the term ranges go to blockIdx.y: 2^18 entries are 64 workgroups of
// 4 products per lane.  The serial kernel adds the gridDim.y (<= 8) partial tables (SmallArgs::stride).
constexpr int BF_BLOCK = 1024;
static __global__ __launch_bounds__(BF_BLOCK) void blockfold_kernel(const uint64_t* __restrict__ in, uint32_t m, uint32_t log_ow,
                                                                    uint32_t per, const uint64_t* __restrict__ weights,
                                                                    uint64_t* __restrict__ partial) {
    __shared__ Fr part[BF_BLOCK];
    const uint32_t ow = 1u << log_ow, sl_cnt = BF_BLOCK >> log_ow;
    const uint32_t o = threadIdx.x & (ow - 1), sl = threadIdx.x >> log_ow;
    const uint32_t c = blockIdx.x * ow + o;
    const uint32_t b0 = (blockIdx.y * sl_cnt + sl) * per;
    WideAcc acc;
    acc.clear();
    for (uint32_t u = 0; u < per; u += 2) {
        Fr v0 = load_fr(in, (size_t)(b0 + u) * m + c), w0 = load_fr(weights, b0 + u);
        Fr v1 = v0, w1 = w0;
        const bool two = u + 1 < per;
        if (two) { v1 = load_fr(in, (size_t)(b0 + u + 1) * m + c); w1 = load_fr(weights, b0 + u + 1); }
        acc.mac(w0, v0);
        if (two) acc.mac(w1, v1);
    }
    part[threadIdx.x] = wide_reduce(acc.lo, acc.hi);
    __syncthreads();
    for (uint32_t half = sl_cnt >> 1; half >= 1; half >>= 1) {       // tree over the slices
        if (sl < half) part[threadIdx.x] = part[threadIdx.x] + part[threadIdx.x + (half << log_ow)];
        __syncthreads();
    }
    if (sl == 0) store_fr(partial, (size_t)blockIdx.y * m + c, fr_from_mont_outlined(part[o]));
}

}  // namespace zk
