// shard_protocol.hpp -- the exchange protocols of the provers over tables SHARDED across ranks, as host logic
// that is independent of HIP: what turns the split-phase primitives (zkhip_sc_* / zkhip_mc_* in include/zkhip.h) into a proof.
// The loops they stand for: sumcheck/src/sumcheck.rs:29-61, sumcheck/src/composed/composed_sumcheck.rs:32-67,
// sumcheck/src/composed/multi_composed_sumcheck.rs:64-121.
//
// Rank g holds entry j * world + g of every table at local index j; rounds fold the most significant variable, so folds are local
// and a rank contributes partial sums only.  Modular addition is not a collective reduction: records are ALL-GATHERED and every rank
// adds them itself; the transcript is replicated.
//
// The protocols are templates over an ENGINE (the per-rank compute) and a COMM (the all-gather):
//   * libzkhip instantiates them with its HIP engines and a stream-ordered comm (shard.hip): nothing between a kernel and the
//     collective behind it waits for the host;
//   * tests/cpp/shard_protocol_host.cpp instantiates the same code with a callback engine over host memory, driven by the CPU
//     oracle over gloo (tests/test_distributed_cpu.py): the protocol itself is exercised with world_size 2 / 4 without a GPU.
// Buffers: engine.buffer(id, elems) returns engine-owned scratch of `elems` field elements (4 x uint64 each) for id 0..3, valid
// until the next request for the same id.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zkshard {

enum : int { OK = 0, ERR_PEER = -7 /* = ZKHIP_ERR_PEER */ };
#define ZKSHARD_TRY(expr)            \
    do {                             \
        const int _rc = (expr);      \
        if (_rc != 0) return _rc;    \
    } while (0)

// ---- a failing rank must not hang its peers ---------------------------------------------------------------------------------
// Every exchange is a collective: a rank that returned on an error before one would leave the others waiting in it for ever.  So a
// rank whose step fails (out of memory, a busy workspace, a failed launch) keeps a Fault and STAYS IN THE PROTOCOL: its compute
// steps are skipped, and it enters every remaining exchange with a POISON record of the right length -- the status word travels
// in band: the first element of the record is all ones, which no healthy rank can send (field elements are < r < 2^255), the rest is
// zero.  comm.check() looks at the first element of every rank's record behind each gather:
//   * a comm that sees the gathered bytes on the host (comm.immediate(): the gloo / callback-on-host transports) reports ERR_PEER at
//     once, and every rank -- the failed one included -- leaves the protocol right behind that exchange;
//   * the stream-ordered HIP comm cannot look without waiting for the GPU, so its check is one tiny kernel that raises a sticky flag
//     the entry point reads after the proof's final synchronisation (-> ZKHIP_ERR_PEER on the healthy ranks); the failed rank walks
//     the WHOLE remaining schedule.  That is why the plan calls below (overlap_plan, stage_plan, stage_record_len, record_len,
//     table_count, tail_capacity, use_stages) take the protocol's own n_local and must be PURE functions of shapes: they are still
//     called after a failure, when the engine may have no state at all (a failed begin).
// The failed rank returns its own error.  comm.inject(i) lets a test fail a rank in front of exchange i.
struct Fault {
    int rc = 0;                 // this rank's first error; 0 = healthy
    explicit operator bool() const { return rc != 0; }
};
// a compute step: skipped once the rank has failed; its error becomes the Fault (on one rank without a transport: returned at once)
#define ZKSHARD_STEP(f, comm, expr)                                   \
    do {                                                              \
        if (!(f).rc) {                                                \
            const int _rc = (expr);                                   \
            if (_rc != 0) {                                           \
                if ((comm).passthrough()) return _rc;                 \
                (f).rc = _rc;                                         \
            }                                                         \
        }                                                             \
    } while (0)
// engine scratch for a compute step (nullptr once failed: nothing reads it)
#define ZKSHARD_BUF(f, comm, var, e, id, elems)                       \
    uint64_t* var = nullptr;                                          \
    ZKSHARD_STEP(f, comm, ((var = (e).buffer(id, elems)) != nullptr) ? 0 : (e).nomem())

// one exchange: every rank's `elems` field elements in rank order.  One rank without a transport (comm.passthrough()): the send
// buffer IS the gathered record; a one-rank comm WITH a transport (a one-rank RCCL communicator) is exchanged through it like any other.
template <class E, class C>
static inline int gather(E& e, C& comm, Fault& f, const uint64_t* send, size_t elems, int recv_id, const uint64_t** out, uint32_t* exchanges) {
    const uint32_t idx = (*exchanges)++;
    if (!f.rc) {
        const int inj = comm.inject(idx);                                    // test hook: fail this rank in front of exchange idx
        if (inj != 0) {
            if (comm.passthrough()) return inj;
            f.rc = inj;
        }
    }
    if (comm.passthrough()) { *out = send; return OK; }
    uint64_t* recv = nullptr;
    if (!f.rc) {
        recv = e.buffer(recv_id, elems * comm.world());
        if (!recv) f.rc = e.nomem();
    }
    if (f.rc) {
        // enter the exchange all the same: a poison record of the right length (buffers the comm set aside when it was created)
        const uint64_t* p_send = nullptr;
        uint64_t* p_recv = nullptr;
        ZKSHARD_TRY(comm.poison(elems, &p_send, &p_recv));                   // (no buffers even for that: nothing more this rank can do)
        ZKSHARD_TRY(comm.all_gather(p_send, p_recv, elems * 32));
        *out = p_recv;
        return comm.immediate() ? f.rc : OK;                                 // immediate: every rank stops here; stream-ordered: walk on
    }
    ZKSHARD_TRY(comm.all_gather(send, recv, elems * 32));
    ZKSHARD_TRY(comm.check(recv, elems));                                    // ERR_PEER at once (immediate) or a sticky device-side flag
    *out = recv;
    return OK;
}

// Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61) over a sharded table.  Engine (see zkhip_sc_* for the meaning of each step):
//   size_t local_len(); bool use_stages(); uint32_t tail_capacity();
//   int overlap_plan(world, n_local, &k1, &k2, &mid, failed);  int overlap_sums(out, n);  int overlap_rounds1(gathered, world, claimed, mid_out, mid);
//   int overlap_rounds2(gathered, world, mid);
//   int stage_plan(world, n_local, &k, failed);  int stage_block_sums(out, n);  int stage_absorb(gathered, world, claimed, n);  int stage_fold();
//   int local_half_sums(out);  int absorb(gathered, world, claimed);  int fold();
//   int local_table(out, n);  int interleave(gathered, world, n_tables, n_local, out);  int tail(values, m, claimed);
// claimed: the sum the transcript absorbs first (nullptr: the true sum), consumed by whichever step opens the transcript.
template <class E, class C>
int sumcheck_prove(E& e, C& comm, const uint64_t* claimed, uint32_t* exchanges, int failed_before = 0) {
    const uint32_t world = comm.world();
    size_t n_local = e.local_len();
    bool absorbed = false;
    *exchanges = 0;
    Fault f;
    f.rc = failed_before;                                                    // a failed begin: the rank only walks the schedule
    if (f.rc && comm.passthrough()) return f.rc;
    const uint64_t* g = nullptr;
    if (e.use_stages()) {
        // overlapped stage (shards of 2^19..2^24 entries): k1 rounds on coarse block sums, then k2 rounds on the fine sums folded by
        // those k1 challenges WHILE the shard's k1-variable fold runs beside them -- the second exchange and the serial rounds hide
        // behind the one pass over the shard
        uint32_t k1 = 0, k2 = 0, mid = 0;
        ZKSHARD_TRY(e.overlap_plan(world, n_local, &k1, &k2, &mid, (bool)f));
        if (k1) {
            ZKSHARD_BUF(f, comm, mine, e, 0, (size_t)1 << k1);
            ZKSHARD_STEP(f, comm, e.overlap_sums(mine, (size_t)1 << k1));
            ZKSHARD_TRY(gather(e, comm, f, mine, (size_t)1 << k1, 1, &g, exchanges));         // exchange 1: coarse sums
            ZKSHARD_BUF(f, comm, mids, e, 2, mid);
            ZKSHARD_STEP(f, comm, e.overlap_rounds1(g, world, claimed, mids, mid));
            ZKSHARD_TRY(gather(e, comm, f, mids, mid, 3, &g, exchanges));                     // exchange 2, beside the fold
            ZKSHARD_STEP(f, comm, e.overlap_rounds2(g, world, mid));
            absorbed = true;
            n_local >>= k1 + k2;
        }
        // stage form: one exchange per k rounds (32 * 2^k bytes per rank), then one local k-variable fold
        for (;;) {
            uint32_t k = 0;
            ZKSHARD_TRY(e.stage_plan(world, n_local, &k, (bool)f));
            if (!k) break;
            ZKSHARD_BUF(f, comm, mine, e, 0, (size_t)1 << k);
            ZKSHARD_STEP(f, comm, e.stage_block_sums(mine, (size_t)1 << k));
            ZKSHARD_TRY(gather(e, comm, f, mine, (size_t)1 << k, 1, &g, exchanges));
            ZKSHARD_STEP(f, comm, e.stage_absorb(g, world, absorbed ? nullptr : claimed, (size_t)1 << k));
            absorbed = true;
            ZKSHARD_STEP(f, comm, e.stage_fold());
            n_local >>= k;
        }
    }
    // round form: one 64-byte exchange per round -- the whole protocol for engines without stages, and the way down to the tail
    // size where a stage no longer fits (shards of a few entries on many ranks)
    const size_t cap = e.tail_capacity();
    while (n_local * world > cap && n_local > 1) {
        ZKSHARD_BUF(f, comm, send, e, 0, 2);
        ZKSHARD_STEP(f, comm, e.local_half_sums(send));
        ZKSHARD_TRY(gather(e, comm, f, send, 2, 1, &g, exchanges));
        ZKSHARD_STEP(f, comm, e.absorb(g, world, absorbed ? nullptr : claimed));      // local modular add + transcript -> challenge
        absorbed = true;
        ZKSHARD_STEP(f, comm, e.fold());                                               // local: partners share the low index bits
        n_local /= 2;
    }
    if (n_local * world > 1) {
        // the whole remaining table fits the replicated tail: gather it and finish on every rank
        ZKSHARD_BUF(f, comm, mine, e, 0, n_local);
        ZKSHARD_STEP(f, comm, e.local_table(mine, n_local));
        ZKSHARD_TRY(gather(e, comm, f, mine, n_local, 1, &g, exchanges));
        const uint64_t* full = g;
        if (world > 1) {                                                      // entry j * world + g <- rank g, local j
            ZKSHARD_BUF(f, comm, il, e, 2, n_local * world);
            ZKSHARD_STEP(f, comm, e.interleave(g, world, 1, n_local, il));
            full = il;
        }
        ZKSHARD_STEP(f, comm, e.tail(full, (uint32_t)(n_local * world), absorbed ? nullptr : claimed));
    }
    return f.rc;
}

// ComposedSumcheck::prove (composed_sumcheck.rs:32-67) / MultiComposedSumcheckProver::prove_partial (multi_composed_sumcheck.rs:56-121)
// over sharded tables.  One exchange per round: a record of (K_p + 1) partial sums per term.  Claims whose terms are products of TWO
// tables take TWO rounds per exchange where use_stages asks for it (the product is bilinear in the block sums: the record is the 16
// cross-block sums + 4 additive block sums per term, csrc/composed_stage.hpp).  Engine (zkhip_mc_*):
//   size_t local_len(); uint32_t tail_capacity(); uint32_t record_len(); uint32_t table_count();
//   int stage_record_len(world, n_local, &vals, failed);  int stage_sums(out, vals);  int stage_absorb(gathered, world, vals);
//   int round_sums(out, rec);  int absorb(gathered, world, rec);
//   int local_tables(out, n_tables, n_local);  int interleave(...);  int tail(tables, m, n_tables);
template <class E, class C>
int composed_prove(E& e, C& comm, bool use_stages, uint32_t* exchanges, int failed_before = 0) {
    const uint32_t world = comm.world();
    size_t n_local = e.local_len();
    const size_t cap = e.tail_capacity();
    const uint32_t rec = e.record_len();
    *exchanges = 0;
    Fault f;
    f.rc = failed_before;
    if (f.rc && comm.passthrough()) return f.rc;
    const uint64_t* g = nullptr;
    if (use_stages) {
        while (n_local * world > cap && n_local >= 4) {
            uint32_t vals = 0;
            ZKSHARD_TRY(e.stage_record_len(world, n_local, &vals, (bool)f));
            if (!vals) break;
            ZKSHARD_BUF(f, comm, send, e, 0, vals);
            ZKSHARD_STEP(f, comm, e.stage_sums(send, vals));                       // 16 cross-block sums (+ 4 block sums) per term
            ZKSHARD_TRY(gather(e, comm, f, send, vals, 1, &g, exchanges));         // ONE exchange for two rounds
            ZKSHARD_STEP(f, comm, e.stage_absorb(g, world, vals));                 // two transcript rounds + the fold by both challenges
            n_local /= 4;
        }
    }
    while (n_local * world > cap && n_local > 1) {
        ZKSHARD_BUF(f, comm, send, e, 0, rec);
        ZKSHARD_STEP(f, comm, e.round_sums(send, rec));                            // fold at the previous challenge + partial sums
        ZKSHARD_TRY(gather(e, comm, f, send, rec, 1, &g, exchanges));              // <= 768 B per rank
        ZKSHARD_STEP(f, comm, e.absorb(g, world, rec));                            // local modular add + transcript -> challenge
        n_local /= 2;
    }
    if (n_local * world > 1) {
        const uint32_t nt = e.table_count();
        ZKSHARD_BUF(f, comm, mine, e, 0, (size_t)nt * n_local);
        ZKSHARD_STEP(f, comm, e.local_tables(mine, nt, n_local));
        ZKSHARD_TRY(gather(e, comm, f, mine, (size_t)nt * n_local, 1, &g, exchanges));
        const uint64_t* full = g;
        if (world > 1) {                                                      // table t, entry j * world + g <- rank g, table t, local j
            ZKSHARD_BUF(f, comm, il, e, 2, (size_t)nt * n_local * world);
            ZKSHARD_STEP(f, comm, e.interleave(g, world, nt, n_local, il));
            full = il;
        }
        ZKSHARD_STEP(f, comm, e.tail(full, (uint32_t)(n_local * world), nt));
    }
    return f.rc;
}

}  // namespace zkshard
