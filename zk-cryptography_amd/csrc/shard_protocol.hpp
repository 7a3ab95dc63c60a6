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

enum : int { OK = 0 };
#define ZKSHARD_TRY(expr)            \
    do {                             \
        const int _rc = (expr);      \
        if (_rc != 0) return _rc;    \
    } while (0)

// one exchange: every rank's `elems` field elements in rank order.  One rank without a transport (comm.passthrough()): the send
// buffer IS the gathered record; a one-rank comm WITH a transport (a one-rank RCCL communicator) is exchanged through it like any other.
template <class E, class C>
static inline int gather(E& e, C& comm, const uint64_t* send, size_t elems, int recv_id, const uint64_t** out, uint32_t* exchanges) {
    ++*exchanges;
    if (comm.passthrough()) { *out = send; return OK; }
    uint64_t* recv = e.buffer(recv_id, elems * comm.world());
    if (!recv) return e.nomem();
    ZKSHARD_TRY(comm.all_gather(send, recv, elems * 32));
    *out = recv;
    return OK;
}

// Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61) over a sharded table.  Engine (see zkhip_sc_* for the meaning of each step):
//   size_t local_len(); bool use_stages(); uint32_t tail_capacity();
//   int overlap_plan(world, &k1, &k2, &mid);  int overlap_sums(out, n);  int overlap_rounds1(gathered, world, claimed, mid_out, mid);
//   int overlap_rounds2(gathered, world, mid);
//   int stage_plan(world, &k);  int stage_block_sums(out, n);  int stage_absorb(gathered, world, claimed, n);  int stage_fold();
//   int local_half_sums(out);  int absorb(gathered, world, claimed);  int fold();
//   int local_table(out, n);  int interleave(gathered, world, n_tables, n_local, out);  int tail(values, m, claimed);
// claimed: the sum the transcript absorbs first (nullptr: the true sum), consumed by whichever step opens the transcript.
template <class E, class C>
int sumcheck_prove(E& e, C& comm, const uint64_t* claimed, uint32_t* exchanges) {
    const uint32_t world = comm.world();
    size_t n_local = e.local_len();
    bool absorbed = false;
    *exchanges = 0;
    const uint64_t* g = nullptr;
    if (e.use_stages()) {
        // overlapped stage (shards of 2^19..2^24 entries): k1 rounds on coarse block sums, then k2 rounds on the fine sums folded by
        // those k1 challenges WHILE the shard's k1-variable fold runs beside them -- the second exchange and the serial rounds hide
        // behind the one pass over the shard
        uint32_t k1 = 0, k2 = 0, mid = 0;
        ZKSHARD_TRY(e.overlap_plan(world, &k1, &k2, &mid));
        if (k1) {
            uint64_t* mine = e.buffer(0, (size_t)1 << k1);
            if (!mine) return e.nomem();
            ZKSHARD_TRY(e.overlap_sums(mine, (size_t)1 << k1));
            ZKSHARD_TRY(gather(e, comm, mine, (size_t)1 << k1, 1, &g, exchanges));            // exchange 1: coarse sums
            uint64_t* mids = e.buffer(2, mid);
            if (!mids) return e.nomem();
            ZKSHARD_TRY(e.overlap_rounds1(g, world, claimed, mids, mid));
            ZKSHARD_TRY(gather(e, comm, mids, mid, 3, &g, exchanges));                        // exchange 2, beside the fold
            ZKSHARD_TRY(e.overlap_rounds2(g, world, mid));
            absorbed = true;
            n_local >>= k1 + k2;
        }
        // stage form: one exchange per k rounds (32 * 2^k bytes per rank), then one local k-variable fold
        for (;;) {
            uint32_t k = 0;
            ZKSHARD_TRY(e.stage_plan(world, &k));
            if (!k) break;
            uint64_t* mine = e.buffer(0, (size_t)1 << k);
            if (!mine) return e.nomem();
            ZKSHARD_TRY(e.stage_block_sums(mine, (size_t)1 << k));
            ZKSHARD_TRY(gather(e, comm, mine, (size_t)1 << k, 1, &g, exchanges));
            ZKSHARD_TRY(e.stage_absorb(g, world, absorbed ? nullptr : claimed, (size_t)1 << k));
            absorbed = true;
            ZKSHARD_TRY(e.stage_fold());
            n_local >>= k;
        }
    }
    // round form: one 64-byte exchange per round -- the whole protocol for engines without stages, and the way down to the tail
    // size where a stage no longer fits (shards of a few entries on many ranks)
    const size_t cap = e.tail_capacity();
    while (n_local * world > cap && n_local > 1) {
        uint64_t* send = e.buffer(0, 2);
        if (!send) return e.nomem();
        ZKSHARD_TRY(e.local_half_sums(send));
        ZKSHARD_TRY(gather(e, comm, send, 2, 1, &g, exchanges));
        ZKSHARD_TRY(e.absorb(g, world, absorbed ? nullptr : claimed));      // local modular add + transcript -> challenge
        absorbed = true;
        ZKSHARD_TRY(e.fold());                                               // local: partners share the low index bits
        n_local /= 2;
    }
    if (n_local * world > 1) {
        // the whole remaining table fits the replicated tail: gather it and finish on every rank
        uint64_t* mine = e.buffer(0, n_local);
        if (!mine) return e.nomem();
        ZKSHARD_TRY(e.local_table(mine, n_local));
        ZKSHARD_TRY(gather(e, comm, mine, n_local, 1, &g, exchanges));
        const uint64_t* full = g;
        if (world > 1) {                                                      // entry j * world + g <- rank g, local j
            uint64_t* il = e.buffer(2, n_local * world);
            if (!il) return e.nomem();
            ZKSHARD_TRY(e.interleave(g, world, 1, n_local, il));
            full = il;
        }
        ZKSHARD_TRY(e.tail(full, (uint32_t)(n_local * world), absorbed ? nullptr : claimed));
    }
    return OK;
}

// ComposedSumcheck::prove (composed_sumcheck.rs:32-67) / MultiComposedSumcheckProver::prove_partial (multi_composed_sumcheck.rs:56-121)
// over sharded tables.  One exchange per round: a record of (K_p + 1) partial sums per term.  Claims whose terms are products of TWO
// tables take TWO rounds per exchange where use_stages asks for it (the product is bilinear in the block sums: the record is the 16
// cross-block sums + 4 additive block sums per term, csrc/composed_stage.hpp).  Engine (zkhip_mc_*):
//   size_t local_len(); uint32_t tail_capacity(); uint32_t record_len(); uint32_t table_count();
//   int stage_record_len(&vals);  int stage_sums(out, vals);  int stage_absorb(gathered, world, vals);
//   int round_sums(out, rec);  int absorb(gathered, world, rec);
//   int local_tables(out, n_tables, n_local);  int interleave(...);  int tail(tables, m, n_tables);
template <class E, class C>
int composed_prove(E& e, C& comm, bool use_stages, uint32_t* exchanges) {
    const uint32_t world = comm.world();
    size_t n_local = e.local_len();
    const size_t cap = e.tail_capacity();
    const uint32_t rec = e.record_len();
    *exchanges = 0;
    const uint64_t* g = nullptr;
    if (use_stages) {
        while (n_local * world > cap && n_local >= 4) {
            uint32_t vals = 0;
            ZKSHARD_TRY(e.stage_record_len(&vals));
            if (!vals) break;
            uint64_t* send = e.buffer(0, vals);
            if (!send) return e.nomem();
            ZKSHARD_TRY(e.stage_sums(send, vals));                            // 16 cross-block sums (+ 4 block sums) per term
            ZKSHARD_TRY(gather(e, comm, send, vals, 1, &g, exchanges));       // ONE exchange for two rounds
            ZKSHARD_TRY(e.stage_absorb(g, world, vals));                      // two transcript rounds + the fold by both challenges
            n_local /= 4;
        }
    }
    while (n_local * world > cap && n_local > 1) {
        uint64_t* send = e.buffer(0, rec);
        if (!send) return e.nomem();
        ZKSHARD_TRY(e.round_sums(send, rec));                                 // fold at the previous challenge + partial sums
        ZKSHARD_TRY(gather(e, comm, send, rec, 1, &g, exchanges));            // <= 768 B per rank
        ZKSHARD_TRY(e.absorb(g, world, rec));                                 // local modular add + transcript -> challenge
        n_local /= 2;
    }
    if (n_local * world > 1) {
        const uint32_t nt = e.table_count();
        uint64_t* mine = e.buffer(0, (size_t)nt * n_local);
        if (!mine) return e.nomem();
        ZKSHARD_TRY(e.local_tables(mine, nt, n_local));
        ZKSHARD_TRY(gather(e, comm, mine, (size_t)nt * n_local, 1, &g, exchanges));
        const uint64_t* full = g;
        if (world > 1) {                                                      // table t, entry j * world + g <- rank g, table t, local j
            uint64_t* il = e.buffer(2, (size_t)nt * n_local * world);
            if (!il) return e.nomem();
            ZKSHARD_TRY(e.interleave(g, world, nt, n_local, il));
            full = il;
        }
        ZKSHARD_TRY(e.tail(full, (uint32_t)(n_local * world), nt));
    }
    return OK;
}

}  // namespace zkshard
