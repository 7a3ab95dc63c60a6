// shard.hpp -- the communicator of the sharded provers (include/zkhip.h, zkhip_comm_*) and the HIP engines that plug the split-phase
// primitives (zkhip_sc_* / zkhip_mc_*) into the exchange protocols of shard_protocol.hpp.  Shared by shard.hip (C ABI) and gkr.hip (the
// sharded GKR prover runs two composed sessions per layer through the same protocol).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zkhip.h"
#include "ctx.hpp"
#include "shard_protocol.hpp"

// One rank's end of the exchange.  Either a caller-supplied all-gather (any transport: rccl-sys in a Rust host, gloo in the tests) or
// an RCCL communicator the library created itself (librccl resolved at run time, shard.hip); both are called with the context's
// stream, so an exchange is ordered behind the kernels that produced its payload without the host waiting for them.
struct zkhip_comm {
    zkhip_ctx* c = nullptr;
    uint32_t rank_ = 0, world_ = 1;
    zkhip_all_gather_fn fn = nullptr;
    void* user = nullptr;
    void* nccl = nullptr;                    // ncclComm_t when the library owns an RCCL communicator
    uint64_t n_exchanges = 0, n_bytes = 0;   // cumulative (zkhip_comm_stats)
    void* buf[4] = {};                       // protocol scratch (send / gathered / interleaved records), grow-only
    size_t cap[4] = {};
    // failure propagation (shard_protocol.hpp): what a FAILED rank still needs to enter the remaining exchanges, set aside when the
    // communicator is created so that no allocation stands between a failure and its report; and the healthy ranks' sticky flag
    void* poison_send = nullptr;             // POISON_ELEMS elements: the first all ones, the rest zero
    void* poison_recv = nullptr;             // world x POISON_ELEMS elements
    uint32_t* peer_flag_h = nullptr;         // host-mapped: raised by shard_check_kernel when a gathered record carries the poison mark
    uint32_t* peer_flag_d = nullptr;
    int inject_at = -1, inject_rc = 0;       // test hook (zkhip_comm_inject_failure): fail in front of exchange inject_at of the next protocol run
    static constexpr size_t POISON_ELEMS = (size_t)1 << 16;   // >= the longest record of any protocol (zk_shard_max_record(), checked per run)
    zkhip_comm* solo = nullptr;              // a one-rank comm on the same context (steps of a sharded proof that run whole on every rank), kept
    zkhip_comm* solo_comm() {
        if (world_ == 1) return this;
        if (!solo) {
            solo = new (std::nothrow) zkhip_comm();
            if (solo) solo->c = c;
        }
        return solo;
    }
    uint32_t world() const { return world_; }
    bool passthrough() const { return world_ == 1 && !nccl && !fn; }
    bool immediate() const { return false; }                           // exchanges are stream-ordered: the host never sees a gathered record
    int inject(uint32_t idx) {
        if (inject_at < 0 || (uint32_t)inject_at != idx) return 0;
        inject_at = -1;
        return inject_rc;
    }
    int all_gather(const void* d_send, void* d_recv, size_t bytes);    // shard.hip
    int poison(size_t elems, const uint64_t** d_send, uint64_t** d_recv);
    int check(const uint64_t* d_gathered, size_t elems);
    int setup_fault_buffers();                                         // world > 1 or a transport: called by the creators
    // after the proof's final synchronisation: did any exchange since the last call carry a poisoned record?
    bool take_peer_failure() {
        if (!peer_flag_h || !*peer_flag_h) return false;
        *peer_flag_h = 0;
        return true;
    }
    uint64_t* buffer(int id, size_t elems) {
        const size_t need = (elems ? elems : 1) * 32;
        if (need > cap[id]) {
            // growth: kernels of an earlier protocol step may still read the old allocation
            if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;
            if (buf[id]) (void)hipFree(buf[id]);
            buf[id] = nullptr; cap[id] = 0;
            const size_t want = std::max<size_t>(need, (size_t)64 * 1024);
            if (hipMalloc(&buf[id], want) != hipSuccess) return nullptr;
            cap[id] = want;
        }
        return (uint64_t*)buf[id];
    }
};

// pure plans (zkhip.hip, composed.hip): the exchange schedule from shapes alone
int zk_sc_plan_stage(size_t cn, uint32_t world, uint32_t* k_out);
int zk_sc_plan_overlap(size_t cn, uint32_t world, uint32_t* k1, uint32_t* k2, uint32_t* mid_entries, uint32_t* ny_out);
int zk_mc_shape(const uint32_t* sizes, uint32_t n_terms, uint32_t n_lin, uint32_t* rec, uint32_t* n_tables, uint32_t* tail_len, uint32_t* stage_vals);
int zk_mc_state_shape(zkhip_mc_state* s, uint32_t* rec, uint32_t* n_tables, uint32_t* tail_len, uint32_t* stage_vals, size_t* n_local);

int zk_shard_interleave(zkhip_ctx* c, const uint64_t* d_gathered, uint32_t world, uint32_t n_tables, size_t n_local, uint64_t* d_out);   // shard.hip

namespace zkshard {

// Sumcheck::prove over a shard: zkhip_sc_* behind the engine interface of shard_protocol.hpp.  st may be nullptr (zkhip_sc_begin failed on
// this rank): then only the pure calls are reached -- local_len / use_stages / tail_capacity / the plans with failed = true.
struct HipScEngine {
    zkhip_sc_state* st;
    zkhip_comm* comm;
    bool stages = true;
    size_t n_local0 = 0;                     // entries of this rank's shard (the protocol's starting point when there is no state)
    int nomem() const { return ZKHIP_ERR_NOMEM; }
    uint64_t* buffer(int id, size_t elems) { return comm->buffer(id, elems); }
    size_t local_len() { size_t n = n_local0; if (st) zkhip_sc_local_len(st, &n); return n; }
    bool use_stages() const { return stages; }
    uint32_t tail_capacity() const { return (uint32_t)zkhip_sc_tail_capacity(); }
    int overlap_plan(uint32_t world, size_t n_local, uint32_t* k1, uint32_t* k2, uint32_t* mid, bool failed) {
        return failed || !st ? zk_sc_plan_overlap(n_local, world, k1, k2, mid, nullptr) : zkhip_sc_overlap_plan(st, world, k1, k2, mid);
    }
    int overlap_sums(uint64_t* out, size_t) { return zkhip_sc_overlap_sums(st, out); }
    int overlap_rounds1(const uint64_t* g, uint32_t world, const uint64_t* claimed, uint64_t* mid_out, uint32_t) { return zkhip_sc_overlap_rounds1(st, g, world, claimed, mid_out); }
    int overlap_rounds2(const uint64_t* g, uint32_t world, uint32_t) { return zkhip_sc_overlap_rounds2(st, g, world); }
    int stage_plan(uint32_t world, size_t n_local, uint32_t* k, bool failed) {
        return failed || !st ? zk_sc_plan_stage(n_local, world, k) : zkhip_sc_stage_plan(st, world, k);
    }
    int stage_block_sums(uint64_t* out, size_t) { return zkhip_sc_stage_block_sums(st, out); }
    int stage_absorb(const uint64_t* g, uint32_t world, const uint64_t* claimed, size_t) { return zkhip_sc_stage_absorb(st, g, world, claimed); }
    int stage_fold() { return zkhip_sc_stage_fold(st); }
    int local_half_sums(uint64_t* out) { return zkhip_sc_local_half_sums(st, out); }
    int absorb(const uint64_t* g, uint32_t world, const uint64_t* claimed) { return zkhip_sc_absorb(st, g, world, claimed); }
    int fold() { return zkhip_sc_fold(st); }
    int local_table(uint64_t* out, size_t) { return zkhip_sc_local_table(st, out); }
    int interleave(const uint64_t* g, uint32_t world, uint32_t nt, size_t n_local, uint64_t* out) { return zk_shard_interleave(comm->c, g, world, nt, n_local, out); }
    int tail(const uint64_t* values, uint32_t m, const uint64_t* claimed) { return zkhip_sc_tail(st, values, m, claimed); }
};

// ComposedSumcheck::prove / MultiComposedSumcheckProver::prove_partial over shards: zkhip_mc_*.  The shape (zk_mc_shape) answers the pure
// calls, with or without a session (st == nullptr: zkhip_mc_begin failed on this rank).
struct HipMcEngine {
    zkhip_mc_state* st;
    zkhip_comm* comm;
    size_t n_local0 = 0;
    uint32_t rec = 0, n_tables = 0, tail_len = 0, stage_vals = 0;
    int shape(const uint32_t* sizes, uint32_t n_terms, uint32_t n_lin, size_t n_local) {      // no session: from the arguments
        n_local0 = n_local;
        return zk_mc_shape(sizes, n_terms, n_lin, &rec, &n_tables, &tail_len, &stage_vals);
    }
    int shape_of_session() { return zk_mc_state_shape(st, &rec, &n_tables, &tail_len, &stage_vals, &n_local0); }
    int nomem() const { return ZKHIP_ERR_NOMEM; }
    uint64_t* buffer(int id, size_t elems) { return comm->buffer(id, elems); }
    size_t local_len() { return n_local0; }
    uint32_t tail_capacity() { return tail_len; }
    uint32_t record_len() { return rec; }
    uint32_t table_count() { return n_tables; }
    // a stage is two rounds on >= 4 local entries (ComposedRun::stage_possible(4); nothing is pending inside the protocol's stage loop)
    int stage_record_len(uint32_t world, size_t n_local, uint32_t* vals, bool failed) {
        const uint32_t pure = (n_local >= 4 && n_local * world >= 4) ? stage_vals : 0u;
        if (failed || !st) { *vals = pure; return ZKHIP_OK; }
        const int rc = zkhip_mc_stage_record_len(st, vals);
        return rc != ZKHIP_OK ? rc : (*vals == pure ? ZKHIP_OK : ZKHIP_ERR_SHAPE);      // the schedule a failed rank would walk IS the schedule
    }
    int stage_sums(uint64_t* out, uint32_t) { return zkhip_mc_stage_sums(st, out); }
    int stage_absorb(const uint64_t* g, uint32_t world, uint32_t) { return zkhip_mc_stage_absorb(st, g, world); }
    int round_sums(uint64_t* out, uint32_t) { return zkhip_mc_round_sums(st, out); }
    int absorb(const uint64_t* g, uint32_t world, uint32_t) { return zkhip_mc_absorb(st, g, world); }
    int local_tables(uint64_t* out, uint32_t, size_t) { return zkhip_mc_local_tables(st, out); }
    int interleave(const uint64_t* g, uint32_t world, uint32_t nt, size_t n_local, uint64_t* out) { return zk_shard_interleave(comm->c, g, world, nt, n_local, out); }
    int tail(const uint64_t* tables, uint32_t m, uint32_t) { return zkhip_mc_tail(st, tables, m); }
};

}  // namespace zkshard
