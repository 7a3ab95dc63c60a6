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
    int all_gather(const void* d_send, void* d_recv, size_t bytes);    // shard.hip
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

int zk_shard_interleave(zkhip_ctx* c, const uint64_t* d_gathered, uint32_t world, uint32_t n_tables, size_t n_local, uint64_t* d_out);   // shard.hip

namespace zkshard {

// Sumcheck::prove over a shard: zkhip_sc_* behind the engine interface of shard_protocol.hpp
struct HipScEngine {
    zkhip_sc_state* st;
    zkhip_comm* comm;
    bool stages = true;
    int nomem() const { return ZKHIP_ERR_NOMEM; }
    uint64_t* buffer(int id, size_t elems) { return comm->buffer(id, elems); }
    size_t local_len() { size_t n = 0; zkhip_sc_local_len(st, &n); return n; }
    bool use_stages() const { return stages; }
    uint32_t tail_capacity() const { return (uint32_t)zkhip_sc_tail_capacity(); }
    int overlap_plan(uint32_t world, uint32_t* k1, uint32_t* k2, uint32_t* mid) { return zkhip_sc_overlap_plan(st, world, k1, k2, mid); }
    int overlap_sums(uint64_t* out, size_t) { return zkhip_sc_overlap_sums(st, out); }
    int overlap_rounds1(const uint64_t* g, uint32_t world, const uint64_t* claimed, uint64_t* mid_out, uint32_t) { return zkhip_sc_overlap_rounds1(st, g, world, claimed, mid_out); }
    int overlap_rounds2(const uint64_t* g, uint32_t world, uint32_t) { return zkhip_sc_overlap_rounds2(st, g, world); }
    int stage_plan(uint32_t world, uint32_t* k) { return zkhip_sc_stage_plan(st, world, k); }
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

// ComposedSumcheck::prove / MultiComposedSumcheckProver::prove_partial over shards: zkhip_mc_*
struct HipMcEngine {
    zkhip_mc_state* st;
    zkhip_comm* comm;
    int nomem() const { return ZKHIP_ERR_NOMEM; }
    uint64_t* buffer(int id, size_t elems) { return comm->buffer(id, elems); }
    size_t local_len() { size_t n = 0; zkhip_mc_local_len(st, &n); return n; }
    uint32_t tail_capacity() { uint32_t cap = 0; zkhip_mc_tail_capacity(st, &cap); return cap; }
    uint32_t record_len() { uint32_t rec = 0; zkhip_mc_record_len(st, &rec, nullptr); return rec; }
    uint32_t table_count() { uint32_t rec = 0, nt = 0; zkhip_mc_record_len(st, &rec, &nt); return nt; }
    int stage_record_len(uint32_t* vals) { return zkhip_mc_stage_record_len(st, vals); }
    int stage_sums(uint64_t* out, uint32_t) { return zkhip_mc_stage_sums(st, out); }
    int stage_absorb(const uint64_t* g, uint32_t world, uint32_t) { return zkhip_mc_stage_absorb(st, g, world); }
    int round_sums(uint64_t* out, uint32_t) { return zkhip_mc_round_sums(st, out); }
    int absorb(const uint64_t* g, uint32_t world, uint32_t) { return zkhip_mc_absorb(st, g, world); }
    int local_tables(uint64_t* out, uint32_t, size_t) { return zkhip_mc_local_tables(st, out); }
    int interleave(const uint64_t* g, uint32_t world, uint32_t nt, size_t n_local, uint64_t* out) { return zk_shard_interleave(comm->c, g, world, nt, n_local, out); }
    int tail(const uint64_t* tables, uint32_t m, uint32_t) { return zkhip_mc_tail(st, tables, m); }
};

}  // namespace zkshard
