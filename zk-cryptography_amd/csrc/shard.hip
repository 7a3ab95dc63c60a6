// shard.hip -- the sharded provers behind one C-ABI call each (include/zkhip.h, "the sharded provers"): the communicator (a
// caller-supplied all-gather, or an RCCL communicator the library opens at run time), the HIP side of the exchange protocols of
// shard_protocol.hpp, and the entry points.  gfx950 only; no CPU fallback.
#include "../../include/zkhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "host_util.hpp"
#include "shard.hpp"

// ---------------------------------------------------------------------------------------
// RCCL, resolved at run time: libzkhip.so does not link it (a host that brings its own transport never loads it)
// ---------------------------------------------------------------------------------------
namespace {
struct RcclId { char internal[128]; };                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    bool ok = false;
};
constexpr int RCCL_UINT8 = 1;                          // ncclUint8
Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        const char* env = std::getenv("ZKHIP_RCCL_LIB");
        // a process that already holds an RCCL (PyTorch ships one) gets that one back by its soname
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            q.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (q.lib) break;
        }
        if (!q.lib) return q;
        q.GetUniqueId = (int (*)(RcclId*))dlsym(q.lib, "ncclGetUniqueId");
        q.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(q.lib, "ncclCommInitRank");
        q.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(q.lib, "ncclAllGather");
        q.CommDestroy = (int (*)(void*))dlsym(q.lib, "ncclCommDestroy");
        q.GetVersion = (int (*)(int*))dlsym(q.lib, "ncclGetVersion");
        q.ok = q.GetUniqueId && q.CommInitRank && q.AllGather && q.CommDestroy;
        return q;
    }();
    return r;
}

// out[t][j * world + g] = in[g][t][j]: the gathered shards of n_tables tables back in natural order (32-byte entries as two 16-byte halves)
__global__ void shard_interleave_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t world, uint32_t log_world,
                                        uint32_t n_tables, size_t n_local) {
    const size_t total = 2 * (size_t)n_tables * n_local * world;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i >> 1, per = n_local << log_world;
        const size_t t = e / per, r = e - t * per;
        const size_t j = r >> log_world, g = r & (world - 1);
        out[i] = in[2 * ((g * n_tables + t) * n_local + j) + (i & 1)];
    }
}
// failure propagation (shard_protocol.hpp): a failed rank's record begins with an all-ones element -- no field element looks like that
__global__ void shard_poison_fill_kernel(uint4* __restrict__ send, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        send[i] = i < 2 ? make_uint4(~0u, ~0u, ~0u, ~0u) : make_uint4(0u, 0u, 0u, 0u);
}
__global__ void shard_check_kernel(const uint64_t* __restrict__ gathered, size_t elems, uint32_t world, uint32_t* flag) {
    const uint32_t g = threadIdx.x;
    if (g < world && gathered[4 * elems * (size_t)g + 3] == ~(uint64_t)0) atomicOr(flag, 1u + g);     // (host-mapped: system-scope store)
}
}  // namespace

int zk_shard_interleave(zkhip_ctx* c, const uint64_t* d_gathered, uint32_t world, uint32_t n_tables, size_t n_local, uint64_t* d_out) {
    const size_t total = 2 * (size_t)n_tables * n_local * world;
    if (!total) return ZKHIP_OK;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 1024);
    hipLaunchKernelGGL(shard_interleave_kernel, dim3(grid), dim3(256), 0, c->stream, (const uint4*)d_gathered, (uint4*)d_out, world,
                       log2_exact(world), n_tables, n_local);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

int zkhip_comm::all_gather(const void* d_send, void* d_recv, size_t bytes) {
    ++n_exchanges;
    n_bytes += bytes;
    if (nccl) {      // also with one rank: a one-rank RCCL communicator is how a single-GPU box exercises this path
        if (rccl().AllGather(d_send, d_recv, bytes, RCCL_UINT8, nccl, c->stream) != 0) return ZKHIP_ERR_HIP;
        return ZKHIP_OK;
    }
    if (world_ == 1) {
        if (d_recv != d_send) ZK_HIP(c, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, c->stream));
        return ZKHIP_OK;
    }
    if (!fn) return ZKHIP_ERR_ARG;
    return fn(user, d_send, d_recv, bytes, (void*)c->stream) == 0 ? ZKHIP_OK : ZKHIP_ERR_HIP;
}

int zkhip_comm::setup_fault_buffers() {
    if (passthrough()) return ZKHIP_OK;              // one rank, no transport: an error simply returns
    ZK_TRY(c->activate());
    if (hipMalloc(&poison_send, POISON_ELEMS * 32) != hipSuccess) return ZKHIP_ERR_NOMEM;
    if (hipMalloc(&poison_recv, POISON_ELEMS * 32 * (size_t)world_) != hipSuccess) return ZKHIP_ERR_NOMEM;
    if (hipHostMalloc((void**)&peer_flag_h, 64, hipHostMallocMapped) != hipSuccess) return ZKHIP_ERR_NOMEM;
    *peer_flag_h = 0;
    if (hipHostGetDevicePointer((void**)&peer_flag_d, peer_flag_h, 0) != hipSuccess) return ZKHIP_ERR_HIP;
    hipLaunchKernelGGL(shard_poison_fill_kernel, dim3(256), dim3(256), 0, c->stream, (uint4*)poison_send, POISON_ELEMS * 2);
    ZK_HIP(c, hipGetLastError());
    return c->wait_stream();
}
int zkhip_comm::poison(size_t elems, const uint64_t** d_send, uint64_t** d_recv) {
    if (!poison_send || !poison_recv || elems > POISON_ELEMS) return ZKHIP_ERR_NOMEM;
    *d_send = (const uint64_t*)poison_send;
    *d_recv = (uint64_t*)poison_recv;
    return ZKHIP_OK;
}
int zkhip_comm::check(const uint64_t* d_gathered, size_t elems) {
    if (!peer_flag_d) return ZKHIP_OK;
    if (elems > POISON_ELEMS) return ZKHIP_ERR_SHAPE;      // a record a failed rank could not answer: a protocol grew past the reserve
    hipLaunchKernelGGL(shard_check_kernel, dim3(1), dim3(64), 0, c->stream, d_gathered, elems, world_, peer_flag_d);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------------------
extern "C" int zkhip_comm_create(zkhip_ctx* c, uint32_t rank, uint32_t world, zkhip_all_gather_fn fn, void* user, zkhip_comm** out) {
    if (!c || !out || world == 0 || rank >= world || (world > 1 && !fn)) return ZKHIP_ERR_ARG;
    if (!is_pow2(world)) return ZKHIP_ERR_SHAPE;          // the tables have 2^n entries
    zkhip_comm* m = new (std::nothrow) zkhip_comm();
    if (!m) return ZKHIP_ERR_NOMEM;
    m->c = c; m->rank_ = rank; m->world_ = world; m->fn = fn; m->user = user;
    const int rc = m->setup_fault_buffers();
    if (rc != ZKHIP_OK) { zkhip_comm_destroy(m); return rc; }
    *out = m;
    return ZKHIP_OK;
}
extern "C" int zkhip_rccl_unique_id(uint8_t* h_id128) {
    if (!h_id128) return ZKHIP_ERR_ARG;
    if (!rccl().ok) return ZKHIP_ERR_HIP;
    RcclId id;
    if (rccl().GetUniqueId(&id) != 0) return ZKHIP_ERR_HIP;
    std::memcpy(h_id128, id.internal, 128);
    return ZKHIP_OK;
}
extern "C" int zkhip_rccl_version(int* version) {
    if (!version) return ZKHIP_ERR_ARG;
    if (!rccl().ok || !rccl().GetVersion || rccl().GetVersion(version) != 0) return ZKHIP_ERR_HIP;
    return ZKHIP_OK;
}
extern "C" int zkhip_comm_create_rccl(zkhip_ctx* c, const uint8_t* h_id128, uint32_t rank, uint32_t world, zkhip_comm** out) {
    if (!c || !h_id128 || !out || world == 0 || rank >= world) return ZKHIP_ERR_ARG;
    if (!is_pow2(world)) return ZKHIP_ERR_SHAPE;
    if (!rccl().ok) return ZKHIP_ERR_HIP;
    ZK_TRY(c->activate());
    zkhip_comm* m = new (std::nothrow) zkhip_comm();
    if (!m) return ZKHIP_ERR_NOMEM;
    m->c = c; m->rank_ = rank; m->world_ = world;
    RcclId id;
    std::memcpy(id.internal, h_id128, 128);
    if (rccl().CommInitRank(&m->nccl, (int)world, id, (int)rank) != 0 || !m->nccl) { delete m; return ZKHIP_ERR_HIP; }
    const int rc = m->setup_fault_buffers();
    if (rc != ZKHIP_OK) { zkhip_comm_destroy(m); return rc; }
    *out = m;
    return ZKHIP_OK;
}
extern "C" int zkhip_comm_destroy(zkhip_comm* m) {
    if (!m) return ZKHIP_ERR_ARG;
    int rc = ZKHIP_OK;
    if (m->c->activate() != ZKHIP_OK || hipStreamSynchronize(m->c->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    if (m->nccl && rccl().CommDestroy(m->nccl) != 0) rc = ZKHIP_ERR_HIP;
    for (auto& b : m->buf) if (b) (void)hipFree(b);
    if (m->poison_send) (void)hipFree(m->poison_send);
    if (m->poison_recv) (void)hipFree(m->poison_recv);
    if (m->peer_flag_h) (void)hipHostFree(m->peer_flag_h);
    if (m->solo) {
        for (auto& b : m->solo->buf) if (b) (void)hipFree(b);
        delete m->solo;
    }
    delete m;
    return rc;
}
extern "C" int zkhip_comm_all_gather(zkhip_comm* m, const void* d_send, void* d_recv, size_t bytes) {
    if (!m || !d_send || !d_recv) return ZKHIP_ERR_ARG;
    ZK_TRY(m->c->activate());
    return m->all_gather(d_send, d_recv, bytes);
}
// Test hook: the NEXT protocol run on this communicator behaves as if a step of this rank failed with `rc` in front of its exchange
// number `exchange_index` (0-based) -- the rank enters that and every later exchange with poison records and returns rc; every other
// rank returns ZKHIP_ERR_PEER.  exchange_index < 0 disarms.
extern "C" int zkhip_comm_inject_failure(zkhip_comm* m, int exchange_index, int rc) {
    if (!m || (exchange_index >= 0 && rc == 0)) return ZKHIP_ERR_ARG;
    m->inject_at = exchange_index;
    m->inject_rc = rc;
    return ZKHIP_OK;
}
extern "C" int zkhip_comm_stats(zkhip_comm* m, uint64_t* exchanges, uint64_t* bytes) {
    if (!m) return ZKHIP_ERR_ARG;
    if (exchanges) *exchanges = m->n_exchanges;
    if (bytes) *bytes = m->n_bytes;
    return ZKHIP_OK;
}
extern "C" int zkhip_comm_measure(zkhip_comm* m, size_t bytes, uint32_t iters, double* us_b2b, double* us_wait) {
    if (!m || !bytes || !iters || !us_b2b || !us_wait) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = m->c;
    ZK_TRY(c->activate());
    uint64_t* send = m->buffer(0, (bytes + 31) / 32);
    uint64_t* recv = m->buffer(1, ((bytes + 31) / 32) * m->world_);
    if (!send || !recv) return ZKHIP_ERR_NOMEM;
    ZK_HIP(c, hipMemsetAsync(send, 0, bytes, c->stream));
    const uint64_t ex0 = m->n_exchanges, by0 = m->n_bytes;
    // an exchange as the protocols issue it: the all-gather and, behind it, the look at every rank's record for the poison mark
    // (shard_protocol.hpp gather(): one tiny kernel; nothing where there is no transport)
    const size_t elems = (bytes + 31) / 32;
    auto exchange = [&]() -> int { ZK_TRY(m->all_gather(send, recv, bytes)); return m->check(recv, elems); };
    for (int w = 0; w < 3; ++w) ZK_TRY(exchange());             // warm-up (first use sets up RCCL's channels)
    ZK_TRY(c->wait_stream());
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    for (uint32_t i = 0; i < iters; ++i) ZK_TRY(exchange());
    ZK_TRY(c->wait_stream());
    *us_b2b = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / iters;
    t0 = clk::now();
    for (uint32_t i = 0; i < iters; ++i) { ZK_TRY(exchange()); ZK_TRY(c->wait_stream()); }
    *us_wait = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / iters;
    m->n_exchanges = ex0; m->n_bytes = by0;                                         // a measurement, not a prover's traffic
    (void)m->take_peer_failure();
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// Sumcheck::prove over a sharded table
// ---------------------------------------------------------------------------------------
// rank-local failures of a begin (memory, a busy workspace, the runtime) are carried into the protocol; argument and shape errors are the
// same on every rank (SPMD) and return at once
static inline bool rank_local_error(int rc) { return rc == ZKHIP_ERR_NOMEM || rc == ZKHIP_ERR_BUSY || rc == ZKHIP_ERR_HIP; }

static int sc_prove_sharded_impl(zkhip_sc_state* st, int begin_rc, size_t n_local, zkhip_comm* m, const uint64_t* h_claimed, uint64_t* h_sum,
                                 uint64_t* h_rp, uint64_t* h_ch, uint32_t* exchanges) {
    zkshard::HipScEngine e{st, m};
    e.n_local0 = n_local;
    uint32_t ex = 0;
    const int rc = zkshard::sumcheck_prove(e, *m, h_claimed, &ex, begin_rc);
    if (exchanges) *exchanges = ex;
    if (rc != ZKHIP_OK) {                      // a failed step must not leave the context's workspace lent
        if (st) zkhip_sc_abort(st);
        else (void)m->c->wait_stream();         // the poison exchanges of a rank without a session
        (void)m->take_peer_failure();
        return rc;
    }
    const int frc = zkhip_sc_finish(st, h_sum, h_rp, h_ch, nullptr);         // synchronises the stream: the sticky flag is final
    if (m->take_peer_failure()) return ZKHIP_ERR_PEER;
    return frc;
}
extern "C" int zkhip_sc_prove_sharded(zkhip_sc_state* st, zkhip_comm* m, const uint64_t* h_claimed, uint64_t* h_sum, uint64_t* h_rp,
                                      uint64_t* h_ch, uint32_t* exchanges) {
    if (!st) return ZKHIP_ERR_ARG;
    if (!m) { zkhip_sc_abort(st); return ZKHIP_ERR_ARG; }
    return sc_prove_sharded_impl(st, ZKHIP_OK, 0, m, h_claimed, h_sum, h_rp, h_ch, exchanges);
}
extern "C" int zkhip_sumcheck_prove_sharded(zkhip_comm* m, const uint64_t* d_local, size_t n_local, const uint64_t* h_claimed,
                                            uint64_t* h_sum, uint64_t* h_rp, uint64_t* h_ch, uint32_t* exchanges) {
    if (!m || !d_local || !h_sum) return ZKHIP_ERR_ARG;
    if (!is_pow2(n_local)) return ZKHIP_ERR_SHAPE;                 // Multilinear::new evaluation_form.rs:16-20
    const uint32_t rounds = log2_exact(n_local) + log2_exact(m->world_);
    if (rounds > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    if (rounds && (!h_rp || !h_ch)) return ZKHIP_ERR_ARG;
    zkhip_sc_state* st = nullptr;
    const int brc = zkhip_sc_begin(m->c, d_local, n_local, &st);
    if (brc != ZKHIP_OK && (!rank_local_error(brc) || m->passthrough())) return brc;
    return sc_prove_sharded_impl(brc == ZKHIP_OK ? st : nullptr, brc, n_local, m, h_claimed, h_sum, h_rp, h_ch, exchanges);
}

// ---------------------------------------------------------------------------------------
// composed / multi-composed provers over sharded tables
// ---------------------------------------------------------------------------------------
// the protocol over a session (st != nullptr) or, for a rank whose begin failed with begin_rc, over the shape `e` already holds
static int mc_prove_sharded_impl(zkshard::HipMcEngine& e, int begin_rc, int use_stages, uint32_t* h_lens, uint64_t* h_rp, uint64_t* h_ch,
                                 uint32_t* exchanges) {
    zkhip_comm* m = e.comm;
    uint32_t ex = 0;
    const bool stages = use_stages < 0 ? m->world_ > 1 : use_stages != 0;
    const int rc = zkshard::composed_prove(e, *m, stages, &ex, begin_rc);
    if (exchanges) *exchanges = ex;
    if (rc != ZKHIP_OK) {
        if (e.st) zkhip_mc_abort(e.st);
        else (void)m->c->wait_stream();
        (void)m->take_peer_failure();
        return rc;
    }
    const int frc = zkhip_mc_finish(e.st, h_lens, h_rp, h_ch);
    // nothing collected (gkr.hip chains two sessions per layer): finish did not wait for the stream, so the sticky flag is not final --
    // it is read by the next call on this communicator that does collect
    if (!h_rp && !h_ch) return frc;
    if (m->take_peer_failure()) return ZKHIP_ERR_PEER;
    return frc;
}
extern "C" int zkhip_mc_prove_sharded(zkhip_mc_state* st, zkhip_comm* m, int use_stages, uint32_t* h_lens, uint64_t* h_rp, uint64_t* h_ch,
                                      uint32_t* exchanges) {
    if (!st) return ZKHIP_ERR_ARG;
    if (!m) { zkhip_mc_abort(st); return ZKHIP_ERR_ARG; }
    zkshard::HipMcEngine e{st, m};
    const int src = e.shape_of_session();
    if (src != ZKHIP_OK) { zkhip_mc_abort(st); return src; }
    return mc_prove_sharded_impl(e, ZKHIP_OK, use_stages, h_lens, h_rp, h_ch, exchanges);
}
static int mc_begin_and_prove(zkhip_comm* m, const uint64_t* const* ptrs, const uint32_t* term_sizes, uint32_t n_terms, size_t n_local, int multi,
                              const uint64_t* h_sum, int use_stages, uint32_t* h_lens, uint64_t* h_rp, uint64_t* h_ch, uint32_t* exchanges) {
    zkhip_mc_state* st = nullptr;
    const int brc = zkhip_mc_begin(m->c, ptrs, term_sizes, n_terms, n_local, m->world_, multi, h_sum, &st);
    if (brc != ZKHIP_OK && (!rank_local_error(brc) || m->passthrough())) return brc;
    zkshard::HipMcEngine e{brc == ZKHIP_OK ? st : nullptr, m};
    const int src = e.st ? e.shape_of_session() : e.shape(term_sizes, n_terms, 0, n_local);
    if (src != ZKHIP_OK) { if (e.st) zkhip_mc_abort(e.st); return src; }
    return mc_prove_sharded_impl(e, brc, use_stages, h_lens, h_rp, h_ch, exchanges);
}
extern "C" int zkhip_composed_prove_sharded(zkhip_comm* m, const uint64_t* const* ptrs, uint32_t k, size_t n_local, int use_stages,
                                            uint64_t* h_rp, uint64_t* h_ch, uint32_t* exchanges) {
    if (!m || !ptrs || !h_rp || !h_ch) return ZKHIP_ERR_ARG;
    return mc_begin_and_prove(m, ptrs, &k, 1, n_local, 0, nullptr, use_stages, nullptr, h_rp, h_ch, exchanges);
}
extern "C" int zkhip_multi_composed_prove_sharded(zkhip_comm* m, const uint64_t* const* ptrs, const uint32_t* term_sizes, uint32_t n_terms,
                                                  size_t n_local, const uint64_t* h_sum, int use_stages, uint32_t* h_lens, uint64_t* h_rp,
                                                  uint64_t* h_ch, uint32_t* exchanges) {
    if (!m || !ptrs || !term_sizes || !h_sum || !h_lens || !h_rp || !h_ch) return ZKHIP_ERR_ARG;
    return mc_begin_and_prove(m, ptrs, term_sizes, n_terms, n_local, 1, h_sum, use_stages, h_lens, h_rp, h_ch, exchanges);
}

// ---------------------------------------------------------------------------------------
// KZG commit over (scalars, SRS) sharded across the ranks: sub-MSM, one all-gather of 128-byte records, group sum on every rank
// ---------------------------------------------------------------------------------------
extern "C" int zkhip_kzg_commit_sharded(zkhip_comm* m, const uint64_t* d_points_xy, const void* d_table, const uint8_t* d_points_inf,
                                        size_t n_points, const uint64_t* d_scalars, size_t n_scalars, int require_equal_len,
                                        uint64_t* h_out_xy, uint8_t* h_out_inf) {
    if (!m || !h_out_xy || !h_out_inf || (!d_points_xy == !d_table)) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = m->c;
    uint64_t rec[16] = {};
    uint8_t inf = 0;
    int local_rc = d_table ? zkhip_kzg_commit_table(c, d_table, d_points_inf, n_points, d_scalars, n_scalars, require_equal_len, rec, &inf)
                           : zkhip_kzg_commit(c, d_points_xy, d_points_inf, n_points, d_scalars, n_scalars, require_equal_len, rec, &inf);
    if (local_rc == ZKHIP_OK && m->inject(0) != 0) local_rc = m->inject_rc;     // test hook (zkhip_comm_inject_failure)
    if (m->world_ == 1) {
        if (local_rc != ZKHIP_OK) return local_rc;
        std::memcpy(h_out_xy, rec, 96);
        *h_out_inf = inf;
        ++m->n_exchanges; m->n_bytes += 128;      // counted as the protocol's one exchange (a copy on one rank)
        return ZKHIP_OK;
    }
    // the record: x | y | infinity flag | STATUS (word 13): a rank whose sub-commit failed still enters the one exchange, with a zero
    // payload and its error there, and every rank returns -- the failed one its own error, the others ZKHIP_ERR_PEER
    if (local_rc != ZKHIP_OK) std::memset(rec, 0, sizeof(rec));
    rec[12] = inf;
    rec[13] = (uint64_t)(uint32_t)local_rc;
    const uint32_t world = m->world_;
    if (16 * (size_t)(world + 1) > (size_t)(ZK_PIN_END - ZK_PIN_PROOF)) return ZKHIP_ERR_SHAPE;
    uint64_t* pin = c->pinned_u64(ZK_PIN_PROOF);          // the proof staging area: no prover runs during a commit of the same context
    std::memcpy(pin, rec, 128);
    const bool own = local_rc == ZKHIP_OK;
    uint64_t* send = own ? m->buffer(0, 4) : (uint64_t*)m->poison_recv;              // (a failed rank touches no allocator: the reserve)
    uint64_t* recv = own ? m->buffer(1, 4 * (size_t)world) : (uint64_t*)m->poison_recv + 4 * 4;
    if (!send || !recv) {
        if (!m->poison_recv) return ZKHIP_ERR_NOMEM;
        local_rc = ZKHIP_ERR_NOMEM;
        pin[13] = (uint64_t)(uint32_t)local_rc;
        std::memset(pin, 0, 96);
        send = (uint64_t*)m->poison_recv;
        recv = (uint64_t*)m->poison_recv + 4 * 4;
    }
    ZK_HIP(c, hipMemcpyAsync(send, pin, 128, hipMemcpyHostToDevice, c->stream));
    ZK_TRY(m->all_gather(send, recv, 128));
    ZK_HIP(c, hipMemcpyAsync(pin + 16, recv, 128 * (size_t)world, hipMemcpyDeviceToHost, c->stream));
    ZK_TRY(c->wait_stream());
    if (local_rc != ZKHIP_OK) return local_rc;
    std::vector<uint64_t> xy(12 * (size_t)world);
    std::vector<uint8_t> infs(world);
    for (uint32_t g = 0; g < world; ++g) {
        if (pin[16 + 16 * (size_t)g + 13] != 0) return ZKHIP_ERR_PEER;
        std::memcpy(&xy[12 * (size_t)g], pin + 16 + 16 * (size_t)g, 96);
        infs[g] = (uint8_t)(pin[16 + 16 * (size_t)g + 12] != 0);
    }
    return zkhip_g1_sum_affine(xy.data(), infs.data(), world, h_out_xy, h_out_inf);
}
