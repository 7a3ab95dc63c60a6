// ctx.hpp -- per-device context of libzkhip: stream, workspaces, event-based kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <vector>

#include "../../include/zkhip.h"
#include "host_util.hpp"

#define ZK_MAX_ROUNDS 48          /* rounds of one sumcheck: tables of 2^48 entries are far beyond 288 GB; a GKR layer of width 2^24 takes 2 x 24 */
#define ZK_MAX_PARTIALS 4096      /* (lo, hi) pairs: >= the largest grid any reducing kernel uses */

// fixed carve-up of the small device scratch (offsets in uint64_t units)
enum : size_t {
    ZK_SMALL_R = 0,                                           // 4:  one field element (fold point / scalar)
    ZK_SMALL_RES = 16,                                        // 16: small results
    ZK_SMALL_STATE = 64,                                      // 64: SumcheckDev / transcript state
    ZK_SMALL_PTS = 256,                                       // 4*ZK_MAX_ROUNDS evaluation points
    ZK_SMALL_CHALLENGES = ZK_SMALL_PTS + 4 * ZK_MAX_ROUNDS,   // 4*ZK_MAX_ROUNDS
    ZK_SMALL_ROUNDPOLYS = ZK_SMALL_CHALLENGES + 4 * ZK_MAX_ROUNDS,   // up to 8 coefficients x 2 words... 64*ZK_MAX_ROUNDS
    ZK_SMALL_PARTIALS = ZK_SMALL_ROUNDPOLYS + 64 * ZK_MAX_ROUNDS,    // 8 * ZK_MAX_PARTIALS * 8 (composed: up to 8 sums/block)
    ZK_SMALL_END = ZK_SMALL_PARTIALS + 4 * 8 * ZK_MAX_PARTIALS
};
#define ZK_SMALL_BYTES (ZK_SMALL_END * 8)

// pinned host staging (uint64_t units)
enum : size_t {
    ZK_PIN_R = 0,
    ZK_PIN_RES = 16,
    ZK_PIN_PTS = 64,
    ZK_PIN_PROOF = ZK_PIN_PTS + 4 * ZK_MAX_ROUNDS,
    ZK_PIN_END = ZK_PIN_PROOF + 4 + 8 * ZK_MAX_ROUNDS + 4 * ZK_MAX_ROUNDS + 64 * ZK_MAX_ROUNDS
};
#define ZK_PINNED_BYTES (ZK_PIN_END * 8)

#define ZK_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t _e = (call);                             \
        if (_e != hipSuccess) {                             \
            (ctx)->last_hip = (int)_e;                      \
            return ZKHIP_ERR_HIP;                           \
        }                                                   \
    } while (0)
#define ZK_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s != ZKHIP_OK) return _s; \
    } while (0)

struct ZkProfEvent { hipEvent_t start, stop; };
struct ZkProfRecord { const char* name; size_t event; double bytes; };

struct zkhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip = 0;
    void* d_ws = nullptr;       // large workspace (ping-pong tables, MSM buckets, ...)
    size_t ws_bytes = 0;
    void* d_fingerprint = nullptr;   // zkhip_srs_fingerprint's 52 words, its stream (highest priority) and the event that orders it behind the caller's
    hipStream_t guard_stream = nullptr;
    hipEvent_t guard_ev = nullptr;
    void* d_composed = nullptr;      // composed provers: ComposedDev (transcript + interpolation matrices of every degree), uploaded once
    void* d_gen_table = nullptr;   // SRS generation: d * 2^(8w) * G for 32 windows x 255 digits, affine (+ infinity flags); built on first use
    void* ntt_state = nullptr;     // twiddle tables and pass plans of the transforms this context has run (ntt.hip); ntt_free releases them
    void (*ntt_free)(void*) = nullptr;
    void* d_aux = nullptr;      // second grow-only buffer for entry points that call others which own d_ws (kzg_open)
    uint32_t outer_token = 0;   // sessions that feed an outer transcript (composed_kernels.hpp): a value no earlier session's flags hold
    size_t aux_bytes = 0;
    // pinned result buffers + events for commits whose host epilogue is deferred (msm_enqueue / msm_finish), and the side
    // streams on which MultilinearKZG::open runs its per-round commits next to each other
    static constexpr int MSM_SLOTS = 6;
    void* msm_pin[MSM_SLOTS] = {};
    size_t msm_pin_bytes[MSM_SLOTS] = {};
    hipEvent_t msm_ev[MSM_SLOTS] = {};
    hipStream_t side[MSM_SLOTS] = {};
    hipEvent_t fork_ev = nullptr, join_ev = nullptr, serial_ev = nullptr;
    int ensure_side_streams() {
        for (int i = 0; i < MSM_SLOTS; ++i)
            if (!side[i] && hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking) != hipSuccess) return ZKHIP_ERR_HIP;
        if (!fork_ev && hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
        if (!join_ev && hipEventCreateWithFlags(&join_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
        if (!serial_ev && hipEventCreateWithFlags(&serial_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
        return ZKHIP_OK;
    }
    // The basic prover's streaming fold runs on a LOW-priority stream of its own next to the serial transcript kernel
    // (the dispatcher prefers the serial kernel's single workgroup whenever both are ready).  Measured alternatives
    // (DESIGN.md section 5): a CU-masked fold stream that leaves one CU to the serial kernel -- such streams can only be
    // created "blocking", and the implicit synchronisation with the NULL stream (PyTorch's default) cost 30-60 us per
    // fork / join; sharing a CU with fold waves -- the transcript wave runs at half speed.
    hipStream_t fold_stream = nullptr;
    int ensure_fold_stream() {
        if (fold_stream) return ensure_side_streams();
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return ZKHIP_ERR_HIP;
        if (hipStreamCreateWithPriority(&fold_stream, hipStreamNonBlocking, least) != hipSuccess) return ZKHIP_ERR_HIP;
        return ensure_side_streams();
    }
    // coarse block sums left by zkhip_mle_block_sums for the prover's first rounds: a ring (canonical, then Montgomery: 64 KiB
    // each; more entries than proofs in flight), so that the sums of a table whose proof is in flight survive the poly_sum() of the next tables
    static constexpr int COARSE_RING = 16;
    void* d_coarse[COARSE_RING] = {};
    const void* coarse_of[COARSE_RING] = {}; size_t coarse_n[COARSE_RING] = {}; uint32_t coarse_k1[COARSE_RING] = {};
    int coarse_owner[COARSE_RING] = {};     // result slot + 1 of the proof that still reads the entry (0: nobody): such an entry is never handed out
    int coarse_next = 0;
    int next_coarse(int* slot) {
        for (int tries = 0; tries < COARSE_RING; ++tries) {
            const int k = coarse_next;
            coarse_next = (coarse_next + 1) % COARSE_RING;
            if (coarse_owner[k]) continue;          // a proof in flight reads it from a stream of its own: no event orders a rewrite behind that
            if (!d_coarse[k] && hipMalloc(&d_coarse[k], 2 * 1024 * 32) != hipSuccess) return ZKHIP_ERR_NOMEM;
            coarse_of[k] = nullptr;
            *slot = k;
            return ZKHIP_OK;
        }
        return ZKHIP_ERR_BUSY;                       // unreachable with <= PROOF_SLOTS proofs in flight (one entry each), kept as a guard
    }
    void release_coarse(int proof_slot) {
        for (int k = 0; k < COARSE_RING; ++k) if (coarse_owner[k] == proof_slot + 1) coarse_owner[k] = 0;
    }
    // Sumcheck::prove in flight (zkhip_sumcheck_prove_begin): every ticket has a LANE of its own -- events, workspace and small scratch, a
    // high-priority serial stream and a low-priority fold stream -- so that the streaming passes of one proof run while the transcript rounds
    // of the others hash (the synchronous call keeps the caller's stream and the context's buffers).  Lanes k and k + 4 -- four proofs apart,
    // the older one is done when the younger one begins -- share their two streams: the runtime serves all streams of one priority from four
    // hardware queues, and two streams that take turns in one queue wait for each other's kernels (profiles/r06/NOTES.md section 7: with six
    // serial streams on four queues the first rounds of a proof stood behind another lane's 100 us serial kernel)
    struct ProofLane {
        hipStream_t serial = nullptr, fold = nullptr;
        bool borrowed = false;                  // the streams are those of lane k - 4
        hipEvent_t begin_ev = nullptr, fork_ev = nullptr, serial_ev = nullptr;
        void* ws = nullptr; size_t ws_bytes = 0;
        void* small = nullptr;
    };
    static constexpr int PROOF_SLOTS = 8;      // proofs in flight (measured at 2^24 with 2 .. 12, see bench.py `pipelined`; nothing is gained beyond 8)
    static constexpr int LANE_STREAMS = 4;     // = the hardware queues of one stream priority
    ProofLane lanes[PROOF_SLOTS];
    // With three or more proofs in flight the big fold of a proof is NOT enqueued when the proof begins: it goes onto the CALLER's stream --
    // where poly_sum() puts every table's sums pass -- behind the sums passes of the next one to three tables, so that ONE stream carries
    // all streaming passes back to back (two passes side by side take 2.5 x as long as one, not 2 x) and never waits for a proof's first
    // rounds.  `deferred` holds those second halves, oldest first; they are enqueued by the next zkhip_sumcheck_prove_begin, by
    // zkhip_sumcheck_prove_end of that proof (or of a younger one), and by everything that drains or replaces the caller's stream.
    std::deque<std::pair<int, std::function<int()>>> deferred;      // (proof slot, what is left to enqueue)
    int deferred_rc[PROOF_SLOTS] = {};                              // a second half that could not be enqueued: reported by prove_end of that proof
    // enqueue the oldest entries until `keep` are left, or -- until_slot >= 0 -- until that proof's entry has gone (nothing if it has already)
    int flush_deferred(size_t keep = 0, int until_slot = -1) {
        if (until_slot >= 0) {
            bool there = false;
            for (auto& d : deferred) there = there || d.first == until_slot;
            if (!there) return ZKHIP_OK;
        }
        int rc = ZKHIP_OK;
        while (deferred.size() > keep) {
            const int slot = deferred.front().first;
            std::function<int()> f = std::move(deferred.front().second);
            deferred.pop_front();
            const int r = f();
            if (r != ZKHIP_OK) { deferred_rc[slot] = r; if (rc == ZKHIP_OK) rc = r; }
            if (slot == until_slot) break;
        }
        return rc;
    }
    int ensure_lane_streams(int k) {
        ProofLane& L = lanes[k];
        if (L.serial) return ZKHIP_OK;
        if (k >= LANE_STREAMS) {
            const int rc = ensure_lane_streams(k - LANE_STREAMS);
            if (rc != ZKHIP_OK) return rc;
            L.serial = lanes[k - LANE_STREAMS].serial; L.borrowed = true;
            return ZKHIP_OK;
        }
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return ZKHIP_ERR_HIP;
        if (hipStreamCreateWithPriority(&L.serial, hipStreamNonBlocking, greatest) != hipSuccess) return ZKHIP_ERR_HIP;
        // (measured and dropped, profiles/r06/NOTES.md: some lanes' serial stream at normal instead of high priority, or their fold stream at
        // normal instead of low: 0.27-0.32 ms per proof in flight instead of 0.23; the serial kernels must get in front of every streaming pass)
        return ZKHIP_OK;
    }
    // A lane's low-priority fold stream (only a proof that begins with at most one other in flight keeps its big fold on a stream of its
    // own: tickets 0 and 1) -- created WITH the lane all the same, for lanes 0..3: the runtime serves the streams of one priority from at
    // most four hardware queues, gives a new stream the queue with the fewest streams on it, the FIRST such queue on a tie -- and a queue it
    // has just created for one stream ties with every other single-stream queue.  With the four fold streams the low-priority pool is full
    // when the lanes exist (as the high-priority one is with the four serial streams), and lanes created later by somebody else -- the GKR
    // batch's -- spread over the queues: with fold streams on demand two of its eight lanes landed on ONE queue in bench.py's process
    // (46 % busy each, the others 80-90 %: 0.47 instead of 0.34 ms per depth-8 proof; profiles/r06/NOTES.md).  The price: the synchronous
    // proof's fork onto its own fold stream costs 4 us more once the lanes exist (five low-priority streams on four queues, tools/ab_lanes.py).
    int ensure_lane_fold(int k) {
        ProofLane& L = lanes[k];
        if (L.fold) return ZKHIP_OK;
        if (k >= LANE_STREAMS) {
            const int rc = ensure_lane_fold(k - LANE_STREAMS);
            if (rc != ZKHIP_OK) return rc;
            L.fold = lanes[k - LANE_STREAMS].fold;
            return ZKHIP_OK;
        }
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return ZKHIP_ERR_HIP;
        if (hipStreamCreateWithPriority(&L.fold, hipStreamNonBlocking, least) != hipSuccess) return ZKHIP_ERR_HIP;
        return ZKHIP_OK;
    }
    int ensure_lane(int k, size_t ws_need) {
        ProofLane& L = lanes[k];
        { const int rc = ensure_lane_streams(k); if (rc != ZKHIP_OK) return rc; }
        { const int rc = ensure_lane_fold(k); if (rc != ZKHIP_OK) return rc; }
        if (!L.small) {
            if (hipEventCreateWithFlags(&L.begin_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
            if (hipEventCreateWithFlags(&L.fork_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
            if (hipEventCreateWithFlags(&L.serial_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
            if (hipMalloc(&L.small, ZK_SMALL_BYTES) != hipSuccess) return ZKHIP_ERR_NOMEM;
        }
        if (ws_need > L.ws_bytes) {                 // grow-only; the lane is idle here (its ticket is free)
            if (L.ws) { if (hipStreamSynchronize(L.serial) != hipSuccess || (L.fold && hipStreamSynchronize(L.fold) != hipSuccess)) return ZKHIP_ERR_HIP; hipFree(L.ws); }
            L.ws = nullptr; L.ws_bytes = 0;
            if (hipMalloc(&L.ws, ws_need) != hipSuccess) return ZKHIP_ERR_NOMEM;
            L.ws_bytes = ws_need;
        }
        return ZKHIP_OK;
    }
    int reserve_msm_pin(int slot, size_t bytes) {
        if (!msm_ev[slot] && hipEventCreateWithFlags(&msm_ev[slot], hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
        if (bytes <= msm_pin_bytes[slot]) return ZKHIP_OK;
        if (msm_pin[slot]) hipHostFree(msm_pin[slot]);
        msm_pin[slot] = nullptr; msm_pin_bytes[slot] = 0;
        if (hipHostMalloc(&msm_pin[slot], bytes, hipHostMallocDefault) != hipSuccess) return ZKHIP_ERR_NOMEM;
        msm_pin_bytes[slot] = bytes;
        return ZKHIP_OK;
    }
    // geometry tables of the commit in slot `slot` (msm_build_geometry): device copy + pinned staging, kept while the geometry is the same
    void* msm_tab_dev[MSM_SLOTS] = {};
    void* msm_tab_pin[MSM_SLOTS] = {};
    size_t msm_tab_bytes[MSM_SLOTS] = {};
    std::shared_ptr<const void> msm_tab_geo[MSM_SLOTS];   // the geometry whose tables the device copy holds (kept alive: its address is its identity)
    int reserve_msm_tab(int slot, size_t bytes) {
        if (bytes <= msm_tab_bytes[slot]) return ZKHIP_OK;
        if (msm_tab_dev[slot]) hipFree(msm_tab_dev[slot]);
        if (msm_tab_pin[slot]) hipHostFree(msm_tab_pin[slot]);
        msm_tab_dev[slot] = nullptr; msm_tab_pin[slot] = nullptr; msm_tab_bytes[slot] = 0; msm_tab_geo[slot].reset();
        const size_t cap = (bytes + 65535) & ~(size_t)65535;
        if (hipMalloc(&msm_tab_dev[slot], cap) != hipSuccess) return ZKHIP_ERR_NOMEM;
        if (hipHostMalloc(&msm_tab_pin[slot], cap, hipHostMallocDefault) != hipSuccess) return ZKHIP_ERR_NOMEM;
        msm_tab_bytes[slot] = cap;
        return ZKHIP_OK;
    }
    ZkHostPool* host_pool = nullptr;     // host epilogues of batched commits (created on first use, min(hardware threads, 32) - 1 workers)
    ZkHostPool* pool() {
        if (!host_pool) {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            host_pool = new (std::nothrow) ZkHostPool(std::min(hw, 32u) - 1);
        }
        return host_pool;
    }
    std::vector<zkhip_ctx*> gkr_lanes;   // child contexts of zkhip_gkr_prove_batch (a stream, scratch and transcript state each), destroyed with this one
    // a lane replays a circuit's launch chain as a HIP graph (gkr.hip): its own copy of the layer values + the graph and the addresses it holds
    bool gkr_lane = false;
    zkhip_ctx* gkr_parent = nullptr;            // a lane's owner: the lanes' proofs that allocate or record a graph take turns under its gkr_warm_mu
    std::mutex gkr_warm_mu;
    void* d_gkr_in = nullptr; size_t gkr_in_bytes = 0;
    struct GkrGraph {
        void* exec = nullptr;                       // hipGraphExec_t
        const void *cir = nullptr, *aux = nullptr, *in = nullptr, *ws = nullptr, *pin = nullptr, *composed = nullptr;
        const void *warm_cir = nullptr, *warm_aux = nullptr, *warm_ws = nullptr;      // a plain proof of this circuit has run here (allocations made)
    } gkr_graph;
    bool ws_lent = false;       // the workspace currently backs a split-phase prover state or commits in flight
    // commits in flight (zkhip_kzg_commit_begin / _end): two slots, each with a region of the workspace, a side stream and a
    // pinned result buffer of its own; `async_pend` is an MsmPending allocated by msm.hip
    static constexpr int ASYNC_SLOTS = 3;   // commits in flight (zkhip_kzg_commit_begin); measured with the pipeline filling and draining inside the timed region: 2 / 3 / 4 slots = 3.03 / 2.91 / 2.98 ms per 2^20-point commit
    void* async_pend[ASYNC_SLOTS] = {};
    size_t async_region = 0;
    // one cached set of small split-phase buffers, so that a steady stream of sharded proves never allocates
    void* sc_small = nullptr; void* sc_stage = nullptr; size_t sc_stage_cap = 0; bool sc_lent = false;
    void* d_small = nullptr;    // fixed small scratch, layout above
    void* h_pinned = nullptr;
    bool profiling = false;
    std::vector<ZkProfEvent> prof_events;
    size_t prof_used = 0;
    std::vector<ZkProfRecord> prof_records;

    int activate() {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        return ZKHIP_OK;
    }
    std::set<const void*> big_lds_done;
    // kernels that carve more than 64 KiB of the CU's 160 KiB LDS need the attribute raised once per function
    int allow_big_lds(const void* fn, size_t bytes) {
        if (bytes <= 64 * 1024 || big_lds_done.count(fn)) return ZKHIP_OK;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        big_lds_done.insert(fn);
        return ZKHIP_OK;
    }
    // Host wait for a stream's work so far: polls an event for up to ~2 ms (a prover call is a few hundred
    // microseconds; a blocking wait's wake-up costs 10-20 us of idle GPU before the next call), then blocks.
    hipEvent_t done_ev = nullptr;
    int wait_stream(hipStream_t s = nullptr) {
        if (!s) s = stream;
        if (!done_ev && hipEventCreateWithFlags(&done_ev, hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
        if (hipEventRecord(done_ev, s) != hipSuccess) return ZKHIP_ERR_HIP;
        if (s != stream && hipStreamWaitEvent(stream, done_ev, 0) != hipSuccess) return ZKHIP_ERR_HIP;   // the caller's stream stays ordered behind it
        for (int spin = 0; spin < 200000; ++spin) {
            const hipError_t e = hipEventQuery(done_ev);
            if (e == hipSuccess) return ZKHIP_OK;
            if (e != hipErrorNotReady) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        }
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        return ZKHIP_OK;
    }
    int wait_event(hipEvent_t ev) {      // the same for an event already recorded
        for (int spin = 0; spin < 200000; ++spin) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return ZKHIP_OK;
            if (e != hipErrorNotReady) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        }
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        return ZKHIP_OK;
    }
    // result slots of the basic prover: pinned copies of [state .. round polynomials] + the event their copy completes at;
    // proof_pending[k] = n_vars of the proof in flight in slot k (0: free)
    void* proof_pin[PROOF_SLOTS] = {};
    hipEvent_t proof_ev[PROOF_SLOTS] = {};
    uint32_t proof_pending[PROOF_SLOTS] = {};
    int ensure_proof_slot(int k) {
        if (!proof_ev[k] && hipEventCreateWithFlags(&proof_ev[k], hipEventDisableTiming) != hipSuccess) return ZKHIP_ERR_HIP;
        if (!proof_pin[k] && hipHostMalloc(&proof_pin[k], ((ZK_SMALL_ROUNDPOLYS - ZK_SMALL_STATE) + 8 * ZK_MAX_ROUNDS) * 8, hipHostMallocDefault) != hipSuccess)
            return ZKHIP_ERR_NOMEM;
        return ZKHIP_OK;
    }
    uint64_t* small_u64(size_t off) { return (uint64_t*)d_small + off; }
    uint64_t* pinned_u64(size_t off) { return (uint64_t*)h_pinned + off; }
    // grow-only workspace; growth synchronises (never inside a steady-state timed loop)
    int reserve_aux(size_t bytes) {
        if (bytes <= aux_bytes) return ZKHIP_OK;
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        if (d_aux) hipFree(d_aux);
        d_aux = nullptr;
        aux_bytes = 0;
        e = hipMalloc(&d_aux, bytes);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_NOMEM; }
        aux_bytes = bytes;
        return ZKHIP_OK;
    }
    int reserve_ws(size_t bytes) {
        if (ws_lent) return ZKHIP_ERR_BUSY;     // a live split-phase session owns it: neither overwrite nor free it
        if (bytes <= ws_bytes) return ZKHIP_OK;
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_HIP; }
        if (d_ws) hipFree(d_ws);
        d_ws = nullptr;
        ws_bytes = 0;
        e = hipMalloc(&d_ws, bytes);
        if (e != hipSuccess) { last_hip = (int)e; return ZKHIP_ERR_NOMEM; }
        ws_bytes = bytes;
        return ZKHIP_OK;
    }
};

// Brackets one kernel launch with HIP events on the context's stream when profiling is on.
struct ProfScope {
    zkhip_ctx* c;
    size_t idx = 0;
    bool on;
    hipStream_t s;
    ProfScope(zkhip_ctx* ctx, const char* name, double bytes, hipStream_t on_stream = nullptr)
        : c(ctx), on(ctx->profiling), s(on_stream ? on_stream : ctx->stream) {
        if (!on) return;
        if (c->prof_used == c->prof_events.size()) {
            ZkProfEvent ev;
            hipEventCreate(&ev.start);
            hipEventCreate(&ev.stop);
            c->prof_events.push_back(ev);
        }
        idx = c->prof_used++;
        hipEventRecord(c->prof_events[idx].start, s);
        c->prof_records.push_back({name, idx, bytes});
    }
    ~ProfScope() {
        if (on) hipEventRecord(c->prof_events[idx].stop, s);
    }
};
