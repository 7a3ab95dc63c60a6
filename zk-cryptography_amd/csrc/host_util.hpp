// host_util.hpp -- small host helpers shared by the translation units of libzkhip
#pragma once
#include <stddef.h>
#include <stdint.h>
static inline uint32_t log2_exact(size_t n) {
    uint32_t k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}
static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
// optional arguments of zk_multi_composed_enqueue (composed.hip; internal, used by gkr.hip): see there
struct ZkMcExtra {
    const uint64_t* d_sum;        // the claimed sum, in device memory
    void* outer;                  // OuterDev* (composed_kernels.hpp): an outer transcript fed beside the rounds
    uint32_t token;               // the value the rounds' flags are raised to
    uint64_t* d_round_polys;      // 64 u64 per round, instead of the context's small buffer
    uint64_t* d_challenges;       // 4 u64 per round
};

#ifdef __cplusplus
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
// A few host threads kept for the life of a context: the per-problem host epilogues of a batched commit (~0.25 ms of serial point
// arithmetic each, twenty of them behind MultilinearKZG::open) start within microseconds, where twenty std::thread constructions cost as much
// as the work.  run(n, fn) calls fn(0) .. fn(n - 1), each exactly once, on the workers and the calling thread, and returns when all are done.
// Every run() owns a JOB object (function, count, claim counter, pending counter) that the workers reach through a shared_ptr copied under
// the mutex: a worker that wakes late -- after its run has returned, possibly while the next one is being set up -- only ever touches the
// finished job it copied (whose claim counter is exhausted, so it never calls through the job's function pointer) or finds none.
class ZkHostPool {
public:
    explicit ZkHostPool(unsigned n_workers) {
        for (unsigned i = 0; i < n_workers; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~ZkHostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++generation_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    ZkHostPool(const ZkHostPool&) = delete;
    ZkHostPool& operator=(const ZkHostPool&) = delete;
    void run(unsigned n, const std::function<void(unsigned)>& fn) {
        if (n == 0) return;
        auto job = std::make_shared<Job>();
        job->fn = &fn; job->n = n;
        job->pending.store(n);                       // before the job becomes visible to any worker
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = job; ++generation_;
        }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return job->pending.load() == 0; });
        if (job_ == job) job_.reset();               // late wakers find no job
    }
private:
    struct Job {
        const std::function<void(unsigned)>* fn = nullptr;
        unsigned n = 0;
        std::atomic<unsigned> next{0}, pending{0};
    };
    void work(Job& j) {
        for (;;) {
            const unsigned i = j.next.fetch_add(1);
            if (i >= j.n) break;                     // an exhausted (possibly finished) job: fn is never touched
            (*j.fn)(i);                              // i < n was claimed here alone, so run() is still waiting for it: fn is alive
            if (j.pending.fetch_sub(1) == 1) {       // the last task: wake run() (under the mutex, so the wake-up cannot be lost)
                std::lock_guard<std::mutex> lk(m_);
                done_cv_.notify_all();
            }
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::shared_ptr<Job> job_;
    unsigned long generation_ = 0;
    bool stop_ = false;
};
#endif
