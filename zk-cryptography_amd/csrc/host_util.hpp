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

#ifdef __cplusplus
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
// A few host threads kept for the life of a context: the per-problem host epilogues of a batched commit (~0.25 ms of serial point
// arithmetic each, twenty of them behind MultilinearKZG::open) start within microseconds, where twenty std::thread constructions cost as much
// as the work.  run(n, fn) calls fn(0) .. fn(n - 1), each exactly once, on the workers and the calling thread, and returns when all are done.
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
    void run(unsigned n, const std::function<void(unsigned)>& fn) {
        if (n == 0) return;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; next_.store(0); pending_.store(n); ++generation_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return pending_.load() == 0 && active_ == 0; });
        fn_ = nullptr;
    }
private:
    void work() {
        for (;;) {
            const unsigned i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
            pending_.fetch_sub(1);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                ++active_;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                --active_;
            }
            done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned n_ = 0, active_ = 0;
    unsigned long generation_ = 0;
    std::atomic<unsigned> next_{0}, pending_{0};
    bool stop_ = false;
};
#endif
