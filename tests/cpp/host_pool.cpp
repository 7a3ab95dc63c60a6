// The context's host thread pool (csrc/host_util.hpp: ZkHostPool, the per-problem epilogues of a batched commit): run(n, fn) must call
// fn(0) .. fn(n - 1) exactly once each and return only when all are done -- with no workers, with a few, and for many runs in a row.
#include "../../zk-cryptography_amd/csrc/host_util.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const unsigned workers = argc > 1 ? (unsigned)std::atoi(argv[1]) : 3;
    const unsigned runs = argc > 2 ? (unsigned)std::atoi(argv[2]) : 2000;
    ZkHostPool pool(workers);
    unsigned long long total = 0;
    for (unsigned r = 0; r < runs; ++r) {
        const unsigned n = (r * 7 + 1) % 41;            // 0 .. 40 tasks, more and fewer than threads
        std::vector<std::atomic<int>> hits(n ? n : 1);
        for (auto& h : hits) h.store(0);
        std::atomic<unsigned long long> sum{0};
        pool.run(n, [&](unsigned i) { hits[i].fetch_add(1); sum.fetch_add(i + 1); });
        for (unsigned i = 0; i < n; ++i)
            if (hits[i].load() != 1) { std::printf("run %u task %u ran %d times\n", r, i, hits[i].load()); return 1; }
        if (sum.load() != (unsigned long long)n * (n + 1) / 2) { std::printf("run %u: sum mismatch\n", r); return 1; }
        total += n;
    }
    std::printf("ok %u workers %u runs %llu tasks\n", workers, runs, total);
    return 0;
}
