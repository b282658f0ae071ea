// csrc/host_pool.h under ThreadSanitizer: several caller threads (as several contexts would) run loops of different
// sizes on the one pool at once; every part of every loop must run exactly once, a part that throws must make run()
// return false without taking the process down, and no data race may be reported.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>

#include "host_pool.h"

int main() {
    mic::HostPool &pool = mic::HostPool::get();
    std::atomic<long> bad{0}, loops{0};
    auto caller = [&](int seed) {
        for (int round = 0; round < 300; ++round) {
            const int n = 1 + (seed * 7 + round * 13) % 97;
            std::vector<std::atomic<int>> hits((size_t)n);
            for (auto &h : hits) h = 0;
            const bool throws = round % 50 == 49;
            const bool ok = pool.run(n, [&](int i) {
                hits[(size_t)i].fetch_add(1);
                if (throws && i == n / 2) throw std::runtime_error("part failed");
            });
            for (auto &h : hits)
                if (h.load() != 1) ++bad;
            if (ok == throws) ++bad;
            ++loops;
        }
    };
    std::vector<std::thread> callers;
    for (int t = 0; t < 4; ++t) callers.emplace_back(caller, t + 1);
    for (auto &t : callers) t.join();
    // a loop of one part and an empty loop run on the calling thread
    int one = 0;
    if (!pool.run(1, [&](int) { ++one; }) || one != 1) ++bad;
    if (!pool.run(0, [&](int) { ++one; }) || one != 1) ++bad;
    printf("workers=%d loops=%ld bad=%ld\n", pool.workers(), loops.load(), bad.load());
    return bad.load() ? 1 : 0;
}
