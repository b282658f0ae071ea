// csrc/host_pool.h under ThreadSanitizer: several caller threads (as several contexts would) run loops of different
// sizes on the one pool at once; every part of every loop must run exactly once, a part that throws must make run()
// return false without taking the process down, and no data race may be reported.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

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
    // a fork()ed child has the pool object but none of its threads: its loops must still complete, on the
    // calling thread (the child would otherwise wait for workers that do not exist; alarm() turns that into a failure)
    fflush(stdout);
    const pid_t child = fork();
    if (child == 0) {
        alarm(20);
        std::atomic<int> parts{0};
        const bool ok = mic::HostPool::get().run(64, [&](int) { parts.fetch_add(1); });
        _exit(ok && parts.load() == 64 ? 0 : 3);
    }
    int status = -1;
    if (child < 0 || waitpid(child, &status, 0) != child || !WIFEXITED(status) || WEXITSTATUS(status) != 0) ++bad;
    printf("workers=%d loops=%ld bad=%ld fork_child=%s\n", pool.workers(), loops.load(), bad.load(),
           WIFEXITED(status) && WEXITSTATUS(status) == 0 ? "ok" : "FAILED");
    return bad.load() ? 1 : 0;
}
