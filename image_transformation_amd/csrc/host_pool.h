// Parked host threads for short parallel loops inside libmic.so (mic_api.hip: the axis tables of a call).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace mic {

// A few parked host threads for short parallel loops (the axis tables of a call: 64 x ~60 us).  Starting threads per
// call cost more than the work (1.15 ms for 0.5 ms of table building on 8 fresh threads, measured on the GPU box); parked
// workers are woken through one condition variable.  run(n_parts, fn) calls fn(part) for every part in 0..n_parts-1, the
// calling thread taking its share, and returns when all are done.  One loop at a time (callers hold a context lock;
// two contexts take turns).
class HostPool {
  public:
    static HostPool &get() {
        static HostPool *pool = new HostPool();  // (never destroyed: its threads are parked for the life of the process)
        return *pool;
    }
    int workers() const { return (int)threads_.size(); }
    // false: some part threw (std::bad_alloc ...); every other part has still run
    bool run(int n_parts, const std::function<void(int)> &fn) {
        std::lock_guard<std::mutex> one_at_a_time(run_mu_);
        failed_ = false;
        fn_ = &fn;
        n_parts_ = n_parts;
        next_ = 0;
        // A fork()ed child (multiprocessing 'fork', DataLoader workers) inherits this object but none of its threads:
        // waking them would wait forever.  The loop then runs on the calling thread alone.
        if (n_parts > 1 && !threads_.empty() && getpid() == owner_) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                busy_ = (int)threads_.size();
                ++generation_;
            }
            cv_.notify_all();
            drain();
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [&] { return busy_ == 0; });
        } else {
            drain();
        }
        fn_ = nullptr;
        return !failed_;
    }

  private:
    HostPool() : owner_(getpid()) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int n = (int)std::min<unsigned>(15, hw - 1);
        try {
            for (int i = 0; i < n; ++i) threads_.emplace_back([this] { loop(); });
        } catch (...) {  // fewer workers than wanted: the loops still complete
        }
        for (auto &t : threads_) t.detach();
    }
    void drain() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_parts_) break;
            try {
                (*fn_)(i);
            } catch (...) {  // (never let an exception leave a worker thread)
                failed_ = true;
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            drain();
            std::lock_guard<std::mutex> lk(mu_);
            if (--busy_ == 0) done_cv_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    const pid_t owner_;  // the process that started the workers
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    std::atomic<bool> failed_{false};
    int n_parts_ = 0, busy_ = 0;
    uint64_t generation_ = 0;
};

}  // namespace mic
