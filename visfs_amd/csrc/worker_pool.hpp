// worker_pool.hpp — a few persistent host threads for the O(N_obs) host passes of one localOptimize call.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace visfs_ba {

// A few persistent host threads for the O(N_obs) host passes of one localOptimize call (graph build, structure summary, write-back).
// The caller takes part in every region, so a region never runs slower than the serial loop; workers spin briefly after a region
// (the next one of the same call follows within microseconds) and then sleep on a condition variable — nothing spins between calls.
class WorkerPool {
public:
    static constexpr int SPIN_US = 150;        // how long a worker spins for the next region before it sleeps
    explicit WorkerPool(int workers) {
        for (int i = 0; i < workers; ++i) th_.emplace_back([this, i]() { loop(i + 1); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    // Wake the workers ahead of the first region of a call: they leave the condition variable and spin for a while.
    void prewake() {
        if (th_.empty()) return;
        { std::lock_guard<std::mutex> lk(m_); gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
    }
    // fn(task, slot) for every task in [0, n), slot = 0 for the caller and 1 .. size() - 1 for the workers (per-thread accumulators);
    // returns when all have finished.  Tasks are claimed one by one, so a worker that wakes late still takes what is left: hand out
    // several tasks per thread.  fn must not throw.
    void run(int n, const std::function<void(int, int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i, 0); return; }
        { std::lock_guard<std::mutex> lk(m_);
          region_ += 1; fn_ = &fn; n_ = n;
          left_.store(n, std::memory_order_relaxed);
          ticket_.store((unsigned long long)region_ << 32, std::memory_order_release);
          gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        work(0);
        while (left_.load(std::memory_order_acquire) > 0) { /* a worker is finishing its last task */ }
        std::lock_guard<std::mutex> lk(m_);
        fn_ = nullptr;
    }
private:
    // Tasks are handed out through ONE word {region : 32 | next index : 32}: a worker that still holds the snapshot of an earlier region
    // can never take (or repeat) a task of a later one — its compare-and-swap fails on the region half.
    void work(const int slot) {
        const std::function<void(int, int)>* f; int n; unsigned region;
        { std::lock_guard<std::mutex> lk(m_); f = fn_; n = n_; region = region_; }
        if (!f) return;
        unsigned long long t = ticket_.load(std::memory_order_acquire);
        for (;;) {
            if ((unsigned)(t >> 32) != region) return;
            const int i = (int)(t & 0xffffffffull);
            if (i >= n) return;
            if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
            (*f)(i, slot);
            left_.fetch_sub(1, std::memory_order_acq_rel);
            t = ticket_.load(std::memory_order_acquire);
        }
    }
    void loop(const int slot) {
        unsigned seen = 0;
        for (;;) {
            // spin a little (back-to-back regions of one call), then sleep
            const auto t0 = std::chrono::steady_clock::now();
            while (gen_.load(std::memory_order_acquire) == seen) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_.wait(lk, [&]() { return gen_.load(std::memory_order_acquire) != seen; });
                    break;
                }
            }
            seen = gen_.load(std::memory_order_acquire);
            { std::lock_guard<std::mutex> lk(m_); if (stop_) return; }
            work(slot);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned> gen_{ 0 };
    std::atomic<unsigned long long> ticket_{ 0 };
    std::atomic<int> left_{ 0 };
    const std::function<void(int, int)>* fn_ = nullptr;
    int n_ = 0;
    unsigned region_ = 0;
    bool stop_ = false;
};


}  // namespace visfs_ba
