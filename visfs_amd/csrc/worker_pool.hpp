// worker_pool.hpp — a few persistent host threads for the O(N_obs) host passes of one localOptimize call.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

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
    // CPUs that share the last-level cache with the CPU the calling thread is on right now (0: unknown).  The pool's default size follows
    // it: more threads than that domain has cores would have to sit on another L3 (see place_near_caller).
    static int llc_domain_cpus() {
        const int cpu = sched_getcpu();
        if (cpu < 0) return 0;
        cpu_set_t set;
        if (!read_llc_domain(cpu, set)) return 0;
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) CPU_AND(&set, &set, &allowed);
        return CPU_COUNT(&set);
    }
    // Wake the workers ahead of the first region of a call: they leave the condition variable and spin for a while.
    void prewake() {
        if (th_.empty()) return;
        { std::lock_guard<std::mutex> lk(m_); gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
    }
    // fn(task, slot) for every task in [0, n), slot = 0 for the caller and 1 .. size() - 1 for the workers (per-thread accumulators);
    // returns when all have finished.  Tasks are claimed one by one, so a worker that wakes late still takes what is left: hand out
    // several tasks per thread.  A task that throws (the bodies allocate: std::bad_alloc) never escapes a worker thread and never
    // leaves the region early: the exception is swallowed where it is thrown, every task is still counted, and run() rethrows
    // std::bad_alloc on the CALLER's thread once all tasks have drained (the C ABI's guarded() turns it into an error code).
    void run(int n, const std::function<void(int, int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i, 0); return; }
        place_near_caller();
        { std::lock_guard<std::mutex> lk(m_);
          region_ += 1; fn_ = &fn; n_ = n;
          failed_.store(false, std::memory_order_relaxed);
          left_.store(n, std::memory_order_relaxed);
          ticket_.store((unsigned long long)region_ << 32, std::memory_order_release);
          gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        work(0);
        while (left_.load(std::memory_order_acquire) > 0) { /* a worker is finishing its last task */ }
        { std::lock_guard<std::mutex> lk(m_); fn_ = nullptr; }
        if (failed_.load(std::memory_order_acquire)) throw std::bad_alloc();
    }
private:
    // The caller has just produced (or is about to consume) the arrays a region works on: they sit in ITS last-level cache.  A worker on
    // another L3 domain (a Zen CCD: 8 cores) fetches every line across the fabric and, for the arrays it writes, takes the lines away from
    // the cache the next consumer reads them from — measured on the GPU box (2 x 64 cores, 16 L3 domains) the region then runs no faster
    // than the serial loop (profiles/r03_thread_scaling*.log).  So the workers are kept on the L3 domain the caller is running on
    // (sysfs cache/index3/shared_cpu_list); the caller's own affinity is never touched, and when the scheduler has moved the caller to
    // another domain the workers follow at the next region.  VISFS_BA_POOL_AFFINITY=0 leaves placement to the scheduler.
    void place_near_caller() {
        static const bool on = []() { const char* e = std::getenv("VISFS_BA_POOL_AFFINITY"); return !(e && e[0] == '0'); }();
        if (!on) return;
        const int cpu = sched_getcpu();
        if (cpu < 0 || (cpu < CPU_SETSIZE && placed_ && CPU_ISSET(cpu, &domain_))) return;
        cpu_set_t set;
        if (!read_llc_domain(cpu, set)) return;
        // only CPUs this process may use at all
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) { CPU_AND(&set, &set, &allowed); if (CPU_COUNT(&set) < 2) return; }
        for (auto& t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
        domain_ = set; placed_ = true;
    }
    static bool read_llc_domain(const int cpu, cpu_set_t& set) {
        char path[128];
        std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        std::FILE* f = std::fopen(path, "r");
        if (!f) return false;
        char buf[512];
        const bool got = std::fgets(buf, sizeof(buf), f) != nullptr;
        std::fclose(f);
        if (!got) return false;
        CPU_ZERO(&set);
        int n_set = 0;
        for (char* q = buf; *q && *q != '\n';) {                // "0-7,128-135"
            char* e = nullptr;
            const long a = std::strtol(q, &e, 10);
            if (e == q) break;
            long b = a;
            q = e;
            if (*q == '-') { b = std::strtol(q + 1, &e, 10); q = e; }
            for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); ++n_set; }
            if (*q == ',') ++q;
        }
        return n_set >= 2;
    }
    cpu_set_t domain_;
    bool placed_ = false;
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
            try { (*f)(i, slot); } catch (...) { failed_.store(true, std::memory_order_release); }
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
    std::atomic<bool> failed_{ false };
    const std::function<void(int, int)>* fn_ = nullptr;
    int n_ = 0;
    unsigned region_ = 0;
    bool stop_ = false;
};


}  // namespace visfs_ba
