// Test driver for visfs_amd/csrc/worker_pool.hpp (compiled and run by tests/test_worker_pool.py): many back-to-back and spaced parallel
// regions of random size; every task of every region must run exactly once, on a slot in range, and a region must not return before
// its tasks have finished.  Prints "ok <regions> <tasks>" or a diagnostic and exits 1.
#include "../../visfs_amd/csrc/worker_pool.hpp"
#include <cstdint>
#include <cstdio>
#include <random>
using namespace visfs_ba;
int main(int argc, char** argv) {
    const int workers = argc > 1 ? std::atoi(argv[1]) : 3;
    WorkerPool pool(workers);
    std::mt19937 rng(12345);
    long total = 0;
    for (int region = 0; region < 3000; ++region) {
        const int n = 1 + (int)(rng() % 97);
        std::vector<std::atomic<int>> hits(n);
        for (auto& h : hits) h.store(0);
        std::atomic<int> bad_slot{ 0 }, running{ 0 };
        std::function<void(int, int)> fn = [&](int t, int slot) {
            running.fetch_add(1);
            if (slot < 0 || slot >= pool.size()) bad_slot.store(1);
            volatile double a = 0; for (int i = 0; i < 50 + (t * 37) % 400; ++i) a = a + i * 0.5;
            hits[t].fetch_add(1);
            running.fetch_sub(1);
        };
        if (region % 7 == 0) pool.prewake();
        pool.run(n, fn);
        if (running.load() != 0) { std::printf("region %d returned with %d tasks still running\n", region, running.load()); return 1; }
        if (bad_slot.load()) { std::printf("region %d: slot out of range\n", region); return 1; }
        for (int t = 0; t < n; ++t) if (hits[t].load() != 1) { std::printf("region %d: task %d ran %d times\n", region, t, hits[t].load()); return 1; }
        total += n;
        if (region % 500 == 499) std::this_thread::sleep_for(std::chrono::microseconds(400));     // let the workers fall asleep now and then
    }
    std::printf("ok 3000 %ld\n", total);
    return 0;
}
