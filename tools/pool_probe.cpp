// Diagnostic (not product): what a parallel region of worker_pool.hpp costs on this host — cold (workers asleep) and warm (spinning).
#include "../visfs_amd/csrc/worker_pool.hpp"
#include <cmath>
#include <cstdio>
using namespace visfs_ba;
int main() {
    for (int workers : { 1, 3, 7 }) {
        WorkerPool pool(workers);
        std::vector<double> out(64 * 16);
        for (int rep = 0; rep < 3; ++rep) {
            std::this_thread::sleep_for(std::chrono::milliseconds(3));
            std::vector<int> slot_of(64, -1);
            std::function<void(int, int)> fn = [&](int t, int slot) { double a = 0; for (int i = 0; i < 12000; ++i) a += std::sqrt((double)i + t); out[16 * t] = a; slot_of[t] = slot; };
            auto t0 = std::chrono::steady_clock::now();
            pool.run(16, fn);
            auto t1 = std::chrono::steady_clock::now();
            int by_caller = 0; for (int t = 0; t < 16; ++t) by_caller += slot_of[t] == 0;
            pool.run(16, fn);
            auto t2 = std::chrono::steady_clock::now();
            int by_caller2 = 0; for (int t = 0; t < 16; ++t) by_caller2 += slot_of[t] == 0;
            for (int t = 0; t < 16; ++t) fn(t, 0);
            auto t3 = std::chrono::steady_clock::now();
            std::printf("workers %d: cold region %.1f us (caller ran %d of 16), warm region %.1f us (caller ran %d), serial %.1f us\n", workers,
                        std::chrono::duration<double, std::micro>(t1 - t0).count(), by_caller, std::chrono::duration<double, std::micro>(t2 - t1).count(), by_caller2,
                        std::chrono::duration<double, std::micro>(t3 - t2).count());
        }
    }
    return 0;
}
