// Diagnostic (not product): does a pack-shaped streaming loop (28 bytes read, 24 bytes written per reference, SoA) scale with host
// threads on this box?  Outputs into ordinary memory and into hipHostMalloc'ed memory; 1 / 2 / 4 / 8 threads of worker_pool.hpp.
// build: hipcc -O2 -std=c++17 tools/mem_probe.cpp -o /tmp/mem_probe -lpthread
#include "../visfs_amd/csrc/worker_pool.hpp"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace visfs_ba;
struct Out { float* uvd; int32_t* p; int32_t* c; int32_t* ref; };
int main(int argc, char** argv) {
    const int Nr = argc > 1 ? std::atoi(argv[1]) : 400000;
    std::vector<uint64_t> feat(Nr), pose(Nr);
    std::vector<float> u(Nr), v(Nr), d(Nr);
    for (int k = 0; k < Nr; ++k) { feat[k] = 1000 + k / 12; pose[k] = 1 + k % 12; u[k] = k * 0.5f; v[k] = k * 0.25f; d[k] = 1.0f + (k & 7); }
    auto alloc = [&](bool pinned, Out& o) {
        auto get = [&](size_t bytes) { void* q = nullptr; if (pinned) { if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) std::abort(); } else q = std::malloc(bytes); std::memset(q, 0, bytes); return q; };
        o.uvd = (float*)get((size_t)Nr * 12); o.p = (int32_t*)get((size_t)Nr * 4); o.c = (int32_t*)get((size_t)Nr * 4); o.ref = (int32_t*)get((size_t)Nr * 4);
    };
    for (int pinned = 0; pinned < 2; ++pinned) {
        Out o; alloc(pinned != 0, o);
        for (int threads : { 1, 2, 4, 8 }) {
            WorkerPool pool(threads - 1);
            const int T = threads == 1 ? 1 : 4 * threads;
            std::function<void(int, int)> body = [&](int t, int) {
                const int k0 = (int)((int64_t)Nr * t / T), k1 = (int)((int64_t)Nr * (t + 1) / T);
                for (int k = k0; k < k1; ++k) {
                    if (!(d[k] > 0.0f)) continue;
                    o.uvd[3 * (size_t)k] = u[k]; o.uvd[3 * (size_t)k + 1] = v[k]; o.uvd[3 * (size_t)k + 2] = d[k];
                    o.p[k] = (int32_t)(feat[k] - 1000); o.c[k] = (int32_t)(pose[k] - 1); o.ref[k] = k;
                }
            };
            double best = 1e30;
            for (int rep = 0; rep < 6; ++rep) {
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
                const auto t0 = std::chrono::steady_clock::now();
                if (T > 1) pool.run(T, body); else body(0, 0);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0 && us < best) best = us;
            }
            std::printf("%s output, %d thread(s): %8.1f us for %d references (%.1f GB/s of 52 bytes each)\n", pinned ? "pinned  " : "ordinary", threads, best, Nr, 52.0 * Nr / best * 1e-3);
        }
    }
    return 0;
}
