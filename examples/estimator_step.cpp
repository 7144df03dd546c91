// estimator_step.cpp — the BA step of VISFS's estimator (corelib/src/Estimator.cpp:227-317, 391-395) on the two C ABIs of this
// repository: visfs_window.h (LocalMap's BA side, host only) and visfs_ba.h (localOptimize on the MI355X).
//
//   g++ -std=c++17 -O2 -Iinclude examples/estimator_step.cpp -Lvisfs_amd/lib -lvisfs_window -lvisfs_ba_hip
//       -Wl,-rpath,$PWD/visfs_amd/lib -o estimator_step && ./estimator_step 40
//
// A synthetic stereo front end (a camera flying through a random point cloud) stands in for the tracker.  Per frame:
// insert the signature → build the flat window (zero copy) → solve → apply the result → drop a signature.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "visfs_ba.h"
#include "visfs_window.h"

namespace {
struct Rng {                      // SplitMix64 → uniform / normal
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() { const double u = uni() + 1e-300, v = uni(); return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }
};
struct Track { uint64_t id; double w[3]; int remaining; };
}  // namespace

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 30;
    const double fx = 420, fy = 420, cx = 320, cy = 240;
    const float baseline = 0.12f;
    const double Trc[12] = { 0, 0, 1, 0.1, -1, 0, 0, 0.02, 0, -1, 0, 0.3 };     // image → robot
    visfs_window_map* window = nullptr;
    if (visfs_window_create(0, nullptr, nullptr, &window) != VISFS_BA_OK) return 2;
    visfs_ba_params prm;
    visfs_ba_default_params(&prm);                                               // reference defaults (Parameters.h:184-191)
    visfs_ba_handle* ba = nullptr;
    if (visfs_ba_create(&prm, 0, &ba) != VISFS_BA_OK) { std::fprintf(stderr, "no MI355X / gfx950 device\n"); return 3; }

    Rng rng{ 12345 };
    std::vector<Track> tracks;
    std::vector<uint64_t> prevIds; std::vector<float> prevUv;
    uint64_t nextFeature = 1;
    double x = 0, yaw = 0;                                                       // planar robot: forward motion with a little yaw
    double wheel[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    int solved = 0, culled = 0, blocked = 0;
    std::vector<double> stepMs, solveMs;              // build + solve + apply, and the solve alone, per BA step
    double lastChi = 0;
    for (int f = 1; f <= frames; ++f) {
        const double dyaw = 0.02 * rng.normal(), step = 0.12 + 0.01 * rng.normal();
        yaw += dyaw; x += step;
        const double c = std::cos(yaw), s = std::sin(yaw);
        const double Twr[12] = { c, -s, 0, x, s, c, 0, 0.2 * std::sin(0.3 * x), 0, 0, 1, 0 };
        {   // wheel odometry: the same increment with a little noise
            const double cd = std::cos(dyaw), sd = std::sin(dyaw), tx = step + 0.005 * rng.normal();
            const double d[12] = { cd, -sd, 0, tx, sd, cd, 0, 0.005 * rng.normal(), 0, 0, 1, 0 };
            double o[12];
            for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) o[4 * r + k] = wheel[4 * r] * d[k] + wheel[4 * r + 1] * d[4 + k] + wheel[4 * r + 2] * d[8 + k];
                                          o[4 * r + 3] = wheel[4 * r] * d[3] + wheel[4 * r + 1] * d[7] + wheel[4 * r + 2] * d[11] + wheel[4 * r + 3]; }
            for (int i = 0; i < 12; ++i) wheel[i] = o[i];
        }
        while (tracks.size() < 150) {                                            // spawn points 2..8 m in front of the camera
            const double z = 2 + 6 * rng.uni(), u = 20 + 600 * rng.uni(), v = 20 + 440 * rng.uni();
            const double pc[3] = { (u - cx) / fx * z, (v - cy) / fy * z, z };
            double pr[3], pw[3];
            for (int r = 0; r < 3; ++r) pr[r] = Trc[4 * r] * pc[0] + Trc[4 * r + 1] * pc[1] + Trc[4 * r + 2] * pc[2] + Trc[4 * r + 3];
            for (int r = 0; r < 3; ++r) pw[r] = Twr[4 * r] * pr[0] + Twr[4 * r + 1] * pr[1] + Twr[4 * r + 2] * pr[2] + Twr[4 * r + 3];
            tracks.push_back({ nextFeature++, { pw[0], pw[1], pw[2] }, 2 + (int)(rng.uni() * 10) });
        }
        std::vector<uint64_t> ids; std::vector<float> uv, xyz; std::vector<uint8_t> has3d;
        for (auto& t : tracks) {
            double pr[3], pc[3];                                                 // world → robot → image
            const double d[3] = { t.w[0] - Twr[3], t.w[1] - Twr[7], t.w[2] - Twr[11] };
            for (int r = 0; r < 3; ++r) pr[r] = Twr[r] * d[0] + Twr[4 + r] * d[1] + Twr[8 + r] * d[2];
            const double e[3] = { pr[0] - Trc[3], pr[1] - Trc[7], pr[2] - Trc[11] };
            for (int r = 0; r < 3; ++r) pc[r] = Trc[r] * e[0] + Trc[4 + r] * e[1] + Trc[8 + r] * e[2];
            --t.remaining;
            if (pc[2] < 0.5) { t.remaining = 0; continue; }
            const double u = fx * pc[0] / pc[2] + cx + 0.3 * rng.normal(), v = fy * pc[1] / pc[2] + cy + 0.3 * rng.normal();
            ids.push_back(t.id);
            uv.push_back((float)u); uv.push_back((float)v); uv.push_back((float)(u - fx * baseline / pc[2] + 0.3 * rng.normal())); uv.push_back((float)v);
            for (int r = 0; r < 3; ++r) xyz.push_back((float)(pr[r] + 0.03 * rng.normal()));
            has3d.push_back(1);
        }
        for (size_t i = 0; i < tracks.size();) { if (tracks[i].remaining <= 0) { tracks[i] = tracks.back(); tracks.pop_back(); } else ++i; }
        // ids must ascend (std::map order): tracks are spawned with increasing ids but removal swaps — sort by id
        std::vector<size_t> order(ids.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ids[a] < ids[b]; });
        std::vector<uint64_t> sid; std::vector<float> suv, sxyz;
        for (size_t k : order) { sid.push_back(ids[k]); for (int q = 0; q < 4; ++q) suv.push_back(uv[4 * k + q]); for (int q = 0; q < 3; ++q) sxyz.push_back(xyz[3 * k + q]); }
        const double translation[3] = { step * c, step * s, 0 };
        if (visfs_window_insert(window, (uint64_t)f, Twr, wheel, translation, (int32_t)sid.size(), sid.data(), suv.data(), sxyz.data(), has3d.data(),
                                (int32_t)prevIds.size(), prevIds.data(), prevUv.data()) != 1) { std::fprintf(stderr, "insert refused\n"); return 4; }
        prevIds = sid; prevUv.clear();
        for (size_t k = 0; k < sid.size(); ++k) { prevUv.push_back(suv[4 * k]); prevUv.push_back(suv[4 * k + 1]); }

        if (visfs_window_available(window)) {
            const auto t0 = std::chrono::steady_clock::now();
            visfs_ba_window w;                                                    // pointers into the container's buffers: zero copy
            visfs_window_build(window, Trc, fx, fy, cx, cy, baseline, 2, 1, &w);
            std::vector<uint64_t> outIds(w.n_poses + 1), outF(w.n_refs + 1), outP(w.n_refs + 1), errorVertex(w.n_refs + 1);
            std::vector<double> outT((size_t)(w.n_poses + 1) * 12);
            visfs_ba_result r{};
            r.pose_ids_out = outIds.data(); r.pose_Twr_out = outT.data();
            r.outlier_capacity = w.n_refs + 1; r.outlier_feature = outF.data(); r.outlier_pose = outP.data();
            const auto t1 = std::chrono::steady_clock::now();
            const int status = visfs_ba_solve_window(ba, &w, &r);
            const auto t2 = std::chrono::steady_clock::now();
            if (status != VISFS_BA_OK) { std::fprintf(stderr, "frame %d: status %d (%s)\n", f, status, visfs_ba_last_error(ba)); return 5; }
            if (r.n_poses_out == 6) {                                             // Estimator.cpp:275: only a full window is written back
                int32_t nErr = 0;
                visfs_window_apply(window, &r, errorVertex.data(), (int32_t)errorVertex.size(), &nErr);
                ++solved; culled += r.n_outliers; blocked += nErr; lastChi = r.chi2_final;
            }
            const auto t3 = std::chrono::steady_clock::now();
            if (f >= 10) {                                                        // (the first calls pay allocation and module load)
                stepMs.push_back(std::chrono::duration<double, std::milli>(t3 - t0).count());
                solveMs.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
            }
        }
        visfs_window_remove(window);
    }
    int32_t ns = 0, nf = 0, no = 0;
    visfs_window_counts(window, &ns, &nf, &no);
    auto median = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::printf("{\"frames\": %d, \"solved\": %d, \"outliers\": %d, \"blocked\": %d, \"signatures\": %d, \"features\": %d, \"observations\": %d, \"last_chi2\": %.6g, \"ba_step_ms_median\": %.4f, \"solve_window_ms_median\": %.4f}\n",
                frames, solved, culled, blocked, ns, nf, no, lastChi, median(stepMs), median(solveMs));
    visfs_ba_destroy(ba);
    visfs_window_destroy(window);
    return 0;
}
