// ba_api.cpp — host side of the C ABI (include/visfs_ba.h): graph build, index structures,
// HBM residency, the enqueue loop of the device-side LM state machine, stage hooks.
//
// There is no CPU solve path in this library: every entry point that computes fails with
// VISFS_BA_ERR_DEVICE when no HIP device is available.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/visfs_ba.h"
#include "ba_kernels.hpp"
#include "worker_pool.hpp"

using namespace visfs_ba;

static_assert(MAX_TRACE == VISFS_BA_MAX_TRACE, "trace size mismatch");

namespace {

struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    template <typename T>
    T* take(size_t n) {
        used = (used + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + used);
        used += n * sizeof(T);
        return p;
    }
};

// Host scratch of the window layer that survives from call to call (a fresh std::vector of this size is an mmap, a page fault per
// 4 KiB on first touch and an munmap, every frame — and page faults from several threads at once serialise in the kernel).
// What the run-based Schur kernel (k_schur_runs) needs to know about 8 consecutive landmarks: how many observations they have and
// which poses (index space of the window, fixed ones included) those touch.
struct RunGroup { int32_t cnt = 0; int32_t lo = 0x7fffffff, hi = -1; };
constexpr int RUN_GROUP = 8;
struct alignas(128) SummaryPart {          // (one per thread, updated per observation: no two may share a cache line)
    std::vector<int32_t> cnt, run, pc, track;
    std::vector<RunGroup> grp;
    int64_t pairs = 0; int ok = 0; bool any_run = false, used = false; const char* bad = nullptr;
    int npf = 0; int cur = -1; bool cur_fixed = false;
    int first = -1, last = -1, c = 0; bool gap = false;         // the current landmark's free poses: a contiguous run unless `gap`
    void begin(int Npf, int Nl) {
        used = true; npf = Npf; cnt.assign(Npf + 1, 0); run.assign((size_t)Npf * Npf, 0); pc.clear(); track.clear();
        grp.assign((size_t)(Nl + RUN_GROUP - 1) / RUN_GROUP + 1, RunGroup());
        pairs = 0; ok = 0; any_run = false; bad = nullptr; cur = -1; first = last = -1; c = 0; gap = false;
    }
    // observations arrive landmark by landmark (a landmark never straddles two accumulators), poses ascending inside a landmark
    inline void add(int landmark, bool landmark_fixed, int a /* free index of the pose, -1: fixed */, int pose /* its index in the window */) {
        if (landmark != cur) { flush(); cur = landmark; cur_fixed = landmark_fixed; }
        { RunGroup& G = grp[(size_t)landmark / RUN_GROUP]; G.cnt++; G.lo = std::min(G.lo, pose); G.hi = std::max(G.hi, pose); }
        ok += !(a < 0 && landmark_fixed);
        if (a < 0) return;
        cnt[a + 1]++;
        if (c == 0) { first = last = a; c = 1; return; }
        if (!gap && a != last + 1) { gap = true; track.clear(); for (int q = first; q <= last; ++q) track.push_back(q); }   // (rare) the members so far were a run
        if (gap) track.push_back(a);
        last = a; ++c;
    }
    void flush() {
        if (cur >= 0 && !cur_fixed && c > 0) {
            pairs += (int64_t)c * (c + 1) / 2;
            if (!gap) { run[(size_t)first * npf + last]++; any_run = true; }
            else {
                if (pc.empty()) pc.assign((size_t)npf * npf, 0);      // tracks with gaps are rare: their pairwise table only exists when one shows up
                for (int i = 0; i < c; ++i) { int32_t* row = pc.data() + (size_t)track[i] * npf; for (int j = i; j < c; ++j) row[track[j]]++; }
            }
        }
        c = 0; gap = false; first = last = -1;
    }
};
struct PackedWindow {
    std::vector<uint8_t> pose_fixed, point_used;
    std::vector<int32_t> obs_ref;
    visfs_ba_graph g{};               // its arrays live in the workspace's primary staging arena (the device layout)
    int32_t mono = 0;
};

// One resident window: device arena + host mirrors of what the host needs later.
struct Workspace {
    hipStream_t stream = nullptr;
    char* d_base = nullptr;  size_t d_cap = 0;
    char* h_base = nullptr;  size_t h_cap = 0;     // pinned staging: the index section (same layout as on the device), later the outputs of a solve
    char* d_prim = nullptr;  size_t d_prim_cap = 0;   // the primary section (poses, landmarks, observations, odometry / laser measurements) ...
    char* h_prim = nullptr;  size_t h_prim_cap = 0;   // ... and its pinned staging arena: the window layer packs straight into it
    LmState* h_state = nullptr;                    // pinned
    LmState* d_state = nullptr;                    // the LM state at a FIXED device address (DeviceGraph::st): survives uploads
    DeviceGraph* d_graph = nullptr;                // the resident window's DeviceGraph at a fixed device address (re-written by every upload) ...
    DeviceGraph* h_graph = nullptr;                // ... and its pinned source
    // per-frame replay (round 4): launch sequences of visfs_ba_solve_window captured per geometry class; they read the graph from d_graph,
    // so a sequence captured for one frame's window serves the next frames' windows of the same class
    struct FrameGraph { LaunchDims d; int n0, n1, half, half2, solver, flags; hipGraphExec_t exec; int seen; uint64_t last_use; };
    std::vector<FrameGraph> frame_graphs;
    uint64_t frame_clock = 0;
    DeviceGraph g{};
    bool loaded = false;
    bool batch_member = false;                     // one of visfs_ba_solve_batch / visfs_ba_batch_upload's windows (shares its launches)
    bool fused = false;                            // the resident window runs on k_small_optimize
    bool want_outputs = false;                     // solve_window: the state read of a solve also fetches the outputs (one synchronisation less)
    bool outputs_staged = false;                   // ... they are in the staging arena (both estimate buffers; LmState::sel picks)
    int batch_hint = 1;                            // windows of the batch this workspace was uploaded for (kernel choices that depend on it)
    bool spec = false;                             // units end with the speculative linearisation + LM decision launch
    bool spec_fused = false;                       // ... fused into the back-substitution launch, pose-major role deferred to the next Schur gather (k_backsub<LINA>)
    bool fused_decide = true;                      // gated units: k_backsub carries the LM decision (VISFS_BA_DECIDE_FUSED=0: k_decide, A/B runs and tests)
    int extra_units[2] = { 0, 0 };                 // rejected trials per phase of the previous solve: units enqueued on top of `half`
    bool small_solve = false;                      // reduced system <= 64 x 64: k_small_solve replaces k_schur_finalize + solver
    bool pristine = false;                         // the estimates are the uploaded ones (upload / reset, no optimise since): a solve can be re-run from its start
    bool direct_ready = false;                     // the structures of the direct solver exist although Optimizer/Solver=2: the fallback of a timed-out persistent PCG
    int solver_now = -1;                           // >= 0: the linear solver of the run in progress when it is not Optimizer/Solver (the fallback run: 0)
    int fallbacks = 0;                             // solves of this workspace that were re-run on the direct solver
    bool fell_back_last = false;                   // ... the last one was (batch members: read when their statistics are filled)
    // VISFS_BA_GRAPH=1 (measurement): the up-front launch sequence of a solve captured once per resident graph and replayed
    hipGraphExec_t graph_exec = nullptr;
    int graph_units[2] = { -1, -1 };
    int solves_since_upload = 0;
    bool graph_failed = false;                     // a capture or instantiation failed once: eager launches from then on
    bool last_replayed = false;                    // the last solve ran as a hipGraph replay (measurement: visfs_ba_graph_info)
    bool upload_in_flight = false;                 // ws_upload no longer drains its stream: whoever uses the graph from ANOTHER stream must (batch_optimize)
    // host mirrors for fetch / unpack
    std::vector<int32_t> free_pose, blk_i, blk_j, odo_i, odo_j, pose_free;
    PackedWindow pk;                               // window layer: what the graph build leaves on the host (persistent buffers)
    std::vector<SummaryPart> sum_part;             // per-thread accumulators of summarize_graph (persistent buffers)
    std::vector<int32_t> sum_run;
    int64_t n_pairs = 0;
    size_t device_bytes = 0;
    // measurement: hipEvent pairs around the launches of the enabled kernel classes
    uint32_t prof_mask = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct ProfRec { int k; hipEvent_t a, b; };
    std::vector<ProfRec> recs;
    std::vector<float> durs[VISFS_BA_K_COUNT];
    int64_t active[VISFS_BA_K_COUNT] = { 0 };
    int n6() const { return 6 * g.Npf; }
};

// RAII: one duration sample per launch of a kernel class.  `attached` (classes that are ONE kernel launch): the event pair is
// handed to the launch itself and carries the dispatch's own start / end timestamps — the figure rocprofv3 reports.  Otherwise
// (several kernels per class: direct solver, linearise with odometry, phase end) the pair is recorded around the launches on
// the workspace stream and includes the event mechanism's own share (visfs_ba_profile::null_pair_ms).
struct ProfScope {
    Workspace& w; int k; bool on; bool attached; hipEvent_t a{}, b{};
    static hipEvent_t take(Workspace& w) {
        if (w.ev_used == w.ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; w.ev_pool.push_back(e); }
        return w.ev_pool[w.ev_used++];
    }
    ProfScope(Workspace& w_, int k_, bool attached_ = false) : w(w_), k(k_), on((w_.prof_mask >> k_) & 1u), attached(attached_) {
        if (!on) return;
        a = take(w); b = take(w);
        if (!a || !b) { on = false; return; }
        if (attached) arm_launch_events(a, b);
        else (void)hipEventRecord(a, w.stream);
    }
    ~ProfScope() {
        if (!on) return;
        if (attached) {
            if (launch_events_pending()) { arm_launch_events(nullptr, nullptr); return; }   // nothing was launched: no sample
        } else (void)hipEventRecord(b, w.stream);
        w.recs.push_back({ k, a, b });
    }
};

// after a stream synchronisation: turn the recorded pairs into durations and recycle the events
static void prof_harvest(Workspace& w) {
    for (auto& r : w.recs) { float ms = 0.f; if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) w.durs[r.k].push_back(ms); }
    w.recs.clear();
    w.ev_used = 0;
}

}  // namespace

// device / pinned scratch of a batched run: the DeviceGraph array the kernels index with blockIdx.y, and the gathered LM states
struct BatchScratch {
    DeviceGraph* d_graphs = nullptr; size_t cap_graphs = 0;
    size_t cap_states = 0;
    LmState* d_lm = nullptr; LmState* h_lm = nullptr;      // [cap_states] whole LM states, gathered at the end of a run
    DeviceGraph* d_all = nullptr; size_t cap_all = 0;      // every graph of a resident batch (visfs_ba_batch_upload), for the one-launch reset
    std::vector<DeviceGraph> host_graphs;                  // source of the asynchronous H2D copy of d_graphs: must outlive it
    LaunchDims all_dims{};
};

struct visfs_ba_handle {
    visfs_ba_params prm;
    int device = 0;
    int tuning = VISFS_BA_TUNE_LATENCY;            // visfs_ba_set_tuning: THROUGHPUT selects the single-workgroup PCG for the uploads that follow
    std::string err;
    Workspace ws;
    std::vector<Workspace*> batch;
    int n_batch = 0;                               // graphs resident through visfs_ba_batch_upload
    BatchScratch scratch;
    std::vector<BatchScratch> part_scratch;        // further parts of a split batch (batch_optimize_group): scratch and stream of part k + 1
    std::vector<hipStream_t> part_stream;
    std::unique_ptr<WorkerPool> pool;              // host threads of the window layer (created at the first visfs_ba_solve_window)
    bool pool_tried = false;
};

namespace {

#define HIP_TRY(h, expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            return VISFS_BA_ERR_DEVICE;                                                   \
        }                                                                                 \
    } while (0)

int bad(visfs_ba_handle* h, const char* msg) { h->err = msg; return VISFS_BA_ERR_BAD_ARGUMENT; }

// Host threads for the O(N_obs) passes of a window solve: VISFS_BA_THREADS in total, 1 = none.  Default: the cores that share the
// last-level cache with the calling thread (half its CPUs: SMT siblings do not help a streaming loop), at most 8 — the workers are kept
// on that cache domain (worker_pool.hpp), so more threads than it has cores would only queue up; min(4, hardware threads) when the
// topology cannot be read.  The regions are bursts of 50-400 us per call; nothing spins between calls.
WorkerPool* host_pool(visfs_ba_handle* h) {
    if (!h->pool_tried) {
        h->pool_tried = true;
        const char* e = std::getenv("VISFS_BA_THREADS");
        const int llc = WorkerPool::llc_domain_cpus();
        const int dflt = llc >= 2 ? std::min(8, std::max(1, llc / 2)) : (int)std::min(4u, std::max(1u, std::thread::hardware_concurrency()));
        int n = e ? std::atoi(e) : dflt;
        n = std::max(1, std::min(n, 16));
        if (n > 1) { try { h->pool.reset(new WorkerPool(n - 1)); } catch (...) { h->pool.reset(); } }
    }
    return h->pool.get();
}

// No exception may cross the C ABI: host allocations (std::vector, std::string, std::thread) can throw.
template <typename F>
int guarded(visfs_ba_handle* h, F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { try { if (h) h->err = "out of host memory"; } catch (...) {} return VISFS_BA_ERR_DEVICE; }
    catch (const std::exception& e) { try { if (h) h->err = e.what(); } catch (...) {} return VISFS_BA_ERR_DEVICE; }
    catch (...) { return VISFS_BA_ERR_DEVICE; }
}

void ws_release(Workspace& w) {
    if (w.d_base) (void)hipFree(w.d_base);
    if (w.h_base) (void)hipHostFree(w.h_base);
    if (w.d_prim) (void)hipFree(w.d_prim);
    if (w.h_prim) (void)hipHostFree(w.h_prim);
    if (w.h_state) (void)hipHostFree(w.h_state);
    for (hipEvent_t e : w.ev_pool) (void)hipEventDestroy(e);
    if (w.graph_exec) (void)hipGraphExecDestroy(w.graph_exec);
    for (auto& fg : w.frame_graphs) if (fg.exec) (void)hipGraphExecDestroy(fg.exec);
    if (w.d_state) (void)hipFree(w.d_state);
    if (w.d_graph) (void)hipFree(w.d_graph);
    if (w.h_graph) (void)hipHostFree(w.h_graph);
    if (w.stream) (void)hipStreamDestroy(w.stream);
    w = Workspace{};
}

int ws_init(visfs_ba_handle* h, Workspace& w) {
    if (w.stream) return VISFS_BA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&w.h_state), sizeof(LmState), hipHostMallocDefault));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&w.d_state), sizeof(LmState)));
    HIP_TRY(h, hipMemset(w.d_state, 0, sizeof(LmState)));                       // (decide_epoch counts on from zero, across uploads)
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&w.d_graph), sizeof(DeviceGraph)));
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&w.h_graph), sizeof(DeviceGraph), hipHostMallocDefault));
    return VISFS_BA_OK;
}

// bytes of the pinned arena a solve's outputs need when they travel with its state read (layout: output_stage_of, below)
size_t output_stage_bytes(const DeviceGraph& g) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    return 2 * up((size_t)g.Np * POSE_STRIDE * 8) + 2 * up((size_t)g.Nl * 24) + (size_t)g.No;
}

// ------------------------------------------------------------------ graph → device structures
// The PRIMARY section: what localOptimize's inputs carry (poses, landmarks, observations, odometry / laser measurements), in its own
// device allocation and pinned staging arena.  Its layout depends only on the sizes below, so (a) the window layer packs straight
// into the staging arena (no second copy of anything of size N_obs) and (b) its host-to-device copy starts before the structure of
// S has been worked out.
struct PrimSizes { int Np = 0, Nl = 0, cap_obs = 0, cap_odo = 0, Nz = 0; size_t grid_cells = 0; bool raw_refs = false; };   // raw_refs: (u, v, depth) floats instead of (u_l, v_l, u_r) doubles
void layout_prim(Arena& A, DeviceGraph& g, const PrimSizes& z) {
    g.pose0 = A.take<double>((size_t)z.Np * POSE_STRIDE);
    g.pt0 = A.take<double>((size_t)std::max(z.Nl, 1) * 3);
    g.pose_free = A.take<int32_t>(z.Np);
    g.free_pose = A.take<int32_t>(z.Np);
    g.pt_fixed = A.take<uint8_t>(std::max(z.Nl, 1));
    g.obs_pose = A.take<int32_t>(std::max(z.cap_obs, 1));
    g.obs_pt = A.take<int32_t>(std::max(z.cap_obs, 1));
    if (z.raw_refs) { g.obs_uvd = A.take<float>((size_t)std::max(z.cap_obs, 1) * 3); g.obs_uvr = nullptr; }
    else { g.obs_uvr = A.take<double>((size_t)std::max(z.cap_obs, 1) * 3); g.obs_uvd = nullptr; }
    g.odo_i = A.take<int32_t>(std::max(z.cap_odo, 1));
    g.odo_j = A.take<int32_t>(std::max(z.cap_odo, 1));
    g.odo_tq = A.take<double>((size_t)std::max(z.cap_odo, 1) * 7);
    g.laser_xyz = A.take<double>((size_t)std::max(z.Nz, 1) * 3);
    g.grid.cost = A.take<float>(std::max<size_t>(z.grid_cells, 1));
}
size_t prim_bytes(const PrimSizes& z) { Arena a{ nullptr, 0, 0 }; DeviceGraph g{}; layout_prim(a, g, z); return (a.used + 255) & ~size_t(255); }
int ensure_prim(visfs_ba_handle* h, Workspace& w, size_t bytes) {
    HIP_TRY(h, hipSetDevice(h->device));
    if (w.h_prim_cap < bytes) {
        // (the previous upload's host-to-device copy reads the arena that is about to be replaced)
        if (w.stream) HIP_TRY(h, hipStreamSynchronize(w.stream));
        if (w.h_prim) (void)hipHostFree(w.h_prim);
        w.h_prim = nullptr; w.h_prim_cap = 0;
        const size_t want = bytes + bytes / 4;                                                   // a sliding window grows and shrinks by a few observations per frame
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&w.h_prim), want, hipHostMallocDefault));
        w.h_prim_cap = want;
    }
    if (w.d_prim_cap < bytes) {
        if (w.stream) HIP_TRY(h, hipStreamSynchronize(w.stream));
        if (w.d_prim) (void)hipFree(w.d_prim);
        w.d_prim = nullptr; w.d_prim_cap = 0;
        const size_t want = bytes + bytes / 4;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&w.d_prim), want));
        w.d_prim_cap = want;
    }
    return VISFS_BA_OK;
}

// What the host has to know about the O(N_obs) part of the graph: observations per free pose and co-observation pairs per block of S.
struct GraphSummary {
    std::vector<int32_t> cnt;          // [Npf + 1] prefix: observations of free pose a at pose-major positions [cnt[a], cnt[a + 1])
    std::vector<int32_t> pcount;       // [Npf][Npf] (a <= b): free landmarks observed by both
    int64_t pairs_seen = 0;
    int n_edges_ok = 0;
    const char* bad = nullptr;
    std::vector<RunGroup> grp;         // per RUN_GROUP consecutive landmarks
};
// One pass over the observations (landmark-major), shared out over the pool at landmark boundaries.
// A landmark seen by the free poses {a_1 < ... < a_k} adds one pair to every block (a_i, a_j), i <= j.  Tracks are runs of consecutive
// key-frames almost always (a feature is tracked frame to frame and never re-acquired), so a landmark whose free poses form the
// contiguous range [s, e] only bumps run[s][e]; the blocks are then counted by one 2-D inclusion sum, pcount[a][b] = sum over s <= a,
// e >= b of run[s][e] — O(Nl + Npf^2) instead of O(sum k^2).  Landmarks with gaps in their track take the pairwise loop.
void merge_summary(std::vector<SummaryPart>& part, const int NT, const int Npf, GraphSummary& S, std::vector<int32_t>& run) {
    const size_t nn = (size_t)Npf * Npf;
    S.cnt.assign(Npf + 1, 0); S.pcount.assign(nn, 0); S.pairs_seen = 0; S.n_edges_ok = 0; S.bad = nullptr;
    run.assign(nn, 0);
    bool any_run = false;
    for (int t = 0; t < NT; ++t) {
        const SummaryPart& P = part[t];
        if (!P.used) continue;
        if (P.bad) { S.bad = P.bad; return; }
        for (int a = 0; a <= Npf; ++a) S.cnt[a] += P.cnt[a];
        if (!P.pc.empty()) for (size_t q = 0; q < nn; ++q) S.pcount[q] += P.pc[q];
        for (size_t q = 0; q < nn; ++q) run[q] += P.run[q];
        S.pairs_seen += P.pairs; S.n_edges_ok += P.ok; any_run = any_run || P.any_run;
    }
    for (int a = 0; a < Npf; ++a) S.cnt[a + 1] += S.cnt[a];
    S.grp.clear();
    for (int t = 0; t < NT; ++t) {
        const SummaryPart& P = part[t];
        if (!P.used) continue;
        if (S.grp.size() < P.grp.size()) S.grp.resize(P.grp.size());
        for (size_t q = 0; q < P.grp.size(); ++q) { RunGroup& G = S.grp[q]; G.cnt += P.grp[q].cnt; G.lo = std::min(G.lo, P.grp[q].lo); G.hi = std::max(G.hi, P.grp[q].hi); }
    }
    if (any_run) {
        // in place: run[a][b] <- sum_{s <= a, e >= b} run[s][e]
        for (int a = 0; a < Npf; ++a)
            for (int b2 = Npf - 1; b2 >= 0; --b2) {
                int64_t v = run[(size_t)a * Npf + b2];
                if (a > 0) v += run[(size_t)(a - 1) * Npf + b2];
                if (b2 + 1 < Npf) v += run[(size_t)a * Npf + b2 + 1];
                if (a > 0 && b2 + 1 < Npf) v -= run[(size_t)(a - 1) * Npf + b2 + 1];
                run[(size_t)a * Npf + b2] = (int32_t)v;
            }
        for (int a = 0; a < Npf; ++a) for (int b2 = a; b2 < Npf; ++b2) S.pcount[(size_t)a * Npf + b2] += run[(size_t)a * Npf + b2];
    }
}
void summarize_graph(const visfs_ba_graph* gr, const int32_t* pose_free, const int Npf, WorkerPool* pool, const bool check, GraphSummary& S,
                     std::vector<SummaryPart>& part, std::vector<int32_t>& run) {
    const int No = gr->n_obs, Nl = gr->n_points, Np = gr->n_poses;
    const size_t nn = (size_t)Npf * Npf;
    // several tasks per thread (a worker that wakes late still finds work); accumulators per THREAD, not per task
    const int NT = (pool && No >= 16384 && nn <= ((size_t)1 << 17)) ? pool->size() : 1;
    const int T = NT > 1 ? 4 * NT : 1;
    std::vector<int> cut(T + 1, No);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        int k = (int)((int64_t)No * t / T);
        k = std::max(k, cut[t - 1]);
        while (k > 0 && k < No && gr->obs_point[k] == gr->obs_point[k - 1]) ++k;     // a landmark's observations stay with one task
        cut[t] = k;
    }
    if ((int)part.size() < NT) part.resize(NT);
    for (int t = 0; t < NT; ++t) part[t].used = false;
    auto body = [&](int t, int slot) {
        SummaryPart& P = part[slot];
        if (!P.used) P.begin(Npf, Nl);
        if (P.bad) return;
        for (int k = cut[t]; k < cut[t + 1]; ++k) {
            const int l = gr->obs_point[k], cp = gr->obs_pose[k];
            if (check) {
                if (l < 0 || l >= Nl || cp < 0 || cp >= Np) { P.bad = "observation index out of range"; return; }
                if (k > 0 && (l < gr->obs_point[k - 1] || (l == gr->obs_point[k - 1] && cp <= gr->obs_pose[k - 1]))) { P.bad = "observations must be sorted by (point, pose) and unique"; return; }
            }
            P.add(l, gr->point_fixed[l] != 0, pose_free[cp], cp);
        }
        P.flush();
    };
    if (T > 1) pool->run(T, body); else body(0, 0);
    merge_summary(part, NT, Npf, S, run);
}

struct UploadOpts {
    bool in_staging = false;   // the graph's arrays already live in the workspace's primary staging arena, laid out for `cap` (window layer)
    PrimSizes cap;             // ... the sizes that layout was made for
    bool trusted = false;      // ... and were produced by the library's own graph build: no validation pass
    WorkerPool* pool = nullptr;
    double baseline = 0.0;     // raw references (cap.raw_refs): the stereo baseline the device forms the disparity with
    int summary_slots = 0;     // > 0: the graph build has already fed w.sum_part[0 .. summary_slots) (no second pass over the observations)
};

int ws_upload(visfs_ba_handle* h, Workspace& w, const visfs_ba_graph* gr, const UploadOpts& opt = UploadOpts()) {
    const bool timing = std::getenv("VISFS_BA_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (timing) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "  upload %-14s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - T0).count()); T0 = t; } };
    const visfs_ba_params& prm = h->prm;
    const bool ceres = prm.framework == 1;
    // From the first mutation below (the primary arrays are overwritten before the graph has been validated) until the upload has gone
    // through, NO graph is resident: a failed upload must not leave the previous graph's index structures marked usable over the
    // rejected graph's arrays (visfs_ba_optimize then returns VISFS_BA_ERR_NOT_LOADED instead of walking them).
    w.loaded = false;
    // (Optimizer.cpp:405-422: the Ceres branch never adds a wheel-odometry factor — links between two window poses fall in its "TODO" arm)
    const int Np = gr->n_poses, Nl = gr->n_points, No = gr->n_obs, Ne = ceres ? 0 : gr->n_odo;
    if (Np < 1 || Nl < 0 || No < 0 || Ne < 0) return bad(h, "negative sizes");
    for (int e = 0; e < Ne; ++e)
        if (gr->odo_from[e] < 0 || gr->odo_from[e] >= Np || gr->odo_to[e] < 0 || gr->odo_to[e] >= Np || gr->odo_from[e] == gr->odo_to[e])
            return bad(h, "odometry edge index out of range");
    // laser occupied-space edges (Optimizer.cpp:224-258): active only while their pose is free (allVerticesFixed otherwise)
    int Nz = 0;
    size_t grid_cells = 0;
    if (gr->n_laser < 0) return bad(h, "negative sizes");
    if (gr->n_laser > 0) {
        if (!gr->grid || !gr->laser_xyz) return bad(h, "laser points need a grid and coordinates");
        const visfs_ba_grid& G = *gr->grid;
        if (gr->laser_pose < 0 || gr->laser_pose >= Np) return bad(h, "laser pose index out of range");
        if (!(G.resolution > 0.0) || G.num_x_cells <= 0 || G.num_y_cells <= 0 || !G.correspondence_cost) return bad(h, "invalid grid");
        if ((int64_t)G.num_x_cells * G.num_y_cells > (int64_t)1 << 28) { h->err = "grid larger than 2^28 cells"; return VISFS_BA_ERR_UNSUPPORTED; }
        if (!gr->pose_fixed[gr->laser_pose]) { Nz = gr->n_laser; grid_cells = (size_t)G.num_x_cells * G.num_y_cells; }
    }
    int rc = ws_init(h, w);
    if (rc != VISFS_BA_OK) return rc;
    if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; w.graph_units[0] = w.graph_units[1] = -1; }   // kernel arguments change
    w.solves_since_upload = 0;
    HIP_TRY(h, hipSetDevice(h->device));

    // ---- the primary section: staged (or already there), then on its way to the device while the structure of S is worked out
    PrimSizes pz;
    if (opt.in_staging) pz = opt.cap;
    else {
        pz.Np = Np; pz.Nl = Nl; pz.cap_obs = No; pz.cap_odo = Ne; pz.Nz = Nz; pz.grid_cells = grid_cells;
        HIP_TRY(h, hipStreamSynchronize(w.stream));        // the previous upload's H2D copy reads the pinned staging arenas filled below
    }
    const size_t pbytes = prim_bytes(pz);
    rc = ensure_prim(h, w, pbytes);
    if (rc != VISFS_BA_OK) return rc;
    DeviceGraph hg{};                 // pointers into the pinned staging arenas
    { Arena hp{ w.h_prim, w.h_prim_cap, 0 }; layout_prim(hp, hg, pz); }
    if (!opt.in_staging) {
        double* p0 = const_cast<double*>(hg.pose0);
        for (int i = 0; i < Np; ++i) { for (int q = 0; q < 7; ++q) p0[POSE_STRIDE * i + q] = gr->pose_tq[7 * i + q]; p0[POSE_STRIDE * i + 7] = 0.0; }
        if (Nl) std::memcpy(const_cast<double*>(hg.pt0), gr->point_xyz, (size_t)Nl * 24);
        if (Nl) std::memcpy(const_cast<uint8_t*>(hg.pt_fixed), gr->point_fixed, Nl);
        if (No) {
            std::memcpy(const_cast<int32_t*>(hg.obs_pose), gr->obs_pose, (size_t)No * 4);
            std::memcpy(const_cast<int32_t*>(hg.obs_pt), gr->obs_point, (size_t)No * 4);
            std::memcpy(const_cast<double*>(hg.obs_uvr), gr->obs_uvr, (size_t)No * 24);
        }
        if (Ne) {
            std::memcpy(const_cast<int32_t*>(hg.odo_i), gr->odo_from, (size_t)Ne * 4);
            std::memcpy(const_cast<int32_t*>(hg.odo_j), gr->odo_to, (size_t)Ne * 4);
            std::memcpy(const_cast<double*>(hg.odo_tq), gr->odo_tq, (size_t)Ne * 56);
        }
    }
    if (Nz) {
        std::memcpy(const_cast<double*>(hg.laser_xyz), gr->laser_xyz, (size_t)Nz * 24);
        std::memcpy(const_cast<float*>(hg.grid.cost), gr->grid->correspondence_cost, grid_cells * 4);
    }
    // buildIndexMapping: free poses in index (= id) order
    std::vector<int32_t> pose_free(Np), free_pose;
    for (int i = 0; i < Np; ++i) { if (gr->pose_fixed[i]) pose_free[i] = -1; else { pose_free[i] = (int32_t)free_pose.size(); free_pose.push_back(i); } }
    const int Npf = (int)free_pose.size();
    std::memcpy(const_cast<int32_t*>(hg.pose_free), pose_free.data(), (size_t)Np * 4);
    if (Npf) std::memcpy(const_cast<int32_t*>(hg.free_pose), free_pose.data(), (size_t)Npf * 4);
    HIP_TRY(h, hipMemcpyAsync(w.d_prim, w.h_prim, pbytes, hipMemcpyHostToDevice, w.stream));
    lap("primary");

    // ---- what the host needs of the O(N_obs) part: observations per free pose, co-observation pairs per block
    GraphSummary sum;
    if (opt.summary_slots > 0) merge_summary(w.sum_part, opt.summary_slots, Npf, sum, w.sum_run);
    else summarize_graph(gr, pose_free.data(), Npf, opt.pool, !opt.trusted, sum, w.sum_part, w.sum_run);
    if (sum.bad) return bad(h, sum.bad);
    const std::vector<int32_t>& cnt = sum.cnt;
    std::vector<int32_t>& pcount = sum.pcount;
    const int64_t pairs_seen = sum.pairs_seen;
    const int n_edges_ok = sum.n_edges_ok;
    const int n_pose_obs = cnt[Npf];
    lap("summary");

    // pose-major chunks
    std::vector<int32_t> chunk_pose, chunk_ptr, pose_chunk_ptr(Npf + 1, 0);
    chunk_pose.reserve((size_t)cnt[Npf] / LIN_CHUNK + Npf + 1); chunk_ptr.reserve((size_t)cnt[Npf] / LIN_CHUNK + Npf + 2);
    for (int a = 0; a < Npf; ++a) {
        pose_chunk_ptr[a] = (int32_t)chunk_pose.size();
        for (int s = cnt[a]; s < cnt[a + 1]; s += LIN_CHUNK) { chunk_pose.push_back(a); chunk_ptr.push_back(s); }
    }
    pose_chunk_ptr[Npf] = (int32_t)chunk_pose.size();
    const int n_chunks = (int)chunk_pose.size();
    chunk_ptr.push_back(cnt[Npf]);
    // k_linearize reads [chunk_ptr[c], chunk_ptr[c+1]) — consecutive chunks are contiguous: chunk c of pose a ends where the next starts
    for (int c = 0; c + 1 < n_chunks; ++c) if (chunk_ptr[c + 1] != std::min(chunk_ptr[c] + LIN_CHUNK, cnt[chunk_pose[c] + 1])) return bad(h, "internal: chunk layout");

    // odometry incidence (per free pose: its edges in edge order, 2e = as `from`, 2e + 1 = as `to`; counting pass, then fill — no
    // per-pose containers: this section runs on the caller's thread between the graph build and the first launch)
    std::vector<int32_t> pose_odo_ptr(Npf + 1, 0), pose_odo;
    {
        for (int e = 0; e < Ne; ++e) {
            const int a = pose_free[gr->odo_from[e]], b = pose_free[gr->odo_to[e]];
            if (a >= 0) pose_odo_ptr[a + 1]++;
            if (b >= 0) pose_odo_ptr[b + 1]++;
        }
        // the laser edges' aggregate lives in slot Ne of odo_blk as the "from" side of a pseudo edge (added last, as g2o does)
        if (Nz > 0) pose_odo_ptr[pose_free[gr->laser_pose] + 1]++;
        for (int a = 0; a < Npf; ++a) pose_odo_ptr[a + 1] += pose_odo_ptr[a];
        pose_odo.resize(pose_odo_ptr[Npf]);
        std::vector<int32_t> at(pose_odo_ptr.begin(), pose_odo_ptr.end() - 1);
        for (int e = 0; e < Ne; ++e) {
            const int a = pose_free[gr->odo_from[e]], b = pose_free[gr->odo_to[e]];
            if (a >= 0) pose_odo[at[a]++] = 2 * e;
            if (b >= 0) pose_odo[at[b]++] = 2 * e + 1;
        }
        if (Nz > 0) pose_odo[at[pose_free[gr->laser_pose]]++] = 2 * Ne;
    }

    // S block structure (g2o buildStructure analogue): per block (i<=j) the co-observation pairs
    std::vector<uint8_t> has_odo((size_t)Npf * Npf, 0);
    if (pairs_seen > 0x7fffffff) { h->err = "window too large (pair list)"; return VISFS_BA_ERR_UNSUPPORTED; }
    for (int e = 0; e < Ne; ++e) {
        int a = pose_free[gr->odo_from[e]], b = pose_free[gr->odo_to[e]];
        if (a < 0 || b < 0) continue;
        if (a > b) std::swap(a, b);
        has_odo[(size_t)a * Npf + b] = 1;
    }
    std::vector<int32_t> blk_i, blk_j, blk_ptr(1, 0), blk_of((size_t)Npf * Npf, -1);
    blk_i.reserve((size_t)Npf * 16); blk_j.reserve((size_t)Npf * 16); blk_ptr.reserve((size_t)Npf * 16 + 1);
    int64_t npairs = 0;
    for (int a = 0; a < Npf; ++a)
        for (int b = a; b < Npf; ++b) {
            const size_t key = (size_t)a * Npf + b;
            if (a == b || pcount[key] > 0 || has_odo[key]) {
                blk_of[key] = (int32_t)blk_i.size();
                blk_i.push_back(a); blk_j.push_back(b);
                npairs += pcount[key];
                if (npairs > 0x7fffffff) { h->err = "window too large (pair list)"; return VISFS_BA_ERR_UNSUPPORTED; }
                blk_ptr.push_back((int32_t)npairs);
            }
        }
    const int n_blk = (int)blk_i.size();
    std::vector<int32_t> blk_odo_ptr(n_blk + 1, 0), blk_odo;
    if (Ne > 0) {
        // Aij is (row = from, col = to): stored block is (min,max); transposed when from > to
        auto slot = [&](int e, int& code) {
            const int a = pose_free[gr->odo_from[e]], b = pose_free[gr->odo_to[e]];
            if (a < 0 || b < 0) return -1;
            if (a < b) { code = 2 * e; return (int)blk_of[(size_t)a * Npf + b]; }
            code = 2 * e + 1; return (int)blk_of[(size_t)b * Npf + a];
        };
        int code = 0;
        for (int e = 0; e < Ne; ++e) { const int b = slot(e, code); if (b >= 0) blk_odo_ptr[b + 1]++; }
        for (int b = 0; b < n_blk; ++b) blk_odo_ptr[b + 1] += blk_odo_ptr[b];
        blk_odo.resize(blk_odo_ptr[n_blk]);
        std::vector<int32_t> at(blk_odo_ptr.begin(), blk_odo_ptr.end() - 1);
        for (int e = 0; e < Ne; ++e) { const int b = slot(e, code); if (b >= 0) blk_odo[at[b]++] = code; }
    }
    // block-row adjacency of the symmetric S for the mat-vec: row r = its transposed blocks (i < r, code 2b + 1) in ascending i, then its
    // own blocks (j >= r, code 2b) in ascending j — the blocks are enumerated in (i, j) order, so both halves arrive sorted by column
    std::vector<int32_t> row_ptr(Npf + 1, 0), row_col, row_blk;
    {
        for (int b = 0; b < n_blk; ++b) { row_ptr[blk_i[b] + 1]++; if (blk_i[b] != blk_j[b]) row_ptr[blk_j[b] + 1]++; }
        for (int a = 0; a < Npf; ++a) row_ptr[a + 1] += row_ptr[a];
        row_col.resize(row_ptr[Npf]); row_blk.resize(row_ptr[Npf]);
        std::vector<int32_t> at(row_ptr.begin(), row_ptr.end() - 1);
        for (int b = 0; b < n_blk; ++b) if (blk_i[b] != blk_j[b]) { const int r = blk_j[b]; row_col[at[r]] = blk_i[b]; row_blk[at[r]] = 2 * b + 1; at[r]++; }
        for (int b = 0; b < n_blk; ++b) { const int r = blk_i[b]; row_col[at[r]] = blk_j[b]; row_blk[at[r]] = 2 * b; at[r]++; }
    }

    // ---- Schur complement by RUNS OF LANDMARKS (k_schur_runs, ba_kernels.hip): the plan.  A workgroup owns run_lr x run_m consecutive
    // landmarks; run_lr is the largest of 64 / 32 / 16 / 8 whose sub-batches never hold more than RUN_MAX_TILES observations (their tiles are
    // staged in LDS); a run's blocks are those of its pose span [lo, lo + W) — from the structure summary, which has seen every observation's
    // landmark and pose.  Windows it does not fit (a span above RUN_MAX_W poses: landmarks whose ids do not grow with time; reduced systems
    // that k_small_solve serves; VISFS_BA_SCHUR_RUNS=0) keep the pair-list gather.  A property of the window alone.
    struct RunPlan { int LR = 0, M = 1, n = 0, cap = 0, wmax = 0; size_t lds = 0; int64_t total = 0; std::vector<int4> desc; std::vector<int32_t> first, last, k0; } rp;
    {
        // Measured (profiles/r04_schur_lds_tiles.log): a third of the gather's VALU instructions per pair, but every workgroup is a chain
        // of barrier-separated phases (descriptors, global loads, D_l / R_i, tiles, accumulate, three reduction passes: 12 us stamped at C2)
        // at two to three waves per SIMD — 15.7 us per launch against the gather's 8.7 at C2, 34 against 29.5 at the 200-key-frame window,
        // 16 resident windows 78.6 k against 79.0 k it/s at best.  The gather stays the default; VISFS_BA_SCHUR_RUNS=1 selects this kernel
        // (tests, A/B runs).
        const char* e = std::getenv("VISFS_BA_SCHUR_RUNS");
        const bool want = e && e[0] == '1';
        const int ng = (Nl + RUN_GROUP - 1) / RUN_GROUP;
        if (want && Npf >= 1 && No > 0 && Nl > 0 && (size_t)6 * Npf > (size_t)SM_MAX_N6 && (int)sum.grp.size() >= ng && Np < 65536) {
            // (first choice: sub-batches of at most 176 tiles — 30 KB of LDS, three workgroups and more per CU; else whatever fits at all)
            for (int limit : { 176, RUN_MAX_TILES }) {
                for (int LR : { 64, 32, 16, 8 }) {
                    const int gper = LR / RUN_GROUP;
                    int maxc = 0;
                    for (int g0 = 0; g0 < ng; g0 += gper) { int c = 0; for (int q = g0; q < std::min(ng, g0 + gper); ++q) c += sum.grp[q].cnt; maxc = std::max(maxc, c); }
                    if (maxc <= limit) { rp.LR = LR; rp.cap = maxc; break; }
                }
                if (rp.LR) break;
            }
            { const char* el = std::getenv("VISFS_BA_RUN_LR"); const int q = el ? std::atoi(el) : 0;                    // tuning override (never above what fits)
              if (rp.LR && (q == 8 || q == 16 || q == 32 || q == 64) && q < rp.LR) { rp.LR = q; rp.cap = 0; const int gper = q / RUN_GROUP;
                  for (int g0 = 0; g0 < ng; g0 += gper) { int c = 0; for (int qq = g0; qq < std::min(ng, g0 + gper); ++qq) c += sum.grp[qq].cnt; rp.cap = std::max(rp.cap, c); } } }
            if (rp.LR) {
                // sub-batches per workgroup: one, until the window has so many landmarks that a partial per run and block costs more traffic
                // than the workgroups it keeps busy (a 200-key-frame window: ~940 runs of 32 landmarks at M = 1)
                rp.M = std::max(1, 32 / rp.LR);                 // at least 32 landmarks per workgroup: a partial per run and block is traffic
                while ((Nl + rp.LR * rp.M - 1) / (rp.LR * rp.M) > 1024 && rp.M < 16) rp.M *= 2;
                { const char* em = std::getenv("VISFS_BA_RUN_M"); const int q = em ? std::atoi(em) : 0; if (q >= 1 && q <= 16) rp.M = q; }
                const int per = rp.LR * rp.M, gper = per / RUN_GROUP;
                rp.n = (Nl + per - 1) / per;
                rp.desc.assign(rp.n, make_int4(0, 0, 0, 0));
                bool ok = rp.n < (1 << 20);
                int nbmax = 0;
                for (int r = 0; r < rp.n && ok; ++r) {
                    int lo = 0x7fffffff, hi = -1;
                    for (int q = r * gper; q < std::min(ng, (r + 1) * gper); ++q) if (sum.grp[q].cnt > 0) { lo = std::min(lo, sum.grp[q].lo); hi = std::max(hi, sum.grp[q].hi); }
                    const int W = hi >= lo ? hi - lo + 1 : 0;
                    if (W > RUN_MAX_W) { ok = false; break; }
                    rp.desc[r] = make_int4(W ? lo : 0, W, (int)rp.total, 0);
                    rp.total += (int64_t)W * (W + 1) / 2;
                    rp.wmax = std::max(rp.wmax, W); nbmax = std::max(nbmax, W * (W + 1) / 2);
                    if (rp.total > 0x3fffffff) ok = false;
                }
                if (ok) {
                    // per stored block of S: the range of runs whose span holds both of its poses (k_schur_finalize walks it)
                    rp.first.assign(n_blk, 0x7fffffff); rp.last.assign(n_blk, -1);
                    for (int r = 0; r < rp.n; ++r) {
                        const int lo = rp.desc[r].x, W = rp.desc[r].y;
                        for (int li = 0; li < W; ++li) {
                            const int a = pose_free[lo + li];
                            if (a < 0) continue;
                            for (int lj = li; lj < W; ++lj) {
                                const int b2 = pose_free[lo + lj];
                                if (b2 < 0) continue;
                                const int bk = blk_of[(size_t)a * Npf + b2];
                                if (bk >= 0) { rp.first[bk] = std::min(rp.first[bk], r); rp.last[bk] = std::max(rp.last[bk], r); }
                            }
                        }
                    }
                    for (int b = 0; b < n_blk && ok; ++b) if (rp.last[b] >= 0 && rp.last[b] - rp.first[b] + 1 > 4095) ok = false;
                }
                if (!ok) { rp = RunPlan(); }
                else {
                    rp.lds = (((size_t)std::max(rp.cap, (256 * 14 + RUN_TILE - 1) / RUN_TILE) * RUN_TILE + 9 * (size_t)rp.LR + 9 * (size_t)rp.wmax) * 8 + (((size_t)rp.LR * rp.wmax + 3) & ~size_t(3)) * 2 + (size_t)nbmax * 2 + 15) & ~size_t(15);
                    // first observation of every sub-batch of run_lr landmarks (observations are landmark-major: a prefix sum of the groups' counts)
                    const int nsub = rp.n * rp.M, gl = rp.LR / RUN_GROUP;
                    rp.k0.assign((size_t)nsub + 1, 0);
                    int acc = 0;
                    for (int sb = 0; sb < nsub; ++sb) { rp.k0[sb] = acc; for (int q = sb * gl; q < std::min(ng, (sb + 1) * gl); ++q) acc += sum.grp[q].cnt; }
                    rp.k0[nsub] = acc;
                    if (acc != No) rp = RunPlan();                 // (cannot happen: every observation was counted once)
                }
            }
        }
    }
    const bool run_path = rp.n > 0;
    // Schur chunks: <= SCH_CHUNK co-observation pairs of one block per wavefront
    // A lane may take several pairs of its chunk (64 per pass) and add them serially before the wave's reduce-scatter.  Measured
    // (profiles/r01_v8_schur_passes.log): two passes win 4-5 % on a lone mid-size window (C2: half the waves, half the partials
    // for k_schur_finalize to sum, all of them still resident in one round) and lose 4-7 % once the launch fills the machine
    // (C4, batched windows: the serial variant needs 216 VGPRs, 2 waves per SIMD instead of 4).  Results differ from the
    // one-pass chunking in the last bits only (summation order).
    int64_t chunks64 = 0;
    for (int b = 0; b < n_blk; ++b) chunks64 += (blk_ptr[b + 1] - blk_ptr[b] + SCH_CHUNK - 1) / SCH_CHUNK;
    int sch_passes = (!w.batch_member && chunks64 >= 1024 && chunks64 <= 6144) ? 2 : 1;
    { const char* e = std::getenv("VISFS_BA_SCH_PASSES"); if (e) { const int q = std::atoi(e); if (q >= 1 && q <= 8) sch_passes = q; } }
    const int sch_chunk = SCH_CHUNK * sch_passes;
    std::vector<int32_t> blk_chunk_ptr(n_blk + 1, 0), sch_blk, sch_ptr;
    if (!run_path) { sch_blk.reserve((size_t)(npairs / sch_chunk) + n_blk + 1); sch_ptr.reserve((size_t)(npairs / sch_chunk) + n_blk + 1); }
    for (int b = 0; b < n_blk; ++b) {
        blk_chunk_ptr[b] = (int32_t)sch_blk.size();
        if (run_path) continue;                             // (no pair lists, no chunks: the runs replace them)
        for (int e = blk_ptr[b]; e < blk_ptr[b + 1]; e += sch_chunk) { sch_blk.push_back(b); sch_ptr.push_back(e); }
    }
    blk_chunk_ptr[n_blk] = (int32_t)sch_blk.size();
    const int n_sch = (int)sch_blk.size();
    std::vector<int4> sch_desc(std::max(n_sch, 1)), blk_desc(2 * (size_t)std::max(n_blk, 1));
    for (int c = 0; c < n_sch; ++c) {
        const int b = sch_blk[c];
        sch_desc[c] = make_int4(sch_ptr[c], std::min(sch_ptr[c] + sch_chunk, blk_ptr[b + 1]), free_pose[blk_i[b]], free_pose[blk_j[b]]);
    }
    for (int b = 0; b < n_blk; ++b) {
        const int a = blk_i[b];
        const bool dg = (a == blk_j[b]);
        blk_desc[2 * b] = make_int4(blk_chunk_ptr[b], blk_chunk_ptr[b + 1], dg ? pose_odo_ptr[a] : blk_odo_ptr[b], dg ? pose_odo_ptr[a + 1] : blk_odo_ptr[b + 1]);
        if (run_path) {
            const int cnt_r = rp.last[b] >= 0 ? rp.last[b] - rp.first[b] + 1 : 0;
            blk_desc[2 * b].x = (int)((unsigned)(cnt_r ? rp.first[b] : 0) | ((unsigned)cnt_r << 20));
            blk_desc[2 * b].y = (int)((unsigned)free_pose[a] | ((unsigned)free_pose[blk_j[b]] << 16));
        }
        blk_desc[2 * b + 1] = make_int4(a, blk_j[b], pose_chunk_ptr[a], pose_chunk_ptr[a + 1]);
    }
    if (prm.solver == 2 && Npf > MAX_PCG_FREE_POSES) { h->err = "Optimizer/Solver=2 (PCG) supports at most 1024 free poses; use the direct solver"; return VISFS_BA_ERR_UNSUPPORTED; }
    int max_row = 0;
    for (int a = 0; a < Npf; ++a) max_row = std::max(max_row, row_ptr[a + 1] - row_ptr[a]);
    // persistent PCG: LDS plan.  d, q, scalars, the rows' column/code tables always; s = Minv r when an owner thread holds several
    // blocks; the own block rows of S when they fit.  Co-residency of the hand-off needs every workgroup resident: one block row
    // per workgroup up to 256 free poses, ceil(Npf / 256) rows per workgroup beyond (at most 256 workgroups, one per CU).
    const int pcg_rpw = Npf > MAX_PCG_ONE_ROW_POSES ? (Npf + MAX_PCG_ONE_ROW_POSES - 1) / MAX_PCG_ONE_ROW_POSES : 1;
    size_t pcg_lds = (size_t)(2 * 6 * Npf + 32 + 32 * pcg_rpw + (Npf > MAX_PCG_ONE_ROW_POSES ? 6 * Npf : 0)) * 8 + (size_t)8 * max_row * pcg_rpw + 16;
    int lds_minv = 0, lds_srow = 0;                      // Minv: registers (one block per owner) or, beyond 256 free poses, read from HBM
    {
        const size_t budget = (size_t)150 * 1024;        // one workgroup per CU may own most of its 160 KiB
        if (pcg_lds + (size_t)288 * max_row * pcg_rpw <= budget) { lds_srow = 1; pcg_lds += (size_t)288 * max_row * pcg_rpw; }
    }
    if (prm.solver == 2 && pcg_lds > (size_t)160 * 1024) { h->err = "reduced camera system too large for the persistent PCG (LDS); use the direct solver"; return VISFS_BA_ERR_UNSUPPORTED; }

    // k_pcg1 (one wavefront per block row, <= 64 free poses): the block of S at (i, a) as a dense code table
    const bool pcg1 = [&]() { const char* e = std::getenv("VISFS_BA_PCG1"); return prm.solver == 2 && Npf >= 1 && Npf <= 64 && !(e && e[0] == '0'); }();
    std::vector<int32_t> pcg1_code;
    if (pcg1) {
        pcg1_code.assign((size_t)Npf * Npf, -1);
        for (int a = 0; a < Npf; ++a)
            for (int n = row_ptr[a]; n < row_ptr[a + 1]; ++n) pcg1_code[(size_t)a * Npf + row_col[n]] = row_blk[n];
    }
    // k_pcg_cu: the whole PCG in one workgroup when every block row is short enough to sit in registers (3 threads per scalar row)
    // One window: slower than k_pcg1 (34.7 vs 21.6 us per solve at C2, profiles/r02_pcg_cu_vs_handoff.log).  Many windows sharing every
    // launch: nothing spins and no hand-off is paid — 16 C2 windows 68.0 -> 70.6 k it/s, 32 windows 66.6 -> 74.3 k, but 8 windows
    // 58.8 -> 55.6 k (profiles/r02_v3_pcg_cu_batches.log).  Its mat-vec sums associate differently from k_pcg1's, so a window's result
    // would depend on how many OTHER windows were submitted with it if the kernel followed the batch size: it follows the HANDLE
    // (visfs_ba_set_tuning(THROUGHPUT), ABI 8; VISFS_BA_PCG_CU=0|1 overrides), never the batch — every window is bit-identical to
    // its solve as a batch of one through the same handle however a batch is cut or sharded.
    const bool pcg_cu = [&]() { const char* e = std::getenv("VISFS_BA_PCG_CU"); int mr = 0; for (int a = 0; a < Npf; ++a) mr = std::max(mr, row_ptr[a + 1] - row_ptr[a]);
                                const bool want = e ? e[0] == '1' : h->tuning == VISFS_BA_TUNE_THROUGHPUT;      // (the environment overrides every handle)
                                return prm.solver == 2 && pcg_cu_fits(Npf, mr) && 6 * Npf > 64 && want; }();
    // ... and where each stored block sits in the row lists: k_schur_finalize writes S a second time by scalar row for that kernel
    const int cu_T = (int)((6 * (size_t)Npf + 63) / 64 * 64);
    int cu_max_row = 0;
    std::vector<int32_t> blk_slot(std::max(n_blk, 1), 255 | (255 << 8));
    for (int a = 0; a < Npf; ++a) {
        cu_max_row = std::max(cu_max_row, row_ptr[a + 1] - row_ptr[a]);
        if (pcg_cu)
            for (int n = row_ptr[a]; n < row_ptr[a + 1]; ++n) {
                const int b = row_blk[n] >> 1, k = n - row_ptr[a];
                if (row_blk[n] & 1) blk_slot[b] = (blk_slot[b] & 0xff) | (k << 8); else blk_slot[b] = (blk_slot[b] & ~0xff) | k;
            }
    }
    // direct solver (Optimizer/Solver 0, 1, 3 and every solve of the Ceres branch): S of a sliding window is block-banded — the banded
    // factorisation in one workgroup (k_band_chol) when the band is narrow enough, the dense blocked Cholesky otherwise (VISFS_BA_BAND=0 forces it)
    int band_B = -1, band_rows = 0, band_lds = 0;
    std::vector<int32_t> band_code;
    // (Optimizer/Solver=2 windows get the plan too: it is what a solve falls back to when the persistent PCG's hand-off times out — a GPU
    // kept busy by another process —, so that the call still returns a solution as the reference's solver always does, Optimizer.cpp:76-91)
    if (Npf >= 1) {
        int Bw = 0;
        for (int b = 0; b < n_blk; ++b) Bw = std::max(Bw, blk_j[b] - blk_i[b]);
        const char* e = std::getenv("VISFS_BA_BAND");
        if (!(e && e[0] == '0') && band_plan(Npf, Bw, &band_rows, &band_lds)) {
            { const char* er = std::getenv("VISFS_BA_BAND_ROWS"); const int q = er ? std::atoi(er) : 0;          // tests: force the streaming form
              if (q >= Bw + 3 && q < band_rows) { band_rows = q; band_lds = (int)band_lds_bytes(Npf, Bw, q); } }
            band_B = Bw;
            band_code.assign((size_t)Npf * (Bw + 1), -1);
            for (int b = 0; b < n_blk; ++b) band_code[(size_t)blk_j[b] * (Bw + 1) + (blk_j[b] - blk_i[b])] = b;
        }
    }
    lap("structure");
    // lanes per landmark: smallest power of two >= mean track length, in [4, 64] — but at most 8 once the window has a thousand landmarks:
    // the landmark-major kernels of a 200-key-frame window or of a batch of 50-key-frame windows need more than one round of resident
    // waves at 16 lanes per landmark (k_backsub: 121 VGPRs, four waves per SIMD), and a lane that loops twice over a track of ten costs
    // less than the second round (measured: C4 5 631 -> 6 007 it/s, C4R 5 040 -> 5 330, 16 resident C2 windows 73.0 k -> 80.9 k,
    // 8 windows 63.1 k -> 66.2 k; one C2 window flat: 20.3 k / 20.2 k).  A property of the WINDOW alone: a window's result never
    // depends on what it is batched with.
    int group = 4;
    const double mean_track = Nl > 0 ? (double)No / Nl : 1.0;
    while (group < 64 && group < mean_track) group *= 2;
    if (Nl >= 1024 && group > 8) group = 8;
    { const char* e = std::getenv("VISFS_BA_GROUP"); if (e) { const int gq = std::atoi(e); if (gq == 4 || gq == 8 || gq == 16 || gq == 32 || gq == 64) group = gq; } }   // tuning override
    const int n_lin_a = std::max(1, (Nl + (256 / group) - 1) / (256 / group));
    const int n_eval = (No + 255) / 256 + 1;
    const int n_parts = std::max(n_lin_a + 1, n_eval);
    const size_t n6 = (size_t)6 * Npf;
    const size_t chol_np = std::max<size_t>(32, (n6 + 31) / 32 * 32);
    const bool small_solve_fits_npf = Npf >= 1 && n6 <= (size_t)SM_MAX_N6;
    const size_t n_hist = (size_t)std::max(index_blocks(No), 1) * std::max(Npf, 1);

    // ---- the INDEX section (block-level structure of S, small) and the mutable section: one device arena, the index part staged in
    // pinned memory with the same layout; everything of size N_obs that is derived (lm_ptr, obs_ok, pose_obs, obs_ppos, the pair
    // lists) is built on the device from the primary section
    auto layout = [&](Arena& A, DeviceGraph& g) {
        g.chunk_pose = A.take<int32_t>(std::max(n_chunks, 1));
        g.chunk_ptr = A.take<int32_t>(n_chunks + 1);
        g.pose_chunk_ptr = A.take<int32_t>(Npf + 1);
        g.pose_odo_ptr = A.take<int32_t>(Npf + 1);
        g.pose_odo = A.take<int32_t>(std::max<size_t>(pose_odo.size(), 1));
        g.blk_i = A.take<int32_t>(std::max(n_blk, 1));
        g.blk_j = A.take<int32_t>(std::max(n_blk, 1));
        g.blk_ptr = A.take<int32_t>(n_blk + 1);
        g.blk_chunk_ptr = A.take<int32_t>(n_blk + 1);
        g.sch_desc = A.take<int4>(std::max(n_sch, 1));
        g.blk_desc = A.take<int4>(2 * (size_t)std::max(n_blk, 1));
        g.blk_odo_ptr = A.take<int32_t>(n_blk + 1);
        g.blk_odo = A.take<int32_t>(std::max<size_t>(blk_odo.size(), 1));
        g.row_ptr = A.take<int32_t>(Npf + 1);
        g.row_col = A.take<int32_t>(std::max<size_t>(row_col.size(), 1));
        g.row_blk = A.take<int32_t>(std::max<size_t>(row_blk.size(), 1));
        g.blk_slot = A.take<int32_t>(std::max(n_blk, 1));
        g.fin_exp = A.take<int32_t>(std::max(n_blk, 1));
        g.sch_blk = A.take<int32_t>(std::max(n_sch, 1));
        g.diag_blk = A.take<int32_t>(std::max(Npf, 1));
        g.pcg1_code = pcg1 ? A.take<int32_t>((size_t)Npf * Npf) : nullptr;
        g.band_code = band_B >= 0 ? A.take<int32_t>(band_code.size()) : nullptr;
        g.run_desc = run_path ? A.take<int4>(rp.desc.size()) : nullptr;
        g.run_k0 = run_path ? A.take<int32_t>(rp.k0.size()) : nullptr;
    };
    Arena sizing{ nullptr, 0, 0 };
    layout(sizing, hg);
    const size_t static_bytes = (sizing.used + 255) & ~size_t(255);

    Arena dyn{ nullptr, 0, static_bytes };
    DeviceGraph dg{};
    int32_t* d_hist = nullptr;
    size_t zero_end = 0;
    auto layout_dyn = [&](Arena& A, DeviceGraph& g) {
        // device-built index arrays
        g.lm_ptr = A.take<int32_t>(Nl + 1);
        g.obs_ok = A.take<uint8_t>(std::max(No, 1));
        g.obs_ppos = A.take<int32_t>(std::max(No, 1));
        g.pose_obs = A.take<int32_t>(std::max(n_pose_obs, 1));
        if (pz.raw_refs) g.obs_uvr = A.take<double>((size_t)std::max(No, 1) * 3);      // written by k_index_count from obs_uvd
        d_hist = A.take<int32_t>(n_hist);
        for (int k = 0; k < 2; ++k) {                       // the two linearisation sets: same layout, constant distance
            LinBuf& L = g.lin[k];
            L.obs_w = A.take<double>(std::max(No, 1));
            L.obs_pcw = A.take<double>((size_t)std::max(No, 1) * 4);
            L.pose_pcw = A.take<double>((size_t)std::max(n_pose_obs, 1) * 4);
            L.Hll = A.take<double>((size_t)std::max(Nl, 1) * 6);
            L.bl = A.take<double>((size_t)std::max(Nl, 1) * 3);
            L.hpp_part = A.take<double>((size_t)std::max(n_chunks, 1) * 27);
            L.odo_blk = A.take<double>((size_t)(Ne + 1) * 120);
        }
        g.lin_stride = reinterpret_cast<char*>(g.lin[1].obs_w) - reinterpret_cast<char*>(g.lin[0].obs_w);
        g.pose[0] = A.take<double>((size_t)Np * POSE_STRIDE); g.pose[1] = A.take<double>((size_t)Np * POSE_STRIDE);
        g.pt[0] = A.take<double>((size_t)std::max(Nl, 1) * 3); g.pt[1] = A.take<double>((size_t)std::max(Nl, 1) * 3);
        g.obs_outlier = A.take<uint8_t>(std::max(No, 1));   // (right behind the estimates: pose[0] .. obs_outlier are one span in the order and alignment of
                                                             // output_stage_of, so a solve's outputs come back in ONE device-to-host copy)
        g.obs_level = A.take<uint8_t>(std::max(No, 1));
        g.obs_chi2_out = A.take<double>(std::max(No, 1));
        g.obs_chi2 = A.take<double>(std::max(No, 1));
        g.Hpp = A.take<double>((size_t)std::max(Npf, 1) * 36);
        g.bp = A.take<double>(std::max<size_t>(n6, 1));
        g.lin_part = A.take<double>((size_t)n_parts * 2);
        g.S = A.take<double>((size_t)std::max(n_blk, 1) * 36);
        g.S_rows = A.take<double>(pcg_cu ? (size_t)cu_max_row * 6 * cu_T : 1);      // (inside the span the upload clears: slots a row does not use read as zero)
        g.bs = A.take<double>(std::max<size_t>(n6, 1));
        g.Minv = A.take<double>((size_t)std::max(Npf, 1) * 36);
        g.x = A.take<double>(std::max<size_t>(n6, 1));
        g.sch_part = A.take<double>((size_t)std::max<int64_t>(std::max<int64_t>(n_sch, rp.total), 1) * 42);
        g.granules = A.take<unsigned long long>(std::max<size_t>(4 * n6 + Npf, 1));        // + one placement word per block row (k_pcg1)
        g.dxl = A.take<double>((size_t)std::max(Nl, 1) * 3);
        g.trial_part = A.take<double>((size_t)n_parts * 2);
        g.trial_gran = A.take<unsigned long long>((size_t)n_parts * 4);
        g.fin_flag = A.take<uint32_t>((size_t)std::max(n_blk, 1));
        g.fin_cnt = A.take<uint32_t>((size_t)std::max(n_blk, 1));
        g.aux_part = A.take<double>((size_t)n_parts);
        g.dl_part = A.take<double>(ceres && prm.trust_region == 1 ? (size_t)n_parts * 4 : 1);
        g.s2l = A.take<double>((size_t)std::max(Nl, 1) * 3);
        g.s2p = A.take<double>(std::max<size_t>(n6, 1));
        const bool dense_chol = band_B < 0 && (prm.solver != 2 || (!small_solve_fits_npf && chol_np <= 2048));   // (PCG windows: the fallback's scratch, up to 67 MB)
        g.dense = A.take<double>(dense_chol ? chol_np * chol_np : 1);
        g.chol_f = A.take<double>(dense_chol ? chol_np * chol_np : 1);
        g.band_L = A.take<double>(band_B >= 0 && band_rows < Npf ? (size_t)Npf * (band_B + 1) * 36 : 1);
        g.chol_y = A.take<double>(chol_np);
        g.chol_linv = A.take<double>(2 * 32 * 32);
        g.stamps = A.take<unsigned long long>(128);
        g.st = w.d_state;                                   // (its own allocation: the address does not move with the window's sizes)
        // ---- from here on: arrays that are written in full before anything reads them — not part of the upload's clearing pass (a third
        // of the arena at C2: the stage hooks' H_pl tiles and residuals exist for every window but are only written under `debug`)
        zero_end = (A.used + 255) & ~size_t(255);
        g.blk_pairs = A.take<int4>((size_t)std::max<int64_t>(run_path ? 0 : npairs, 1));      // every slot filled on the device (k_build_pairs: the counts are the summary's)
        g.pose_lm = A.take<int32_t>(std::max(n_pose_obs, 1));                   // every slot filled by k_index_scatter
        g.pose_rec = A.take<DeviceGraph::PoseRec>(std::max(n_pose_obs, 1));     // likewise
        g.obs_err = A.take<double>((size_t)std::max(No, 1) * 3);               // debug: every observation, active or not (k_linearize)
        g.W = A.take<double>((size_t)std::max(No, 1) * 18);                    // likewise
    };
    layout_dyn(dyn, dg);
    const size_t total_bytes = (dyn.used + 255) & ~size_t(255);

    if (w.d_cap < total_bytes) {
        HIP_TRY(h, hipStreamSynchronize(w.stream));
        if (w.d_base) (void)hipFree(w.d_base);
        w.d_base = nullptr; w.d_cap = 0;
        const size_t want = total_bytes + total_bytes / 8;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&w.d_base), want));
        w.d_cap = want;
    }
    // the pinned arena also receives the outputs of a solve (ws_read_state): both estimate buffers and the outlier flags
    size_t host_need = static_bytes;
    { DeviceGraph sz{}; sz.Np = Np; sz.Nl = Nl; sz.No = No; host_need = std::max(host_need, output_stage_bytes(sz)); }
    if (w.h_cap < host_need) {
        HIP_TRY(h, hipStreamSynchronize(w.stream));
        if (w.h_base) (void)hipHostFree(w.h_base);
        w.h_base = nullptr; w.h_cap = 0;
        const size_t want = host_need + host_need / 8;
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&w.h_base), want, hipHostMallocDefault));
        w.h_cap = want;
    }
    lap("alloc");
    // fill the index staging
    Arena hs{ w.h_base, w.h_cap, 0 };
    layout(hs, hg);
    {
        if (n_chunks) std::memcpy(const_cast<int32_t*>(hg.chunk_pose), chunk_pose.data(), (size_t)n_chunks * 4);
        std::memcpy(const_cast<int32_t*>(hg.chunk_ptr), chunk_ptr.data(), (size_t)(n_chunks + 1) * 4);
        std::memcpy(const_cast<int32_t*>(hg.pose_chunk_ptr), pose_chunk_ptr.data(), (size_t)(Npf + 1) * 4);
        std::memcpy(const_cast<int32_t*>(hg.pose_odo_ptr), pose_odo_ptr.data(), (size_t)(Npf + 1) * 4);
        if (!pose_odo.empty()) std::memcpy(const_cast<int32_t*>(hg.pose_odo), pose_odo.data(), pose_odo.size() * 4);
        if (n_blk) { std::memcpy(const_cast<int32_t*>(hg.blk_i), blk_i.data(), (size_t)n_blk * 4); std::memcpy(const_cast<int32_t*>(hg.blk_j), blk_j.data(), (size_t)n_blk * 4); }
        std::memcpy(const_cast<int32_t*>(hg.blk_ptr), blk_ptr.data(), (size_t)(n_blk + 1) * 4);
        std::memcpy(const_cast<int32_t*>(hg.blk_chunk_ptr), blk_chunk_ptr.data(), (size_t)(n_blk + 1) * 4);
        std::memcpy(const_cast<int4*>(hg.sch_desc), sch_desc.data(), sch_desc.size() * sizeof(int4));
        std::memcpy(const_cast<int4*>(hg.blk_desc), blk_desc.data(), blk_desc.size() * sizeof(int4));
        std::memcpy(const_cast<int32_t*>(hg.blk_odo_ptr), blk_odo_ptr.data(), (size_t)(n_blk + 1) * 4);
        if (!blk_odo.empty()) std::memcpy(const_cast<int32_t*>(hg.blk_odo), blk_odo.data(), blk_odo.size() * 4);
        std::memcpy(const_cast<int32_t*>(hg.row_ptr), row_ptr.data(), (size_t)(Npf + 1) * 4);
        if (!row_col.empty()) { std::memcpy(const_cast<int32_t*>(hg.row_col), row_col.data(), row_col.size() * 4); std::memcpy(const_cast<int32_t*>(hg.row_blk), row_blk.data(), row_blk.size() * 4); }
        if (n_blk) std::memcpy(const_cast<int32_t*>(hg.blk_slot), blk_slot.data(), (size_t)n_blk * 4);
        {
            int32_t* fe = const_cast<int32_t*>(hg.fin_exp); int32_t* db = const_cast<int32_t*>(hg.diag_blk);
            for (int b = 0; b < n_blk; ++b) {
                const int a = blk_i[b];
                const bool dgb = (a == blk_j[b]);
                fe[b] = (blk_chunk_ptr[b + 1] - blk_chunk_ptr[b]) | ((dgb ? pose_chunk_ptr[a + 1] - pose_chunk_ptr[a] : 0) << 16);
                if (dgb) db[a] = b;
            }
            if (n_sch) std::memcpy(const_cast<int32_t*>(hg.sch_blk), sch_blk.data(), (size_t)n_sch * 4);
        }
        if (pcg1) std::memcpy(const_cast<int32_t*>(hg.pcg1_code), pcg1_code.data(), pcg1_code.size() * 4);
        if (band_B >= 0) std::memcpy(const_cast<int32_t*>(hg.band_code), band_code.data(), band_code.size() * 4);
        if (run_path) { std::memcpy(const_cast<int4*>(hg.run_desc), rp.desc.data(), rp.desc.size() * sizeof(int4)); std::memcpy(const_cast<int32_t*>(hg.run_k0), rp.k0.data(), rp.k0.size() * 4); }
    }
    lap("stage fill");
    // device pointers: same offsets
    { Arena dp{ w.d_prim, w.d_prim_cap, 0 }; layout_prim(dp, dg, pz); }
    Arena ds{ w.d_base, w.d_cap, 0 };
    layout(ds, dg);
    Arena dd{ w.d_base, w.d_cap, static_bytes };
    layout_dyn(dd, dg);
    dg.Np = Np; dg.Nl = Nl; dg.No = No; dg.Ne = Ne; dg.Npf = Npf;
    dg.n_pose_obs = n_pose_obs;
    dg.n_chunks = n_chunks; dg.n_blk = n_blk; dg.n_lin_a = n_lin_a; dg.group = group; dg.n_edges_ok = n_edges_ok;
    // The finalisation of S as the prologue of the one-wave PCG launch (k_pcg1<FIN>; VERDICT r03 item 9's "drop the k_schur_finalize launch").
    // MEASURED (profiles/r04_fin_pcg_fusion_ab.log): bit-identical, and no faster — the fused launch takes 27.0 us where the two launches take
    // 5.65 + 20.9 (the in-launch hand-off, a drain + flag + sc1 reload, costs what the kernel boundary cost; the finalising waves run at
    // k_pcg1's one wave per SIMD); C2 19.8-20.0 k against 20.2 k it/s, 16 resident windows 65 k against 80 k.  Opt-in: VISFS_BA_FIN_PCG=1.
    { const char* e = std::getenv("VISFS_BA_FIN_PCG"); const char* gv = std::getenv("VISFS_BA_PCG_GATHER");
      dg.fin_pcg = (pcg1 && !pcg_cu && !small_solve_fits_npf && (e && e[0] == '1') && !(gv && std::atoi(gv) != 1)) ? 1 : 0; }
    dg.n_runs = rp.n; dg.run_lr = rp.LR; dg.run_m = rp.M; dg.run_cap = rp.cap; dg.run_wmax = rp.wmax; dg.run_lds_bytes = (int32_t)rp.lds;
    dg.n_sch = n_sch; dg.sch_chunk = sch_chunk; dg.pcg_lds_minv = lds_minv; dg.pcg_lds_srow = lds_srow; dg.pcg_max_row = max_row; dg.pcg_rows_per_wg = pcg_rpw; { // fin_arrive (opt-in, VISFS_BA_FIN_ARRIVE=1): needs every stored block to own at least one gather chunk (somebody has to arrive) and counts that fit 16 bits
      const char* e = std::getenv("VISFS_BA_FIN_ARRIVE");
      bool ok = e && e[0] == '1' && !w.batch_member && !ceres && !run_path && !small_solve_fits_npf && n_blk > 0;
      for (int b = 0; ok && b < n_blk; ++b) ok = blk_chunk_ptr[b + 1] > blk_chunk_ptr[b] && blk_chunk_ptr[b + 1] - blk_chunk_ptr[b] < 65536;
      for (int a = 0; ok && a < Npf; ++a) ok = pose_chunk_ptr[a + 1] - pose_chunk_ptr[a] < 32768;
      dg.fin_arrive = ok ? 1 : 0; }
    dg.pcg_cu = pcg_cu ? 1 : 0; dg.cu_T = cu_T; dg.pcg_lds_bytes = pcg_cu ? (int32_t)pcg_cu_lds_bytes(Npf, max_row) : (int32_t)pcg_lds; dg.chol_np = (int32_t)chol_np;
    dg.band_B = band_B; dg.band_rows = band_rows; dg.band_lds_bytes = band_lds;
    dg.fx = gr->fx; dg.fy = gr->fy; dg.cx = gr->cx; dg.cy = gr->cy; dg.bf = gr->bf;
    dg.stereo_baseline = opt.baseline;
    dg.inv_pixel_var = 1.0 / prm.pixel_variance;          // Optimizer.cpp:153
    dg.inv_pixel_var_out = dg.inv_pixel_var;
    dg.ceres = ceres ? 1 : 0;
    dg.dogleg = (ceres && prm.trust_region == 1) ? 1 : 0;   // Optimizer.cpp:515-519: options.trust_region_strategy_type = ceres::DOGLEG
    dg.inv_odo_cov = 1.0 / prm.odometry_covariance;       // Optimizer.cpp:117-121
    dg.huber_delta = prm.robust_kernel_delta;             // Optimizer.cpp:212-216
    dg.Nz = Nz; dg.laser_pose = Nz ? gr->laser_pose : 0;
    dg.inv_laser_cov = 1.0 / prm.laser_covariance;        // Optimizer.cpp:232
    // Ceres branch: residual = info * e, so the squared norm carries info^2 (StereoObservationFactor.cpp:25-26, OccupiedSpace2dFactor.cpp:45)
    if (ceres) { dg.inv_pixel_var *= dg.inv_pixel_var; dg.inv_laser_cov *= dg.inv_laser_cov; }
    std::memcpy(dg.Tcr, gr->Tcr, 96);
    if (Nz) { dg.grid.nx = gr->grid->num_x_cells; dg.grid.ny = gr->grid->num_y_cells; dg.grid.resolution = gr->grid->resolution;
              dg.grid.max_x = gr->grid->max_x; dg.grid.max_y = gr->grid->max_y; }
    dg.debug = 0;
    { const char* e = std::getenv("VISFS_BA_STAMP_WG"); dg.stamp_wg = e ? std::atoi(e) : 0; }
    { const char* e = std::getenv("VISFS_BA_FAULT_PCG_TIMEOUT"); dg.fault_pcg = (e && e[0] == '1') ? 1 : 0; }      // test hook: force the time-out path once per solve
    w.g = dg;
    // the same graph at its fixed device address (what the launch sequences of the per-frame path read: Many{ d_graph, d_state })
    *w.h_graph = dg;
    HIP_TRY(h, hipMemcpyAsync(w.d_graph, w.h_graph, sizeof(DeviceGraph), hipMemcpyHostToDevice, w.stream));
    // opt-in (VISFS_BA_FUSED=1): one CU's fp64 rate makes the fused kernel slower than the multi-kernel path per window
    // (DESIGN.md §4); it pays only when many small windows run side by side
    { const char* e = std::getenv("VISFS_BA_SMALL_SOLVE"); w.small_solve = small_solve_fits(dg) && !(e && e[0] == '0'); }
    { const char* e = std::getenv("VISFS_BA_FUSED"); w.fused = small_path_fits(dg) && e && e[0] == '1'; }
    // speculative linearise: a rejected trial wastes one linearisation, an accepted one saves k_decide + a launch gap — worth it
    // while the linearisation is cheap (latency-bound windows); large windows (C4: half the trials are rejected) and batch
    // members keep the gated form.
    { const char* e = std::getenv("VISFS_BA_SPEC"); w.spec = (e ? (e[0] == '1') : (No <= 150000)) && Np <= MAX_STAGED_POSES; }
    // Optimizer/Framework=1 runs the plain gated unit with the direct solver (k_small_solve's Cholesky for reduced systems <= 64 x 64):
    // one unit = one iteration of Ceres' minimizer loop
    if (ceres) { w.fused = false; w.spec = false; }
    // the fused form of the speculative unit: one launch less per iteration (VISFS_BA_SPEC_FUSED=0: the two-launch form, A/B runs and tests)
    { const char* e = std::getenv("VISFS_BA_SPEC_FUSED"); w.spec_fused = w.spec && !(e && e[0] == '0'); }
    // Batch members (round 4) CAN run the same fused unit — 4 launches per iteration instead of the gated unit's 6; the two-launch form
    // does not exist for them.  Measured (profiles/r04_batch_fused_unit_ab.log): a batched launch is bound by its arithmetic, not by its
    // launch boundaries — fusing moves the linearisation's work into k_backsub (at 128 instead of 96 VGPRs) and behind the gather without
    // removing any of it: 16 x C2 80.3 k -> 74.4 k it/s, 8 x C2 66.9 k -> 65.8 k.  So the gated unit stays the default for batch members
    // (VISFS_BA_BATCH_SPEC=1 selects the fused unit: tests, A/B runs).
    if (w.batch_member) { const char* e = std::getenv("VISFS_BA_BATCH_SPEC"); if (!w.spec_fused || !(e && e[0] == '1')) { w.spec = false; w.spec_fused = false; } }
    // default: on for a window on its own, off for batch members — measured in round 4 (profiles/r04_batch_decide_fused_ab.log): 8 windows
    // 71.1 -> 68.7 k it/s, 16 windows 90.3 -> 83.7 k with the decision on board k_backsub<Many> (VISFS_BA_DECIDE_FUSED=1 forces it on for both)
    { const char* e = std::getenv("VISFS_BA_DECIDE_FUSED"); w.fused_decide = (e ? (e[0] != '0') : !w.batch_member) && !ceres; }
    w.direct_ready = prm.solver == 2 && !ceres && Npf >= 1 && !w.small_solve && (band_B >= 0 || chol_np <= 2048);
    w.solver_now = -1;
    w.n_pairs = npairs; w.device_bytes = total_bytes + pbytes;
    w.free_pose = free_pose; w.blk_i = blk_i; w.blk_j = blk_j; w.pose_free = pose_free;
    w.odo_i.assign(gr->odo_from, gr->odo_from + Ne); w.odo_j.assign(gr->odo_to, gr->odo_to + Ne);
    if (static_bytes) HIP_TRY(h, hipMemcpyAsync(w.d_base, w.h_base, static_bytes, hipMemcpyHostToDevice, w.stream));
    HIP_TRY(h, hipMemsetAsync(w.d_base + static_bytes, 0, zero_end - static_bytes, w.stream));
    if (configure_kernels(w.g) != 0) { h->err = "hipFuncSetAttribute failed"; return VISFS_BA_ERR_DEVICE; }
    launch_build_index(w.g, d_hist, w.stream);             // lm_ptr, obs_ok, pose_obs / obs_ppos: never exist on the host
    launch_build_pairs(w.g, w.stream);                     // the co-observation pair lists never exist on the host
    launch_reset(w.g, ceres ? prm.iterations : prm.iterations / 2, prm.trust_region == 1, 1, w.stream);
    HIP_TRY(h, hipGetLastError());
    lap("enqueue");
    // no synchronisation here: the launches that follow queue up behind the copies; the pinned staging arenas are only reused by the
    // NEXT upload, which drains the stream first
    if (timing) { HIP_TRY(h, hipStreamSynchronize(w.stream)); lap("h2d+sync"); }
    w.upload_in_flight = true;
    w.loaded = true;
    w.pristine = true;
    return VISFS_BA_OK;
}

// Layout of the outputs in the staging arena when they travel with the state read: both pose buffers, both landmark buffers, outliers.
struct OutputStage { size_t pose[2], pt[2], out, end; };
static OutputStage output_stage_of(const DeviceGraph& g) {
    OutputStage o;
    const size_t b_pose = (size_t)g.Np * POSE_STRIDE * 8, b_pt = (size_t)g.Nl * 24;
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    o.pose[0] = 0; o.pose[1] = up(b_pose); o.pt[0] = o.pose[1] + up(b_pose); o.pt[1] = o.pt[0] + up(b_pt); o.out = o.pt[1] + up(b_pt);
    o.end = o.out + (size_t)g.No;
    return o;
}

int ws_read_state(visfs_ba_handle* h, Workspace& w) {
    HIP_TRY(h, hipMemcpyAsync(w.h_state, w.g.st, sizeof(LmState), hipMemcpyDeviceToHost, w.stream));
    w.outputs_staged = false;
    const OutputStage os = output_stage_of(w.g);
    // (the arena is also the source of the upload's H2D copy: that copy precedes these on the same stream, so it has read its data)
    const bool with_outputs = w.want_outputs && w.h_base && os.end <= w.h_cap;
    if (with_outputs) {
        // which estimate buffer is the final one is only known from the state that is coming back: fetch both (poses are tiny, landmarks
        // 24 bytes each) — the caller's download then needs neither a copy nor a second synchronisation
        const DeviceGraph& g = w.g;
        const char* d0 = reinterpret_cast<const char*>(g.pose[0]);
        const bool one_span = reinterpret_cast<const char*>(g.pose[1]) - d0 == (ptrdiff_t)os.pose[1] && reinterpret_cast<const char*>(g.pt[0]) - d0 == (ptrdiff_t)os.pt[0] &&
                              reinterpret_cast<const char*>(g.pt[1]) - d0 == (ptrdiff_t)os.pt[1] && reinterpret_cast<const char*>(g.obs_outlier) - d0 == (ptrdiff_t)os.out;
        if (one_span) {
            // the device arena holds the five arrays in the staging layout (ws_upload): one copy instead of five — each is a few microseconds
            // of copy-engine latency on the tail of every per-frame call
            HIP_TRY(h, hipMemcpyAsync(w.h_base, d0, os.end, hipMemcpyDeviceToHost, w.stream));
        } else {
            for (int k = 0; k < 2; ++k) {
                HIP_TRY(h, hipMemcpyAsync(w.h_base + os.pose[k], g.pose[k], (size_t)g.Np * POSE_STRIDE * 8, hipMemcpyDeviceToHost, w.stream));
                if (g.Nl) HIP_TRY(h, hipMemcpyAsync(w.h_base + os.pt[k], g.pt[k], (size_t)g.Nl * 24, hipMemcpyDeviceToHost, w.stream));
            }
            if (g.No) HIP_TRY(h, hipMemcpyAsync(w.h_base + os.out, g.obs_outlier, (size_t)g.No, hipMemcpyDeviceToHost, w.stream));
        }
    }
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    w.outputs_staged = with_outputs;
    if (!w.recs.empty()) prof_harvest(w);
    return VISFS_BA_OK;
}

// One unit of the LM state machine (gated on the device; see ba_kernels.hip header).  `first` = first unit of a
// phase: only there lambda has to be initialised from max|diag H| (k_lin_finalize).
// w.spec ("speculative linearise"): only the first unit of a phase linearises up front; every unit ENDS with the launch that
// linearises its trial state beside the LM decision (k_linearize spec = 1), so k_decide and its launch leave the critical path.
void enqueue_unit(visfs_ba_handle* h, Workspace& w, bool first) {
    const int solver = w.solver_now >= 0 ? w.solver_now : h->prm.solver;
    if (!w.spec || first) { ProfScope p(w, VISFS_BA_K_LINEARIZE, w.g.Ne == 0 && w.g.Nz == 0); launch_linearize(w.g, w.stream); }
    if (w.g.ceres) { ProfScope p(w, VISFS_BA_K_LIN_FINALIZE, true); launch_ceres_lin_finalize(w.g, w.stream); }
    else if (first) { ProfScope p(w, VISFS_BA_K_LIN_FINALIZE, true); launch_lin_finalize(w.g, 0, w.stream); }
    { ProfScope p(w, VISFS_BA_K_SCHUR, true); if (w.spec_fused && !first) launch_schur_partial_roleb(w.g, w.stream); else launch_schur_partial(w.g, w.stream); }
    if (w.small_solve) { ProfScope p(w, solver == 2 ? VISFS_BA_K_PCG : VISFS_BA_K_DIRECT, true); launch_small_solve(w.g, solver, w.stream); }
    else {
        const bool fin_on_board = solver == 2 && w.g.fin_pcg && w.g.pcg1_code != nullptr && !w.g.pcg_cu;       // k_pcg1<FIN>: the finalisation rides on the PCG launch
        if (!fin_on_board && !w.g.fin_arrive) { ProfScope p(w, VISFS_BA_K_SCHUR_FINALIZE, true); launch_schur_finalize(w.g, w.stream); }     // (fin_arrive: done inside k_schur_partial)
        if (solver == 2) { ProfScope p(w, VISFS_BA_K_PCG, true); launch_pcg(w.g, w.stream); }
        else { ProfScope p(w, VISFS_BA_K_DIRECT); launch_direct(w.g, w.stream); }
    }
    if (w.g.dogleg) {
        // the solve above was the regularised Gauss-Newton step (H + mu M) dn = b.  Pass 1 keeps its landmark part and sums the inner
        // products of the dogleg construction, the one-workgroup kernel picks the point on the dogleg path, pass 2 evaluates it.
        { ProfScope p(w, VISFS_BA_K_BACKSUB, true); launch_backsub_dogleg(w.g, 1, w.stream); launch_dogleg_mid(w.g, w.stream); launch_backsub_dogleg(w.g, 2, w.stream); }
        { ProfScope p(w, VISFS_BA_K_DECIDE, true); launch_decide(w.g, w.stream); }
        return;
    }
    if (w.spec_fused) { ProfScope p(w, VISFS_BA_K_BACKSUB, true); launch_backsub_lin_decide(w.g, w.stream); return; }
    const bool fused_decide = w.fused_decide;     // the gated unit: the LM decision rides on k_backsub
    {   ProfScope p(w, VISFS_BA_K_BACKSUB, true);
        if (w.spec && (w.g.Ne > 0 || w.g.Nz > 0)) launch_backsub_odospec(w.g, w.stream);
        else if (!w.spec && fused_decide) launch_backsub_decide(w.g, w.stream);
        else launch_backsub(w.g, w.stream); }
    if (w.spec) { ProfScope p(w, VISFS_BA_K_LINEARIZE, true); launch_linearize_decide(w.g, w.stream); }
    else if (!fused_decide) { ProfScope p(w, VISFS_BA_K_DECIDE, true); launch_decide(w.g, w.stream); }
}

void fill_stats(const LmState& st, visfs_ba_stats* out) {
    std::memset(out, 0, sizeof(*out));
    out->status = st.status;
    out->iterations_run[0] = st.iterations_run[0]; out->iterations_run[1] = st.iterations_run[1];
    out->trials_run[0] = st.trials_run[0]; out->trials_run[1] = st.trials_run[1];
    out->pcg_iterations = st.pcg_total;
    out->n_outliers = st.n_outliers;
    out->chi2_initial = st.chi2_initial; out->chi2_phase1 = st.chi2_phase1; out->chi2_final = st.chi2_final;
    out->n_trace = st.n_trace;
    for (int i = 0; i < st.n_trace && i < MAX_TRACE; ++i) { out->trace_lambda[i] = st.trace_lambda[i]; out->trace_chi2[i] = st.trace_chi2[i]; }
    out->n_active_edges[0] = st.n_edges_ok; out->n_active_edges[1] = st.n_edges_ok - st.n_outliers;
    const int p1 = st.ended >= 1 ? st.pcg_phase1 : st.pcg_total;
    out->pcg_iterations_phase[0] = p1; out->pcg_iterations_phase[1] = st.pcg_total - p1;
}

// The persistent PCG (k_pcg / k_pcg1) only terminates when every workgroup of a window's grid is resident on the device.  Inside
// one stream launches are serialised; two handles (or two threads of the legacy one-stream-per-window mode) could otherwise have
// two partially resident grids starve each other until the bounded spins give up (VISFS_BA_ERR_DEVICE).  So every solve that
// carries a persistent PCG holds this per-device lock from its first launch to its last state read.  Other PROCESSES sharing the
// GPU are not covered: the library assumes exclusive use of the device for Optimizer/Solver=2 (include/visfs_ba.h).
//
// Round 2: a budget instead of a mutex.  Grids of the ONE-WAVE kernel (k_pcg1: 64-thread workgroups, no LDS, 248 VGPRs) are
// homogeneous: every SIMD of the device holds two such waves, so 1024 of them (counted conservatively: one per SIMD) are resident
// together wherever the dispatcher puts them — several handles / host threads may run such solves side by side as long as their
// block rows sum to <= 1024 (two 4-window batches of C2 size overlap: the hand-off waits of one hide behind the gathers of the
// other).  Grids of the four-wave kernel (k_pcg: LDS, one workgroup per CU) stay exclusive: a placement of one-wave grids that
// fills one SIMD of many CUs could otherwise keep a four-wave workgroup from ever fitting.
struct PcgBudget { std::mutex m; std::condition_variable cv; int used = 0; int cap = 0; };   // cap: wavefront slots of the device, 0 until first use
PcgBudget& pcg_device_budget(int device) {
    static PcgBudget b[64];
    return b[(unsigned)device % 64u];
}
// One-wave slots of the device, counted conservatively (one per SIMD): compute units x 4.  A partitioned device (CPX / NPS modes: e.g.
// 32 CUs) gets the budget of what it really has; if the query fails the budget is 1, i.e. every persistent PCG solve is exclusive.
static int pcg_device_slots(int device) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 1;
    return cus * 4;
}
class PcgLease {
public:
    PcgLease() = default;
    PcgLease(const PcgLease&) = delete;
    PcgLease& operator=(const PcgLease&) = delete;
    ~PcgLease() { release(); }
    void acquire(int device, int cost) {                       // cost <= 0 or > capacity: exclusive
        b_ = &pcg_device_budget(device);
        std::unique_lock<std::mutex> lk(b_->m);
        if (b_->cap == 0) b_->cap = pcg_device_slots(device);
        cost_ = (cost <= 0 || cost > b_->cap) ? b_->cap : cost;
        b_->cv.wait(lk, [&]() { return b_->used + cost_ <= b_->cap; });
        b_->used += cost_;
    }
    void release() {
        if (!b_) return;
        { std::lock_guard<std::mutex> lk(b_->m); b_->used -= cost_; }
        b_->cv.notify_all();
        b_ = nullptr;
    }
private:
    PcgBudget* b_ = nullptr;
    int cost_ = 0;
};
// the budget of a device (for decisions that depend on whether two leases fit side by side)
static int pcg_budget_capacity(int device) {
    PcgBudget& b = pcg_device_budget(device);
    std::lock_guard<std::mutex> lk(b.m);
    if (b.cap == 0) b.cap = pcg_device_slots(device);
    return b.cap;
}
// Wavefront slots a solve's persistent PCG grid needs (0: exclusive use of the budget).
static int pcg_wave_cost(const LaunchDims& d, int members) {
    static const int gv = []() { const char* e = std::getenv("VISFS_BA_PCG_GATHER"); return e ? std::atoi(e) : 1; }();
    if (!d.pcg_one_wave || gv == 3) return 0;                  // four-wave kernel (or the 8x grid of the XCD-local variant): exclusive
    return d.pcg_rows * std::max(1, members);
}

// Members of one batched launch sequence: rows x members workgroups of the persistent PCG must be resident together.
int batch_members_per_launch(visfs_ba_handle* h, const std::vector<Workspace*>& ws, const std::vector<int>& all) {
    LaunchDims d = dims_of(ws[all[0]]->g);
    bool no_pcg = true;
    for (int i : all) { d = dims_max(d, dims_of(ws[i]->g)); no_pcg = no_pcg && (ws[i]->small_solve || ws[i]->fused || ws[i]->g.pcg_cu || (h->prm.solver != 2 && ws[i]->g.band_B >= 0)); }
    if (no_pcg) return 4096;                               // single-workgroup solvers: no co-residency requirement
    int cap = pcg_resident_capacity(d, true, h->device);
    if (cap <= 0) cap = 256;                               // query failed: one 256-thread workgroup per CU is always admitted
    { const char* e = std::getenv("VISFS_BA_PCG_CAPACITY"); if (e && std::atoi(e) > 0) cap = std::atoi(e); }   // tests: force a split
    return std::max(1, cap / std::max(1, d.pcg_rows));
}

// Optimizer.cpp:261-318 on the resident graph.
// fresh_upload: the caller has just uploaded this window and nothing has touched it since (the window layer) — the upload's own k_reset has
// left exactly the state the reset below would produce.
// fallback_run: the re-run of a solve whose persistent PCG timed out, on the direct solver (w.solver_now = 0): the caller has restored the
// uploaded estimates and re-armed the LM state; no hipGraph (the captured sequence is the PCG one).
int ws_optimize_run(visfs_ba_handle* h, Workspace& w, visfs_ba_stats* stats, const bool fresh_upload, const bool fallback_run) {
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    HIP_TRY(h, hipSetDevice(h->device));          // a process may hold handles on several GPUs
    const int solver = w.solver_now >= 0 ? w.solver_now : h->prm.solver;
    // g2o branch: optimize(iterations / 2) twice (Optimizer.cpp:265,311); Ceres branch: one Solve with max_num_iterations = iterations (:521)
    const int half = w.g.ceres ? h->prm.iterations : h->prm.iterations / 2;
    PcgLease pcg_lease;                                                                         // persistent PCG: co-residency budget of the device
    if (solver == 2 && !w.small_solve && !w.fused && !w.g.pcg_cu) pcg_lease.acquire(h->device, pcg_wave_cost(dims_of(w.g), 1));
    // a fresh optimizer per call (Optimizer.cpp:75): all edges level 0, LM state re-armed, estimates kept
    const bool reset_launched = !fallback_run && !(fresh_upload && w.solves_since_upload == 0);
    if (reset_launched) { ProfScope p(w, VISFS_BA_K_RESET); launch_reset(w.g, half, h->prm.trust_region == 1, 0, w.stream); }
    if (w.fused) {
        // small window: both phases, the outlier pass and the final evaluation in one launch of one workgroup
        { ProfScope p(w, VISFS_BA_K_SMALL); launch_small_optimize(w.g, solver, half, w.stream); }
        HIP_TRY(h, hipGetLastError());
        int rcs = ws_read_state(h, w);
        if (rcs != VISFS_BA_OK) return rcs;
        if (w.h_state->status == VISFS_BA_ERR_DEVICE) { h->err = "fused LM loop did not terminate"; return VISFS_BA_ERR_DEVICE; }
        if (stats) fill_stats(*w.h_state, stats);
        if (w.prof_mask) { w.active[VISFS_BA_K_SMALL] += 1; if (reset_launched) w.active[VISFS_BA_K_RESET] += 1; }
        return w.h_state->status;
    }
    // Both phases, their ends and the outlier pass are enqueued BEFORE the host knows how the first phase went: the phase-end
    // kernels act only when the phase they close is done (LmState::done / ended), so the common solve — no rejected trial beyond
    // the units of its phase — costs ONE state read instead of one per phase (each is a stream drain plus the bubble until the
    // next launches arrive: ~25 us, 5 % of a production-size solve).  Whatever is left is driven from the state that comes back.
    const int half2 = (h->prm.robust_kernel_delta > 0.0 && !w.g.ceres) ? half : 0;                  // :310-311 (gated off on abort); no second pass in the Ceres branch
    // PER-FRAME REPLAY (round 4).  A window that has just been uploaded (visfs_ba_solve_window: every frame) runs its
    // launches through the kernels that read the graph from a FIXED device address (Many{ d_graph, d_state }, one window) on grids rounded
    // up to size classes: the kernel arguments then do not change from frame to frame, and the sequence captured for one frame's window is
    // replayed for the next frames' windows of the same class — the launch gaps of ~80 dependent eager launches leave the call.  Same
    // kernels' arithmetic as the by-value path: results are bit-identical (tests/test_window_map.py, tools/soak_frames.py).
    // MEASURED (profiles/r04_frame_replay_ab.log, both paths interleaved call by call in one process): NO gain — C2 1.258 -> 1.271 ms, PROD
    // 0.455-0.459 -> 0.455-0.477, C1 +3 %, C4 +7 % (its gated unit runs at 96 VGPRs in the fixed-address instantiation).  The stream of a
    // per-frame call is GPU-bound: the host enqueues its ~80 launches at 2.5 us each while a kernel takes 5-20 us, so the queue never runs
    // dry and a replayed sequence meets the same dispatch gaps between dependent kernels (rocprofv3: 1 070 us of kernel time in a 1.26 ms C2
    // call either way).  VERDICT r03 item 4's premise — r03's "0.98 ms replayed vs 1.06 eager" — compared a resident re-optimisation with a
    // per-frame call, not two launch modes of the same call.  The path stays available (VISFS_BA_FRAME_GRAPH=2; tests keep it honest); the
    // default is the by-value path.
    const int frame_mode = []() { const char* e = std::getenv("VISFS_BA_FRAME_GRAPH"); return e ? std::atoi(e) : 0; }();   // 0: by-value path, 1: fixed-address path without capture, 2: with (read per call: tests switch it)
    const bool band_direct = solver != 2 && !w.small_solve && w.g.band_B >= 0;
    const bool ref_mode = frame_mode != 0 && fresh_upload && w.solves_since_upload == 0 && !fallback_run && !w.prof_mask && !w.batch_member && w.d_graph && w.d_state &&
                          (h->prm.framework == 0 || w.small_solve || band_direct) && (solver == 2 || w.small_solve || band_direct) && !(w.spec && !w.spec_fused) &&
                          w.g.Np <= MAX_STAGED_POSES && w.g.Npf <= MAX_PCG_ONE_ROW_POSES && !w.g.pcg_cu;
    const LaunchDims fd = ref_mode ? dims_class(dims_of(w.g)) : LaunchDims{};
    auto enqueue_units = [&](int n, bool first) {
        for (int u = 0; u < n; ++u) {
            if (ref_mode) launch_unit_batch(w.d_graph, 1, fd, first, w.small_solve, solver, w.fused_decide, w.spec_fused, w.stream, w.d_state);
            else enqueue_unit(h, w, first);
            first = false;
        }
    };
    auto phase_end = [&](int which) {
        if (ref_mode) {
            if (which == 0) launch_phase_end_batch(w.d_graph, 1, fd, 0, 1, half2, w.stream, w.d_state);
            else launch_phase_end_batch(w.d_graph, 1, fd, 1, 0, 0, w.stream, w.d_state);
            return;
        }
        ProfScope p(w, VISFS_BA_K_PHASE_END);
        if (which == 0) launch_phase_end(w.g, 0, 1, half2, w.stream);                               // :270-303
        else launch_phase_end(w.g, 1, 0, 0, w.stream);                                              // :315-318
    };
    // (a window that rejected trials last time — an estimator's consecutive frames behave alike — gets that many units more up
    // front: a gated no-op unit costs ~7 us, the state read it saves ~25 us plus the bubble behind it)
    // hipGraph replay of the up-front launch sequence (VISFS_BA_GRAPH=0 disables, =1 captures at the first solve).  A graph holds the
    // kernel arguments (the DeviceGraph travels by value) and the grid sizes, so it serves only RE-optimisations of the same
    // resident graph with the same unit counts: by default it is captured when the same graph is optimised a second time, never
    // for a window that is uploaded, solved once and replaced (visfs_ba_solve_window per frame).  Measured
    // (profiles/r02_hipgraph_replay.log): C2 19.0 -> 19.4 k it/s (GPU-bound stream), PROD 23.5 -> 25.6 k it/s (small kernels:
    // the eager host launch rate shows).
    static const int graph_mode = []() { const char* e = std::getenv("VISFS_BA_GRAPH"); return e ? std::atoi(e) : 2; }();
    const int n0 = half + w.extra_units[0], n1 = half2 > 0 ? half2 + w.extra_units[1] : 0;
    w.solves_since_upload += 1;
    bool replayed = false;
    if (ref_mode && frame_mode >= 2 && graph_mode != 0 && !w.graph_failed) {
        // the sequence of this geometry class: replay it, or — the second time the class shows up — capture it
        const int flags = (w.small_solve ? 1 : 0) | (w.fused_decide ? 2 : 0) | (w.spec_fused ? 4 : 0) | (h->prm.trust_region == 1 ? 8 : 0);
        Workspace::FrameGraph* fg = nullptr;
        for (auto& q : w.frame_graphs)
            if (dims_equal(q.d, fd) && q.n0 == n0 && q.n1 == n1 && q.half == half && q.half2 == half2 && q.solver == solver && q.flags == flags) { fg = &q; break; }
        if (!fg) {
            if (w.frame_graphs.size() >= 8) {                 // least recently used class leaves
                size_t lru = 0;
                for (size_t q = 1; q < w.frame_graphs.size(); ++q) if (w.frame_graphs[q].last_use < w.frame_graphs[lru].last_use) lru = q;
                if (w.frame_graphs[lru].exec) (void)hipGraphExecDestroy(w.frame_graphs[lru].exec);
                w.frame_graphs.erase(w.frame_graphs.begin() + (long)lru);
            }
            w.frame_graphs.push_back(Workspace::FrameGraph{ fd, n0, n1, half, half2, solver, flags, nullptr, 0, 0 });
            fg = &w.frame_graphs.back();
        }
        fg->last_use = ++w.frame_clock;
        const int capture_at = []() { const char* e = std::getenv("VISFS_BA_FRAME_GRAPH_AT"); return e ? std::max(0, std::atoi(e)) : 1; }();
        if (!fg->exec && fg->seen >= capture_at) {
            // (failure path as for the resident-graph capture below: the capture is always ended, a partial graph dropped, eager launches
            // from then on for this workspace)
            hipGraph_t gr = nullptr;
            bool ok = hipStreamBeginCapture(w.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                enqueue_units(n0, true); phase_end(0); enqueue_units(n1, true); phase_end(1);
                const hipError_t launch_err = hipGetLastError();
                const hipError_t end_err = hipStreamEndCapture(w.stream, &gr);
                ok = launch_err == hipSuccess && end_err == hipSuccess && gr != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&fg->exec, gr, nullptr, nullptr, 0) == hipSuccess;
            if (gr) (void)hipGraphDestroy(gr);
            if (!ok) { if (fg->exec) { (void)hipGraphExecDestroy(fg->exec); fg->exec = nullptr; } w.graph_failed = true; (void)hipGetLastError(); }
        }
        fg->seen += 1;
        if (fg->exec) {
            if (hipGraphLaunch(fg->exec, w.stream) == hipSuccess) replayed = true;
            else { (void)hipGraphExecDestroy(fg->exec); fg->exec = nullptr; w.graph_failed = true; (void)hipGetLastError(); }
        }
    }
    else if (graph_mode != 0 && !ref_mode && !fallback_run && !w.prof_mask && !w.graph_failed && (graph_mode == 1 || w.solves_since_upload >= 2)) {
        if (!w.graph_exec || w.graph_units[0] != n0 || w.graph_units[1] != n1) {
            if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; }
            // Any failure between begin and end must still END the capture (the stream is unusable otherwise), drop the partial graph and
            // fall back to the eager sequence for this solve; capture is then not retried for this handle's workspace.
            hipGraph_t gr = nullptr;
            bool ok = hipStreamBeginCapture(w.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                enqueue_units(n0, true); phase_end(0); enqueue_units(n1, true); phase_end(1);
                const hipError_t launch_err = hipGetLastError();
                const hipError_t end_err = hipStreamEndCapture(w.stream, &gr);
                ok = launch_err == hipSuccess && end_err == hipSuccess && gr != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&w.graph_exec, gr, nullptr, nullptr, 0) == hipSuccess;
            if (gr) (void)hipGraphDestroy(gr);
            if (!ok) { if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; } w.graph_failed = true; (void)hipGetLastError(); }
            else { w.graph_units[0] = n0; w.graph_units[1] = n1; }
        }
        if (w.graph_exec) {
            if (hipGraphLaunch(w.graph_exec, w.stream) == hipSuccess) replayed = true;
            else { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; w.graph_failed = true; (void)hipGetLastError(); }
        }
    }
    w.last_replayed = replayed;
    if (!replayed) {
        enqueue_units(n0, true);                                                                    // :265
        phase_end(0);
        enqueue_units(n1, true);
        phase_end(1);
    }
    int rc = VISFS_BA_OK;
    for (int guard = 0;; ++guard) {
        HIP_TRY(h, hipGetLastError());
        rc = ws_read_state(h, w);
        if (rc != VISFS_BA_OK) return rc;
        const LmState& st = *w.h_state;
        if (st.status == VISFS_BA_ERR_DEVICE) { h->err = "a device-side hand-off (persistent PCG / LM decision) timed out"; return VISFS_BA_ERR_DEVICE; }
        if (st.ended >= 2 || st.status != 0) break;                                                // finished, or aborted by a chi2 guard
        if (guard > 16 * h->prm.iterations + 32) { h->err = "LM state machine did not terminate"; return VISFS_BA_ERR_DEVICE; }
        if (!st.done) {
            // rejected trials consumed units without finishing an iteration: top the phase up, then what follows it
            enqueue_units(std::max(1, st.max_iter - st.phase_iter), false);
            if (st.ended == 0) { phase_end(0); enqueue_units(half2, true); }
            phase_end(1);
        } else if (st.ended == 0) { phase_end(0); enqueue_units(half2, true); phase_end(1); }
        else phase_end(1);
    }
    for (int ph = 0; ph < 2; ++ph) w.extra_units[ph] = std::min(10, std::max(0, w.h_state->trials_run[ph] - w.h_state->iterations_run[ph]));
    if (stats) fill_stats(*w.h_state, stats);
    if (w.prof_mask) {
        const LmState& st = *w.h_state;
        w.active[VISFS_BA_K_LINEARIZE] += w.spec_fused ? (st.iterations_run[0] > 0) + (st.iterations_run[1] > 0)
                                          : w.spec ? st.n_active[3] + (st.iterations_run[0] > 0) + (st.iterations_run[1] > 0) : st.n_active[0];
        w.active[VISFS_BA_K_LIN_FINALIZE] += (st.iterations_run[0] > 0) + (st.iterations_run[1] > 0);
        w.active[VISFS_BA_K_SCHUR] += st.n_active[1]; w.active[VISFS_BA_K_SCHUR_FINALIZE] += st.n_active[1];
        if (!w.spec && !w.fused_decide) w.active[VISFS_BA_K_DECIDE] += st.n_active[1];
        w.active[solver == 2 ? VISFS_BA_K_PCG : VISFS_BA_K_DIRECT] += st.n_active[1];
        w.active[VISFS_BA_K_BACKSUB] += st.n_active[3];
        w.active[VISFS_BA_K_PHASE_END] += 2; if (reset_launched) w.active[VISFS_BA_K_RESET] += 1;
    }
    return w.h_state->status;
}

// A solve that cannot be lost to residency.  The persistent PCG needs every workgroup of its grid resident at once;
// when another process keeps part of the GPU busy its hand-off waits give up (LmState::pcg_timeout) and the state machine stops with
// VISFS_BA_ERR_DEVICE.  The reference's linear solver cannot fail that way (Optimizer.cpp:76-91), so the solve is re-run HERE, in the same
// call, from the estimates it started from, on the non-persistent direct solver (k_band_chol, or the dense blocked Cholesky for a wide band)
// — an exact solve of the same damped systems where PCG stops at its tolerance: the result is a valid localOptimize result, not bit-equal to
// the PCG one.  Only a solve that started from the uploaded estimates can be re-run (the window layer always does; the GRAPH layer after an
// upload or visfs_ba_graph_reset); visfs_ba_stats::solver_fallback / visfs_ba_result::solver_fallback report it.
int ws_optimize(visfs_ba_handle* h, Workspace& w, visfs_ba_stats* stats, const bool fresh_upload = false) {
    const bool pristine = w.pristine;
    w.pristine = false;
    w.solver_now = -1;
    int rc = ws_optimize_run(h, w, stats, fresh_upload, false);
    if (stats) stats->solver_fallback = 0;
    if (rc != VISFS_BA_ERR_DEVICE || !w.loaded || !w.h_state || !w.h_state->pcg_timeout) return rc;
    if (!(pristine && w.direct_ready && h->prm.solver == 2)) return rc;
    w.solver_now = 0;
    w.fallbacks += 1;
    // the uploaded estimates back, all edges level 0, LM state re-armed (k_reset also clears pcg_timeout)
    launch_reset(w.g, w.g.ceres ? h->prm.iterations : h->prm.iterations / 2, h->prm.trust_region == 1, 1, w.stream);
    if (hipGetLastError() != hipSuccess) { w.solver_now = -1; return rc; }
    rc = ws_optimize_run(h, w, stats, false, true);
    w.solver_now = -1;
    if (stats) stats->solver_fallback = 1;
    if (rc != VISFS_BA_ERR_DEVICE) h->err.clear();
    return rc;
}

// state_fresh: w.h_state already holds the LM state of the finished run (ws_optimize / batch_optimize read it): no extra round trip.
int ws_download(visfs_ba_handle* h, Workspace& w, double* pose_tq, double* point_xyz, uint8_t* obs_outlier, double* obs_chi2, bool state_fresh = false) {
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    HIP_TRY(h, hipSetDevice(h->device));
    if (!state_fresh) { int rc = ws_read_state(h, w); if (rc != VISFS_BA_OK) return rc; }
    const int sel = w.h_state->sel;
    const DeviceGraph& g = w.g;
    if (state_fresh && w.outputs_staged && !obs_chi2) {
        // the outputs came back with the state (ws_read_state): no copy, no synchronisation
        const OutputStage os = output_stage_of(g);
        const double* ps = reinterpret_cast<const double*>(w.h_base + os.pose[sel]);
        if (pose_tq) for (int i = 0; i < g.Np; ++i) for (int q = 0; q < 7; ++q) pose_tq[7 * i + q] = ps[POSE_STRIDE * i + q];
        if (point_xyz && g.Nl) std::memcpy(point_xyz, w.h_base + os.pt[sel], (size_t)g.Nl * 24);
        if (obs_outlier && g.No) std::memcpy(obs_outlier, w.h_base + os.out, (size_t)g.No);
        w.outputs_staged = false;
        return VISFS_BA_OK;
    }
    // every copy is enqueued first, ONE synchronisation at the end.  The copies land in the workspace's PINNED staging arena (idle
    // between uploads) and are moved to the caller's pageable buffers by the host: a device-to-pageable copy is staged and
    // serialised by the runtime chunk by chunk.
    const size_t b_pose = pose_tq ? (size_t)g.Np * POSE_STRIDE * 8 : 0, b_pt = (point_xyz && g.Nl) ? (size_t)g.Nl * 24 : 0;
    const size_t b_out = (obs_outlier && g.No) ? (size_t)g.No : 0, b_chi = (obs_chi2 && g.No) ? (size_t)g.No * 8 : 0;
    const size_t o_pose = 0, o_pt = (o_pose + b_pose + 255) & ~size_t(255), o_chi = (o_pt + b_pt + 255) & ~size_t(255), o_out = (o_chi + b_chi + 255) & ~size_t(255);
    const bool staged = w.h_base && o_out + b_out <= w.h_cap;
    std::vector<double> tmp;
    char* hb = w.h_base;
    if (!staged && pose_tq) tmp.resize((size_t)g.Np * POSE_STRIDE);
    if (pose_tq) HIP_TRY(h, hipMemcpyAsync(staged ? (void*)(hb + o_pose) : (void*)tmp.data(), g.pose[sel], b_pose, hipMemcpyDeviceToHost, w.stream));
    if (b_pt) HIP_TRY(h, hipMemcpyAsync(staged ? (void*)(hb + o_pt) : (void*)point_xyz, g.pt[sel], b_pt, hipMemcpyDeviceToHost, w.stream));
    if (b_out) HIP_TRY(h, hipMemcpyAsync(staged ? (void*)(hb + o_out) : (void*)obs_outlier, g.obs_outlier, b_out, hipMemcpyDeviceToHost, w.stream));
    if (b_chi) HIP_TRY(h, hipMemcpyAsync(staged ? (void*)(hb + o_chi) : (void*)obs_chi2, g.obs_chi2_out, b_chi, hipMemcpyDeviceToHost, w.stream));
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    const double* ps = staged ? reinterpret_cast<const double*>(hb + o_pose) : tmp.data();
    if (pose_tq) for (int i = 0; i < g.Np; ++i) for (int q = 0; q < 7; ++q) pose_tq[7 * i + q] = ps[POSE_STRIDE * i + q];
    if (staged) {
        if (b_pt) std::memcpy(point_xyz, hb + o_pt, b_pt);
        if (b_out) std::memcpy(obs_outlier, hb + o_out, b_out);
        if (b_chi) std::memcpy(obs_chi2, hb + o_chi, b_chi);
    }
    return VISFS_BA_OK;
}

// ------------------------------------------------------------------ window layer
int find_id(const uint64_t* ids, int n, uint64_t id) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (ids[mid] == id) return mid;
        if (ids[mid] < id) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// Graph build of localOptimize (Optimizer.cpp:100-223) into caller-chosen arrays; `pose` rows are pose_stride doubles apart (7 for the
// public visfs_ba_pack_window, POSE_STRIDE when the window layer packs straight into the device layout).  With a pool the
// references — the only O(N_obs) part — are shared out at feature boundaries (the loop streams 28 B in and 36 B out per reference and
// is bound by memory bandwidth: measured 208 -> 145-170 us at C2 with four threads, no gain at C4 — profiles/r03_host_threads.log); every thread writes its observations where they
// would land if no reference were skipped, and the (rare) gaps are closed afterwards, so the result is the serial one byte for byte.
struct PackOut {
    double* pose = nullptr; int pose_stride = 7; uint8_t* pose_fixed = nullptr; uint8_t* point_used = nullptr;
    int32_t* obs_point = nullptr; int32_t* obs_pose = nullptr; double* obs_uvr = nullptr; int32_t* obs_ref = nullptr;
    float* obs_uvd = nullptr;          // when set: the raw (u, v, depth) floats are stored instead of obs_uvr (the device forms the measurement)
    int32_t* odo_from = nullptr; int32_t* odo_to = nullptr; double* odo_tq = nullptr;
};
struct PackSummary { std::vector<SummaryPart>* part = nullptr; std::vector<int32_t> pose_free; int Npf = 0; int slots = 0; };
int pack_window_impl(const visfs_ba_window* w, const PackOut& o, visfs_ba_graph* g, int32_t* n_mono_skipped, WorkerPool* pool, PackSummary* ps = nullptr) {
    std::memset(g, 0, sizeof(*g));
    const auto te0 = std::chrono::steady_clock::now();
    // std::map order (the direct pose table below and the hinted searches rely on it; visfs_ba_pack_window is public)
    for (int i = 1; i < w->n_poses; ++i) if (w->pose_ids[i] <= w->pose_ids[i - 1]) return VISFS_BA_ERR_BAD_ARGUMENT;
    for (int i = 1; i < w->n_points; ++i) if (w->point_ids[i] <= w->point_ids[i - 1]) return VISFS_BA_ERR_BAD_ARGUMENT;
    // poses: Twc = Twr * Trc ; Tcw = Twc^-1 as CameraPose(R,t) ; fixed iff id == rootId   (Optimizer.cpp:100-114)
    for (int i = 0; i < w->n_poses; ++i) {
        double Twc[12], Tcw[12];
        iso_mul(w->pose_Twr + 12 * i, w->Trc, Twc);
        iso_inv(Twc, Tcw);
        iso_to_tq(Tcw, o.pose + (size_t)o.pose_stride * i);
        for (int q = 7; q < o.pose_stride; ++q) o.pose[(size_t)o.pose_stride * i + q] = 0.0;
        o.pose_fixed[i] = (w->pose_ids[i] == w->root_id);
    }
    // links: T_c1c2 = Trc^-1 * T_r1r2 * Trc as SE3Quat   (Optimizer.cpp:123-150)
    int ne = 0;
    double Tcr[12];
    iso_inv(w->Trc, Tcr);
    for (int k = 0; k < w->n_links; ++k) {
        const uint64_t from = w->link_from[k], to = w->link_to[k];
        if (from == 0 || to == 0) continue;
        const int a = find_id(w->pose_ids, w->n_poses, from), b = find_id(w->pose_ids, w->n_poses, to);
        if (a < 0 || b < 0 || from == to) continue;
        double T1[12], T2[12];
        iso_mul(Tcr, w->link_T + 12 * k, T1);
        iso_mul(T1, w->Trc, T2);
        iso_to_tq(T2, o.odo_tq + 7 * ne);
        o.odo_from[ne] = a; o.odo_to[ne] = b;
        ++ne;
    }
    // landmarks + stereo edges   (Optimizer.cpp:153-223)
    std::memset(o.point_used, 0, (size_t)w->n_points);
    const int Nr = w->n_refs;
    if (ps) {
        // the structure summary of the upload (observations per free pose, co-observation pairs per block of S) is accumulated here,
        // while an observation's indices are in registers: re-reading them from the pinned arena afterwards is a pass over uncached memory
        ps->pose_free.resize(w->n_poses); ps->Npf = 0;
        for (int i = 0; i < w->n_poses; ++i) ps->pose_free[i] = o.pose_fixed[i] ? -1 : ps->Npf++;
        ps->slots = pool ? pool->size() : 1;
        if ((size_t)ps->Npf * ps->Npf > ((size_t)1 << 17) && ps->slots > 1) { ps->slots = 0; ps = nullptr; }     // very wide systems: per-thread tables would not pay
        else { if ((int)ps->part->size() < ps->slots) ps->part->resize(ps->slots); for (int t = 0; t < ps->slots; ++t) (*ps->part)[t].used = false; }
    }
    double baseLine = 0.0;
    if (w->n_cameras > 1) baseLine = (double)w->baseline;                       // :181-183
    const int T = (pool && Nr >= 16384) ? 4 * pool->size() : 1;       // several tasks per thread: a worker that wakes late still finds work
    std::vector<int> cut(T + 1, Nr);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        int k = (int)((int64_t)Nr * t / T);
        k = std::max(k, cut[t - 1]);
        while (k > 0 && k < Nr && w->ref_feature[k] == w->ref_feature[k - 1]) ++k;   // a feature's references (and its point_used flag) stay with one thread
        cut[t] = k;
    }
    // signature ids of a sliding window are a short ascending range: the pose of a reference is found through a direct table
    // (id - first id) instead of a search per reference; windows whose ids are spread wider than 64 k keep the search
    const uint64_t pose_base = w->n_poses > 0 ? w->pose_ids[0] : 0;
    const uint64_t pose_span = w->n_poses > 0 ? w->pose_ids[w->n_poses - 1] - pose_base : 0;
    std::vector<int16_t> pose_tab;
    if (w->n_poses > 0 && w->n_poses < 32768 && pose_span < 65536) {
        pose_tab.assign((size_t)pose_span + 1, (int16_t)-1);
        for (int i = 0; i < w->n_poses; ++i) pose_tab[(size_t)(w->pose_ids[i] - pose_base)] = (int16_t)i;
    }
    const int16_t* const ptab = pose_tab.empty() ? nullptr : pose_tab.data();
    struct Part { int no = 0, mono = 0, first_p = -1, first_c = -1, last_p = -1, last_c = -1; bool bad = false; };
    std::vector<Part> part(T);
    static const bool timing = std::getenv("VISFS_BA_TIMING") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "   pack: before the reference loop %.1f us\n", std::chrono::duration<double, std::micro>(tr0 - te0).count());
    std::atomic<int> by_caller{ 0 };
    // (diagnostic, timing runs only — results are WRONG with it: 2 = no output stores, 4 = no point_used store, 8 = no summary accumulation)
    static const int dbg = []() { const char* e = std::getenv("VISFS_BA_PACK_DEBUG"); return (e && std::getenv("VISFS_BA_TIMING")) ? std::atoi(e) : 0; }();
    struct TaskLap { int slot; double t0, t1; };
    std::vector<TaskLap> task_lap(timing ? T : 0);
    auto body = [&](int t, int slot) {
        if (timing && slot == 0) by_caller.fetch_add(1, std::memory_order_relaxed);
        struct LapGuard { TaskLap* l; std::chrono::steady_clock::time_point z; ~LapGuard() { if (l) l->t1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - z).count(); } } lap_guard{ nullptr, tr0 };
        if (timing) { task_lap[t].slot = slot; task_lap[t].t0 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); lap_guard.l = &task_lap[t]; }
        SummaryPart* SP = ps ? &(*ps->part)[slot] : nullptr;
        if (SP && !SP->used) SP->begin(ps->Npf, w->n_points);
        Part& P = part[t];
        int no = cut[t], mono = 0, last_p = -1, last_c = -1;
        // references arrive in nested-map order, so the id looked up is almost always at (or right after) the previous hit
        int hint_p = 0, hint_c = 0;
        auto find_hinted = [](const uint64_t* ids, int n, uint64_t id, int& hint) {
            if (hint < n && ids[hint] == id) return hint;
            if (hint + 1 < n && ids[hint + 1] == id) return ++hint;
            if (hint < n && ids[hint] < id) {                                      // a few ids without references in between: walk, do not search
                const int stop = std::min(n, hint + 10);
                for (int q = hint + 2; q < stop; ++q) { if (ids[q] == id) { hint = q; return q; } if (ids[q] > id) break; }
            }
            const int f = find_id(ids, n, id);
            if (f >= 0) hint = f;
            return f;
        };
        // one feature at a time (its references are consecutive: nested-map order): the feature is looked up, flagged and handed to the
        // structure summary once, the loop over its references only resolves the pose, tests the depth and stores
        const bool fast_sum = SP && !(dbg & 8);
        for (int k = cut[t]; k < cut[t + 1];) {
            const uint64_t fid = w->ref_feature[k];
            int k1 = k + 1;
            while (k1 < cut[t + 1] && w->ref_feature[k1] == fid) ++k1;
            const int p = find_hinted(w->point_ids, w->n_points, fid, hint_p);
            if (p < 0) { k = k1; continue; }                                        // :158
            if (!(dbg & 4)) o.point_used[p] = 1;
            const bool pfixed = w->point_fixed[p] != 0;
            const int no0 = no;
            for (; k < k1; ++k) {
                const uint64_t pid = w->ref_pose[k];
                int c;
                if (ptab) { const uint64_t off = pid - pose_base; c = off <= pose_span ? ptab[off] : -1; }
                else c = find_hinted(w->pose_ids, w->n_poses, pid, hint_c);
                if (c < 0 || pid == 0) continue;                                    // :172
                const float depth = w->ref_depth[k];                                // :174
                if (!(std::isfinite(depth) && depth > 0.0f && baseLine > 0.0)) { ++mono; continue; }   // the reference dereferences an uninitialised edge pointer here (:179, :197-210); we skip the observation
                if (p < last_p || (p == last_p && c <= last_c)) { P.bad = true; return; }   // nested std::map order
                if (P.first_p < 0) { P.first_p = p; P.first_c = c; }
                last_p = p; last_c = c;
                if (dbg & 2) { ++no; continue; }
                if (o.obs_uvd) {
                    o.obs_uvd[3 * (size_t)no + 0] = w->ref_u[k]; o.obs_uvd[3 * (size_t)no + 1] = w->ref_v[k]; o.obs_uvd[3 * (size_t)no + 2] = depth;
                } else {
                    const float disparity = static_cast<float>(baseLine * w->fx / (double)depth);   // :187
                    o.obs_uvr[3 * (size_t)no + 0] = (double)w->ref_u[k];
                    o.obs_uvr[3 * (size_t)no + 1] = (double)w->ref_v[k];
                    o.obs_uvr[3 * (size_t)no + 2] = (double)(w->ref_u[k] - disparity);              // float - float, :188
                }
                o.obs_point[no] = p; o.obs_pose[no] = c;
                if (o.obs_ref) o.obs_ref[no] = k;
                ++no;
            }
            if (fast_sum && !(dbg & 2)) for (int q = no0; q < no; ++q) SP->add(p, pfixed, ps->pose_free[o.obs_pose[q]], o.obs_pose[q]);
        }
        if (SP) SP->flush();
        P.no = no - cut[t]; P.mono = mono; P.last_p = last_p; P.last_c = last_c;
    };
    if (T > 1) pool->run(T, body); else body(0, 0);
    if (timing) {
        std::fprintf(stderr, "   pack: poses + links + cuts, then %d tasks (%d by the caller): %.1f us since entry\n", T, by_caller.load(),
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count());
        if (T > 1 && std::getenv("VISFS_BA_TIMING_TASKS")) for (int t = 0; t < T; ++t) std::fprintf(stderr, "      task %2d on slot %d: %7.1f -> %7.1f us (%d references)\n", t, task_lap[t].slot, task_lap[t].t0, task_lap[t].t1, cut[t + 1] - cut[t]);
    }
    int no = 0, mono = 0, last_p = -1, last_c = -1;
    for (int t = 0; t < T; ++t) {
        const Part& P = part[t];
        if (P.bad) return VISFS_BA_ERR_BAD_ARGUMENT;
        if (P.no > 0) {
            if (P.first_p < last_p || (P.first_p == last_p && P.first_c <= last_c)) return VISFS_BA_ERR_BAD_ARGUMENT;
            last_p = P.last_p; last_c = P.last_c;
            if (no != cut[t]) {                                                     // close the gap skipped references left
                std::memmove(o.obs_point + no, o.obs_point + cut[t], (size_t)P.no * 4);
                std::memmove(o.obs_pose + no, o.obs_pose + cut[t], (size_t)P.no * 4);
                if (o.obs_uvd) std::memmove(o.obs_uvd + 3 * (size_t)no, o.obs_uvd + 3 * (size_t)cut[t], (size_t)P.no * 12);
                else std::memmove(o.obs_uvr + 3 * (size_t)no, o.obs_uvr + 3 * (size_t)cut[t], (size_t)P.no * 24);
                if (o.obs_ref) std::memmove(o.obs_ref + no, o.obs_ref + cut[t], (size_t)P.no * 4);
            }
        }
        no += P.no; mono += P.mono;
    }
    if (n_mono_skipped) *n_mono_skipped = mono;
    g->n_poses = w->n_poses; g->n_points = w->n_points; g->n_obs = no; g->n_odo = ne;
    g->pose_tq = o.pose; g->pose_fixed = o.pose_fixed;
    g->point_xyz = w->point_xyz; g->point_fixed = w->point_fixed;
    g->obs_point = o.obs_point; g->obs_pose = o.obs_pose; g->obs_uvr = o.obs_uvr;
    g->odo_from = o.odo_from; g->odo_to = o.odo_to; g->odo_tq = o.odo_tq;
    g->fx = w->fx; g->fy = w->fy; g->cx = w->cx; g->cy = w->cy;
    g->bf = ((w->n_cameras > 1) ? (double)w->baseline : 0.0) * w->fx;          // :195
    // range points (Optimizer.cpp:225-258): `!_pointClouds.empty() && _submap != nullptr`; every point hangs off the newest pose
    if (w->n_laser_points > 0 && w->grid != nullptr && w->laser_xyz != nullptr) {
        g->n_laser = w->n_laser_points; g->laser_pose = w->n_poses - 1;
        g->laser_xyz = w->laser_xyz; g->grid = w->grid;
    }
    std::memcpy(g->Tcr, Tcr, 96);                                             // transformRobotToImage_ (TypeOccupiedSpace2D.h:81-82)
    return VISFS_BA_OK;
}

// Optimizer/Framework (Parameters.h:184): 0 = the g2o branch, 1 = the Ceres branch — LEVENBERG_MARQUARDT, or with Optimizer/TrustRegion=1
// the (traditional) DOGLEG strategy (Optimizer.cpp:515-519).
static const char* framework_refusal(const visfs_ba_params& prm) {
    if (prm.framework == 0 || prm.framework == 1) return nullptr;
    return "Optimizer/Framework must be 0 (g2o branch) or 1 (Ceres branch)";
}

// localOptimize in three steps so that a batch can run the middle one for many windows at once.
// prepare_window: guards of Optimizer.cpp:74 / :360-364, graph build (:100-223) straight into the pinned staging arena of the device
// layout, upload.  Returns 1 when the window is resident and has to be optimised, 0 when `r` is already final (pass-through or refused
// input).  pool: host threads for the O(N_obs) passes (nullptr: the calling thread alone).
int prepare_window(visfs_ba_handle* h, Workspace& w, const visfs_ba_window* win, visfs_ba_result* r, PackedWindow& pk, WorkerPool* pool = nullptr) {
    const visfs_ba_params& prm = h->prm;
    r->n_poses_out = 0; r->n_outliers = 0; r->warn_mono_skipped = 0; r->solver_fallback = 0;
    r->iterations_run[0] = r->iterations_run[1] = 0;
    r->chi2_initial = r->chi2_phase1 = r->chi2_final = 0.0;
    if (const char* why = framework_refusal(prm)) { h->err = why; r->status = VISFS_BA_ERR_UNSUPPORTED; return 0; }
    if (win->n_laser_points < 0 || (win->n_laser_points > 0 && win->grid && !win->laser_xyz)) { r->status = bad(h, "laser points without coordinates"); return 0; }
    if (win->n_poses < 0 || win->n_points < 0 || win->n_refs < 0 || win->n_links < 0) { r->status = bad(h, "negative sizes"); return 0; }
    // guards of Optimizer.cpp:74 and :360-364
    if (!(win->n_poses >= 2 && prm.iterations > 0 && win->pose_ids[0] > 0)) {
        if (win->n_poses == 1 || prm.iterations <= 0) {
            for (int i = 0; i < win->n_poses; ++i) { r->pose_ids_out[i] = win->pose_ids[i]; std::memcpy(r->pose_Twr_out + 12 * i, win->pose_Twr + 12 * i, 96); }
            r->n_poses_out = win->n_poses;
            r->status = VISFS_BA_PASSTHROUGH;
            return 0;
        }
        r->status = VISFS_BA_ERR_TOO_FEW_POSES;
        return 0;
    }
    for (int i = 1; i < win->n_poses; ++i) if (win->pose_ids[i] <= win->pose_ids[i - 1]) { r->status = bad(h, "pose ids must ascend (std::map order)"); return 0; }
    for (int i = 1; i < win->n_points; ++i) if (win->point_ids[i] <= win->point_ids[i - 1]) { r->status = bad(h, "point ids must ascend (std::map order)"); return 0; }
    const int Np = win->n_poses, Nl = win->n_points, Nr = win->n_refs, Nk = win->n_links;
    int rc = ws_init(h, w);
    if (rc != VISFS_BA_OK) { r->status = rc; return 0; }
    // the primary staging arena is the source of the previous upload's host-to-device copy: that copy must have left it
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(w.stream) != hipSuccess) { h->err = "hipStreamSynchronize failed"; r->status = VISFS_BA_ERR_DEVICE; return 0; }
    UploadOpts opt;
    opt.in_staging = true; opt.trusted = true; opt.pool = pool;
    opt.cap.Np = Np; opt.cap.Nl = Nl; opt.cap.cap_obs = Nr; opt.cap.cap_odo = Nk;
    // the references cross the memory bus and PCIe as localOptimize receives them (float u, v, depth: 12 bytes); the device forms (u_l, v_l, u_r)
    { static const bool raw = []() { const char* e = std::getenv("VISFS_BA_RAW_REFS"); return !(e && e[0] == '0'); }(); opt.cap.raw_refs = raw; }
    opt.baseline = (win->n_cameras > 1) ? (double)win->baseline : 0.0;
    if (win->n_laser_points > 0 && win->grid != nullptr && win->laser_xyz != nullptr) {
        const visfs_ba_grid& G = *win->grid;
        opt.cap.Nz = win->n_laser_points;
        if (G.num_x_cells > 0 && G.num_y_cells > 0 && (int64_t)G.num_x_cells * G.num_y_cells <= (int64_t)1 << 28) opt.cap.grid_cells = (size_t)G.num_x_cells * G.num_y_cells;
    }
    rc = ensure_prim(h, w, prim_bytes(opt.cap));
    if (rc != VISFS_BA_OK) { r->status = rc; return 0; }
    DeviceGraph hg{};
    { Arena hp{ w.h_prim, w.h_prim_cap, 0 }; layout_prim(hp, hg, opt.cap); }
    pk.pose_fixed.resize(Np); pk.point_used.resize(std::max(Nl, 1)); pk.obs_ref.resize(std::max(Nr, 1));
    PackOut o;
    o.pose = const_cast<double*>(hg.pose0); o.pose_stride = POSE_STRIDE; o.pose_fixed = pk.pose_fixed.data(); o.point_used = pk.point_used.data();
    o.obs_point = const_cast<int32_t*>(hg.obs_pt); o.obs_pose = const_cast<int32_t*>(hg.obs_pose); o.obs_uvr = const_cast<double*>(hg.obs_uvr); o.obs_uvd = const_cast<float*>(hg.obs_uvd); o.obs_ref = pk.obs_ref.data();
    o.odo_from = const_cast<int32_t*>(hg.odo_i); o.odo_to = const_cast<int32_t*>(hg.odo_j); o.odo_tq = const_cast<double*>(hg.odo_tq);
    const bool timing = std::getenv("VISFS_BA_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    PackSummary psum; psum.part = &w.sum_part;
    rc = pack_window_impl(win, o, &pk.g, &pk.mono, pool, &psum);
    opt.summary_slots = psum.slots;     // (0 when the build declined to accumulate: ws_upload then runs its own pass)
    if (timing) std::fprintf(stderr, "  pack (%d thread%s)     %8.1f us\n", pool ? pool->size() : 1, pool ? "s" : "", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp0).count());
    if (rc != VISFS_BA_OK) { r->status = bad(h, "window references must be sorted by (feature, pose)"); return 0; }
    if (Nl) { std::memcpy(const_cast<double*>(hg.pt0), win->point_xyz, (size_t)Nl * 24); std::memcpy(const_cast<uint8_t*>(hg.pt_fixed), win->point_fixed, Nl); }
    r->warn_mono_skipped = pk.mono;
    rc = ws_upload(h, w, &pk.g, opt);
    if (rc != VISFS_BA_OK) { r->status = rc; return 0; }
    return 1;
}

// finish_window: write-back (Optimizer.cpp:284-302 outliers, :320-358 poses and landmarks) from the optimised resident window.
int finish_window(visfs_ba_handle* h, Workspace& w, const visfs_ba_window* win, visfs_ba_result* r, const PackedWindow& pk, int rc, const visfs_ba_stats& st, WorkerPool* pool = nullptr) {
    if (rc == VISFS_BA_ERR_DEVICE || rc == VISFS_BA_ERR_NOT_LOADED) return r->status = rc;
    const int Np = win->n_poses, Nl = win->n_points;
    r->status = rc;
    r->solver_fallback = st.solver_fallback;
    r->iterations_run[0] = st.iterations_run[0]; r->iterations_run[1] = st.iterations_run[1];
    r->chi2_initial = st.chi2_initial; r->chi2_phase1 = st.chi2_phase1; r->chi2_final = st.chi2_final;
    if (rc != VISFS_BA_OK && rc != VISFS_BA_ERR_HUGE_CHI2_2) return rc;
    // the outputs came back with the state (ws_read_state): read them where they are; otherwise fetch them now
    std::vector<double> pose_v, pts_v;
    std::vector<uint8_t> outl_v;
    const double* pose = nullptr; const double* pts = nullptr; const uint8_t* outl = nullptr;
    int pose_stride = 7;
    if (w.outputs_staged) {
        const OutputStage os = output_stage_of(w.g);
        const int sel = w.h_state->sel;
        pose = reinterpret_cast<const double*>(w.h_base + os.pose[sel]); pose_stride = POSE_STRIDE;
        pts = reinterpret_cast<const double*>(w.h_base + os.pt[sel]);
        outl = reinterpret_cast<const uint8_t*>(w.h_base + os.out);
        w.outputs_staged = false;
    } else {
        pose_v.resize((size_t)Np * 7); pts_v.resize((size_t)std::max(Nl, 1) * 3); outl_v.resize(std::max(pk.g.n_obs, 1));
        int rc2 = ws_download(h, w, pose_v.data(), pts_v.data(), outl_v.data(), nullptr, /*state_fresh=*/true);
        if (rc2 != VISFS_BA_OK) return r->status = rc2;
        pose = pose_v.data(); pts = pts_v.data(); outl = outl_v.data();
    }
    // outliers are appended at Optimizer.cpp:296, before the phase-2 abort check.  Large windows: counted and written by ranges of
    // observations on the pool (a range's outliers land where the serial loop would put them).
    static const bool timing = std::getenv("VISFS_BA_TIMING") != nullptr;
    auto F0 = std::chrono::steady_clock::now();
    auto flap = [&](const char* what) { if (timing) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "  finish %-16s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - F0).count()); F0 = t; } };
    int n = 0;
    {
        const int No = pk.g.n_obs;
        const int T = (pool && No >= 65536) ? 4 * pool->size() : 1;
        std::vector<int> cnt(T + 1, 0);
        auto count = [&](int t, int) {
            const int lo = (int)((int64_t)No * t / T), hi = (int)((int64_t)No * (t + 1) / T);
            int c = 0;
            for (int k = lo; k < hi; ++k) c += outl[k] != 0;            // (no branch: the compiler vectorises the byte sum)
            cnt[t + 1] = c;
        };
        if (T > 1) pool->run(T, count); else count(0, 0);
        flap("count outliers");
        for (int t = 0; t < T; ++t) cnt[t + 1] += cnt[t];
        auto write = [&](int t, int) {
            const int lo = (int)((int64_t)No * t / T), hi = (int)((int64_t)No * (t + 1) / T);
            int at = cnt[t], k = lo;
            if (cnt[t + 1] == at) return;
            if ((int64_t)(cnt[t + 1] - at) * 16 > hi - lo && cnt[t + 1] <= r->outlier_capacity) {
                // many outliers (a window that started far off): a test per observation mispredicts every other time — store every
                // observation's ids at the cursor and advance it by the flag (the slot after the last outlier of the range is the next
                // range's first, or spare capacity: the caller sized the arrays for every reference)
                const int end = cnt[t + 1];
                for (; k < hi && at < end; ++k) {
                    const int ref = pk.obs_ref[k];
                    r->outlier_feature[at] = win->ref_feature[ref]; r->outlier_pose[at] = win->ref_pose[ref];
                    at += outl[k] != 0;
                }
                return;
            }
            while (k < hi) {
                if (k + 8 <= hi) { uint64_t eight; std::memcpy(&eight, outl + k, 8); if (eight == 0) { k += 8; continue; } }
                if (outl[k]) { if (at < r->outlier_capacity) { r->outlier_feature[at] = win->ref_feature[pk.obs_ref[k]]; r->outlier_pose[at] = win->ref_pose[pk.obs_ref[k]]; } ++at; }
                ++k;
            }
        };
        if (T > 1) pool->run(T, write); else write(0, 0);
        n = std::min(cnt[T], (int)r->outlier_capacity);
        flap("write outliers");
    }
    r->n_outliers = n;
    if (rc != VISFS_BA_OK) return rc;
    for (int i = 0; i < Np; ++i) { r->pose_ids_out[i] = win->pose_ids[i]; visfs_ba_unpack_pose(pose + (size_t)pose_stride * i, win->Trc, r->pose_Twr_out + 12 * i); }   // :320-340
    r->n_poses_out = Np;
    flap("poses");
    {
        const int T = (pool && Nl >= 16384) ? 2 * pool->size() : 1;
        auto body = [&](int t, int) {
            const int lo = (int)((int64_t)Nl * t / T), hi = (int)((int64_t)Nl * (t + 1) / T);
            for (int l = lo; l < hi; ++l) {                             // :343-358
                double* p = win->point_xyz + 3 * l;
                if (pk.point_used[l]) {
                    const double dx = p[0] - pts[3 * l], dy = p[1] - pts[3 * l + 1], dz = p[2] - pts[3 * l + 2];
                    if (std::sqrt(dx * dx + dy * dy + dz * dz) < 5.0) { p[0] = pts[3 * l]; p[1] = pts[3 * l + 1]; p[2] = pts[3 * l + 2]; }
                } else { p[0] = p[1] = p[2] = std::nan(""); }
            }
        };
        if (T > 1) pool->run(T, body); else body(0, 0);
        flap("landmarks");
    }
    return VISFS_BA_OK;
}

int solve_window_on(visfs_ba_handle* h, Workspace& w, const visfs_ba_window* win, visfs_ba_result* r) {
    static const bool timing = std::getenv("VISFS_BA_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (timing) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, " window %-14s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - T0).count()); T0 = t; } };
    PackedWindow& pk = w.pk;
    if (WorkerPool* pool = host_pool(h)) { if (win->n_refs >= 16384) pool->prewake(); }
    if (!prepare_window(h, w, win, r, pk, host_pool(h))) return r->status;
    lap("prepare");
    visfs_ba_stats st;
    w.want_outputs = true;                          // the outputs travel with the solve's state read
    const int rc = ws_optimize(h, w, &st, /*fresh_upload=*/true);
    w.want_outputs = false;
    lap("optimize");
    const int rcf = finish_window(h, w, win, r, pk, rc, st, host_pool(h));
    lap("finish");
    return rcf;
}



// Optimizer.cpp:261-318 for every member at once.  On return every member's LmState is in ws[i]->h_state.
int batch_optimize(visfs_ba_handle* h, BatchScratch& bs, const std::vector<Workspace*>& ws, const std::vector<int>& members, hipStream_t stream) {
    const int B = (int)members.size();
    if (B == 0) return VISFS_BA_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (bs.cap_graphs < (size_t)B) {
        if (bs.d_graphs) (void)hipFree(bs.d_graphs);
        bs.d_graphs = nullptr; bs.cap_graphs = 0;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&bs.d_graphs), (size_t)B * sizeof(DeviceGraph)));
        bs.cap_graphs = B;
    }
    if (bs.cap_states < (size_t)B) {
        if (bs.d_lm) (void)hipFree(bs.d_lm);
        if (bs.h_lm) (void)hipHostFree(bs.h_lm);
        bs.d_lm = nullptr; bs.h_lm = nullptr; bs.cap_states = 0;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&bs.d_lm), (size_t)B * sizeof(LmState)));
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&bs.h_lm), (size_t)B * sizeof(LmState), hipHostMallocDefault));
        bs.cap_states = B;
    }
    // the members were uploaded on their own streams and the batch runs on `stream`: their copies and index kernels must be done
    for (int b = 0; b < B; ++b) {
        Workspace& w = *ws[members[b]];
        if (w.upload_in_flight) { HIP_TRY(h, hipStreamSynchronize(w.stream)); w.upload_in_flight = false; }
    }
    std::vector<DeviceGraph>& hg = bs.host_graphs;
    hg.resize(B);
    std::vector<char> pristine(B, 0);
    for (int b = 0; b < B; ++b) { Workspace& w = *ws[members[b]]; pristine[b] = w.pristine ? 1 : 0; w.pristine = false; }
    LaunchDims d = dims_of(ws[members[0]]->g);
    bool fused = true, small_solve = true, fused_decide = true, spec_fused = true;
    for (int b = 0; b < B; ++b) {
        const Workspace& w = *ws[members[b]];
        hg[b] = w.g;
        d = dims_max(d, dims_of(w.g));
        fused = fused && w.fused; small_solve = small_solve && w.small_solve; fused_decide = fused_decide && w.fused_decide; spec_fused = spec_fused && w.spec_fused;
    }
    PcgLease pcg_lease;                                                                // persistent PCG: co-residency budget of the device
    if (!fused && !small_solve && !d.pcg_cu && h->prm.solver == 2) pcg_lease.acquire(h->device, pcg_wave_cost(d, B));
    HIP_TRY(h, hipMemcpyAsync(bs.d_graphs, hg.data(), (size_t)B * sizeof(DeviceGraph), hipMemcpyHostToDevice, stream));
    // g2o branch: optimize(iterations / 2) twice; Ceres branch: one Solve of <= iterations trust-region iterations, no second pass
    const bool ceres = h->prm.framework == 1;
    const int half = ceres ? h->prm.iterations : h->prm.iterations / 2;
    launch_reset_batch(bs.d_graphs, B, d, half, h->prm.trust_region == 1, 0, stream);
    if (fused) {
        launch_small_optimize_batch(bs.d_graphs, B, h->prm.solver, half, stream);
    }
    // As for a single window (ws_optimize): both phases and their device-gated ends are enqueued up front and the whole LM states
    // come back in ONE copy; windows that rejected trials are topped up from what comes back.  Every launch is gated per window,
    // so the same sequence is safe for windows at different points of the schedule.
    const int half2 = (h->prm.robust_kernel_delta > 0.0 && !ceres) ? half : 0;                      // :310-311
    auto units = [&](int n, bool first) { for (int u = 0; u < n; ++u) { launch_unit_batch(bs.d_graphs, B, d, first, small_solve, h->prm.solver, fused_decide, spec_fused, stream); first = false; } };
    auto phase_end = [&](int which) {
        if (which == 0) launch_phase_end_batch(bs.d_graphs, B, d, 0, 1, half2, stream);             // :270-303
        else launch_phase_end_batch(bs.d_graphs, B, d, 1, 0, 0, stream);                            // :315-318
    };
    if (!fused) { units(half, true); phase_end(0); units(half2, true); phase_end(1); }              // :265
    for (int guard = 0;; ++guard) {
        launch_gather_lm(bs.d_graphs, B, bs.d_lm, stream);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(bs.h_lm, bs.d_lm, (size_t)B * sizeof(LmState), hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
        if (fused) break;
        bool all_finished = true;
        int top_up = 0;
        for (int b = 0; b < B; ++b) {
            const LmState& st = bs.h_lm[b];
            // (a member whose hand-off timed out has stopped — status 8, every later launch a no-op for it —; the others go on, it is dealt
            // with below)
            if (st.ended >= 2 || st.status != 0) continue;
            all_finished = false;
            if (!st.done) top_up = std::max(top_up, std::max(1, st.max_iter - st.phase_iter));
        }
        if (all_finished) break;
        if (guard > 16 * h->prm.iterations + 32) { h->err = "LM state machine did not terminate"; return VISFS_BA_ERR_DEVICE; }
        units(top_up, false); phase_end(0); units(half2, true); phase_end(1);
    }
    int worst = VISFS_BA_OK;
    pcg_lease.release();
    for (int b = 0; b < B; ++b) {
        Workspace& w = *ws[members[b]];
        *w.h_state = bs.h_lm[b];
        const bool was_pristine = pristine[b];
        w.fell_back_last = false;
        if (bs.h_lm[b].status != VISFS_BA_ERR_DEVICE) continue;
        // this member's persistent PCG (or decision hand-off) timed out: solve it again, alone, on the direct solver (see ws_optimize)
        int rc = VISFS_BA_ERR_DEVICE;
        if (bs.h_lm[b].pcg_timeout && was_pristine && w.direct_ready && h->prm.solver == 2) {
            HIP_TRY(h, hipStreamSynchronize(stream));
            w.solver_now = 0; w.fallbacks += 1;
            launch_reset(w.g, half, h->prm.trust_region == 1, 1, w.stream);
            rc = hipGetLastError() == hipSuccess ? ws_optimize_run(h, w, nullptr, false, true) : (int)VISFS_BA_ERR_DEVICE;
            w.solver_now = -1;
            w.fell_back_last = rc != VISFS_BA_ERR_DEVICE;
        }
        if (rc == VISFS_BA_ERR_DEVICE) { h->err = "a device-side hand-off (persistent PCG / LM decision) timed out"; worst = VISFS_BA_ERR_DEVICE; }
    }
    return worst;
}

// One group of windows that share launches.  Groups of 8 and more members (PCG kernels k_pcg1 or k_pcg_cu) are cut into two halves
// that run side by side — second stream, second host thread, both within the device's co-residency budget — so that the
// latency-bound stretches of one half (PCG hand-offs, launch ramps) hide behind the gathers of the other: 8 C2 windows 58.7 -> 61.3 k
// it/s, 16 windows (k_pcg_cu) 70.5 -> 75.1 k, 32 windows 73.8 -> 77.6 k (bench.py --handles 2 shows the same from outside).  No result depends on how a batch is cut (every member's solve
// is bit-identical to its single-window solve).  VISFS_BA_BATCH_SPLIT=0 keeps one sequence.
int batch_optimize_group(visfs_ba_handle* h, const std::vector<int>& members) {
    static const int parts_env = []() { const char* e = std::getenv("VISFS_BA_BATCH_SPLIT"); return e ? std::atoi(e) : -1; }();   // 0 / 1: one sequence; n: n parts
    const int B = (int)members.size();
    // (production-size windows on k_small_solve LOSE when cut: 8 windows 156.5 -> 130.9 k it/s, 16 windows 284 -> 178 k — their
    // launches are too short for a second host thread to feed a second stream beside them)
    bool one_wave = h->prm.solver == 2, one_cu = h->prm.solver == 2;      // every member on k_pcg1 / every member on k_pcg_cu
    for (int i : members) {
        const Workspace& w = *h->batch[i];
        one_wave = one_wave && w.g.pcg1_code && !w.g.pcg_cu && !w.small_solve && !w.fused;
        one_cu = one_cu && w.g.pcg_cu && !w.small_solve && !w.fused;
    }
    int K = parts_env >= 0 ? parts_env : 2;
    K = std::min(K, B / 4);                                     // at least four windows per part
    if (K >= 2 && one_wave) {
        // the parts hold their leases at the same time: when they do not fit the device's budget together the second would only wait
        // for the first — a thread, a stream and two smaller launch sequences for nothing
        LaunchDims d = dims_of(h->batch[members[0]]->g);
        for (int i : members) d = dims_max(d, dims_of(h->batch[i]->g));
        const int per0 = (B + K - 1) / K;
        if ((int64_t)K * pcg_wave_cost(d, per0) > pcg_budget_capacity(h->device)) K = 1;
    }
    if (K < 2 || !(one_wave || one_cu) || B < 8) return batch_optimize(h, h->scratch, h->batch, members, h->ws.stream);
    HIP_TRY(h, hipSetDevice(h->device));
    while ((int)h->part_stream.size() < K - 1) {
        hipStream_t st = nullptr;
        HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        h->part_stream.push_back(st);
        h->part_scratch.emplace_back();
    }
    HIP_TRY(h, hipStreamSynchronize(h->ws.stream));           // a batch reset queued on the handle's stream precedes every part
    const int per = (B + K - 1) / K;
    std::vector<std::vector<int>> part(K);
    for (int k = 0; k < K; ++k) part[k].assign(members.begin() + std::min(B, k * per), members.begin() + std::min(B, (k + 1) * per));
    std::vector<int> rc(K, VISFS_BA_OK);
    std::vector<std::string> errs(K);
    std::vector<std::thread> th;
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{ th };
    try {
        th.reserve(K - 1);
        for (int k = 1; k < K; ++k) {
            if (part[k].empty()) continue;
            th.emplace_back([&, k]() noexcept {
                try {
                    (void)hipSetDevice(h->device);
                    visfs_ba_handle local;                        // per-thread error string; shares params / device
                    local.prm = h->prm; local.device = h->device; local.tuning = h->tuning;
                    rc[k] = batch_optimize(&local, h->part_scratch[k - 1], h->batch, part[k], h->part_stream[k - 1]);
                    errs[k] = local.err;
                } catch (...) { rc[k] = VISFS_BA_ERR_DEVICE; }
            });
        }
    } catch (...) {                                             // a thread could not be started: the joiner waits for the others, then one sequence
        for (auto& x : th) if (x.joinable()) x.join();
        return batch_optimize(h, h->scratch, h->batch, members, h->ws.stream);
    }
    rc[0] = batch_optimize(h, h->scratch, h->batch, part[0], h->ws.stream);
    for (auto& x : th) x.join();
    for (int k = 1; k < K; ++k) if (rc[0] == VISFS_BA_OK && rc[k] != VISFS_BA_OK) { rc[0] = rc[k]; if (!errs[k].empty()) h->err = errs[k]; }
    return rc[0];
}


}  // namespace

// ====================================================================== exported C ABI
extern "C" {

int visfs_ba_abi_version(void) { return VISFS_BA_ABI_VERSION; }

int visfs_ba_set_tuning(visfs_ba_handle* h, int32_t tuning) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (tuning != VISFS_BA_TUNE_LATENCY && tuning != VISFS_BA_TUNE_THROUGHPUT) return bad(h, "unknown tuning");
    h->tuning = tuning;
    return VISFS_BA_OK;
}

void visfs_ba_default_params(visfs_ba_params* p) {
    // Parameters.h:184-191
    p->framework = 0; p->solver = 0; p->trust_region = 0; p->iterations = 10;
    p->pixel_variance = 1.5; p->odometry_covariance = 0.00005; p->laser_covariance = 0.1; p->robust_kernel_delta = 8.0;
}

// why the last visfs_ba_create of this thread failed (no handle exists to carry the message)
static thread_local std::string tl_create_error;
const char* visfs_ba_create_error(void) { return tl_create_error.c_str(); }

int visfs_ba_create(const visfs_ba_params* params, int device_index, visfs_ba_handle** out) {
    try { tl_create_error.clear(); } catch (...) {}
    auto fail = [](int rc, const std::string& why) { try { tl_create_error = why; } catch (...) {} return rc; };
    if (!out || !params) return fail(VISFS_BA_ERR_BAD_ARGUMENT, "null argument");
    *out = nullptr;
    return guarded(nullptr, [&]() -> int {
        int n = 0;
        const hipError_t ec = hipGetDeviceCount(&n);
        if (ec != hipSuccess) return fail(VISFS_BA_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(ec) + " (no HIP runtime / driver?)");
        if (n <= 0) return fail(VISFS_BA_ERR_DEVICE, "no HIP device present");
        if (device_index < 0 || device_index >= n)
            return fail(VISFS_BA_ERR_BAD_ARGUMENT, "device index " + std::to_string(device_index) + " out of range: " + std::to_string(n) + " device(s) visible");
        hipDeviceProp_t prop;
        const hipError_t ep = hipGetDeviceProperties(&prop, device_index);
        if (ep != hipSuccess) return fail(VISFS_BA_ERR_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(ep));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)                                     // kernels are built for gfx950 only
            return fail(VISFS_BA_ERR_DEVICE, std::string("device ") + std::to_string(device_index) + " is " + prop.gcnArchName + ": the kernels are built for gfx950 (MI355X) only");
        visfs_ba_handle* h = new visfs_ba_handle();
        h->prm = *params;
        // the Ceres branch's linear_solver_type values (DENSE_SCHUR / DENSE_NORMAL_CHOLESKY / DENSE_QR, Optimizer.cpp:506-512) all solve the
        // damped normal equations exactly: one implementation, Schur elimination + the blocked dense Cholesky
        if (h->prm.framework == 1) h->prm.solver = 0;
        h->device = device_index;
        if (ws_init(h, h->ws) != VISFS_BA_OK) {
            const std::string why = "resource allocation on device " + std::to_string(device_index) + " failed: " + h->err;
            ws_release(h->ws);
            delete h;
            return fail(VISFS_BA_ERR_DEVICE, why);
        }
        *out = h;
        return (int)VISFS_BA_OK;
    });
}

void visfs_ba_destroy(visfs_ba_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    ws_release(h->ws);
    for (Workspace* w : h->batch) { ws_release(*w); delete w; }
    if (h->scratch.d_graphs) (void)hipFree(h->scratch.d_graphs);
    if (h->scratch.d_lm) (void)hipFree(h->scratch.d_lm);
    if (h->scratch.h_lm) (void)hipHostFree(h->scratch.h_lm);
    if (h->scratch.d_all) (void)hipFree(h->scratch.d_all);
    for (BatchScratch& b : h->part_scratch) {
        if (b.d_graphs) (void)hipFree(b.d_graphs);
        if (b.d_lm) (void)hipFree(b.d_lm);
        if (b.h_lm) (void)hipHostFree(b.h_lm);
    }
    for (hipStream_t st : h->part_stream) if (st) (void)hipStreamDestroy(st);
    delete h;
}

const char* visfs_ba_last_error(const visfs_ba_handle* h) { return h ? h->err.c_str() : "null handle"; }

int visfs_ba_pack_window(const visfs_ba_params* params, const visfs_ba_window* w,
                         double* pose_tq, uint8_t* pose_fixed, uint8_t* point_used,
                         int32_t* obs_point, int32_t* obs_pose, double* obs_uvr, int32_t* obs_ref,
                         int32_t* odo_from, int32_t* odo_to, double* odo_tq,
                         visfs_ba_graph* g, int32_t* n_mono_skipped) {
    (void)params;
    PackOut o;
    o.pose = pose_tq; o.pose_stride = 7; o.pose_fixed = pose_fixed; o.point_used = point_used;
    o.obs_point = obs_point; o.obs_pose = obs_pose; o.obs_uvr = obs_uvr; o.obs_ref = obs_ref;
    o.odo_from = odo_from; o.odo_to = odo_to; o.odo_tq = odo_tq;
    return pack_window_impl(w, o, g, n_mono_skipped, nullptr);
}

void visfs_ba_unpack_pose(const double* tq, const double* Trc, double* Twr_out) {
    // Twr = Tcw^-1 * Trc^-1   (Optimizer.cpp:324-329)
    const Rt T = pose_to_Rt(tq);
    const double Tcw[12] = { T.R.m00, T.R.m01, T.R.m02, T.t.x, T.R.m10, T.R.m11, T.R.m12, T.t.y, T.R.m20, T.R.m21, T.R.m22, T.t.z };
    double Twc[12], Tcr[12];
    iso_inv(Tcw, Twc);
    iso_inv(Trc, Tcr);
    iso_mul(Twc, Tcr, Twr_out);
}

int visfs_ba_graph_upload(visfs_ba_handle* h, const visfs_ba_graph* g) {
    if (!h || !g) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (const char* why = framework_refusal(h->prm)) { h->err = why; return VISFS_BA_ERR_UNSUPPORTED; }
    return guarded(h, [&]() { return ws_upload(h, h->ws, g); });
}

int visfs_ba_graph_reset(visfs_ba_handle* h) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (!h->ws.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    HIP_TRY(h, hipSetDevice(h->device));
    launch_reset(h->ws.g, h->ws.g.ceres ? h->prm.iterations : h->prm.iterations / 2, h->prm.trust_region == 1, 1, h->ws.stream);
    HIP_TRY(h, hipGetLastError());
    h->ws.pristine = true;
    return VISFS_BA_OK;
}

int visfs_ba_optimize(visfs_ba_handle* h, visfs_ba_stats* stats) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    return guarded(h, [&]() { return ws_optimize(h, h->ws, stats); });
}

int visfs_ba_graph_download(visfs_ba_handle* h, double* pose_tq, double* point_xyz, uint8_t* obs_outlier, double* obs_chi2) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    return guarded(h, [&]() { return ws_download(h, h->ws, pose_tq, point_xyz, obs_outlier, obs_chi2); });
}

int visfs_ba_solve_window(visfs_ba_handle* h, const visfs_ba_window* w, visfs_ba_result* r) {
    if (!h || !w || !r) return VISFS_BA_ERR_BAD_ARGUMENT;
    return guarded(h, [&]() { return solve_window_on(h, h->ws, w, r); });
}

int visfs_ba_solve_batch(visfs_ba_handle* h, int32_t n, const visfs_ba_window* const* w, visfs_ba_result* const* r) {
    if (!h || n < 0 || (n > 0 && (!w || !r))) return VISFS_BA_ERR_BAD_ARGUMENT;
    for (int i = 0; i < n; ++i) if (!w[i] || !r[i]) return VISFS_BA_ERR_BAD_ARGUMENT;
    // Independent windows (BASELINE config 5).  Host work (graph build, upload, write-back) runs on up to 8
    // threads, one workspace per window; the optimisation itself is ONE sequence of launches per group of windows with the
    // same launch-geometry class, blockIdx.y = window (batch_optimize).  Windows that cannot share launches (direct solver on
    // reduced systems above 64 x 64, or VISFS_BA_BATCH=0) are optimised one after another on the calling thread, each on its own
    // stream: their persistent-PCG / panel launches would gain nothing from overlapping and PCG grids must fit the device together (PcgLease).
    return guarded(h, [&]() -> int {
        while ((int)h->batch.size() < n) { h->batch.push_back(new Workspace()); h->batch.back()->batch_member = true; }
        for (int i = 0; i < n; ++i) h->batch[i]->batch_hint = n;
        const int lanes = std::max(1, std::min<int>(n, 8));
        std::vector<int> need(n, 0), rcs(n, VISFS_BA_OK);
        std::vector<std::string> errs(lanes);
        auto parallel = [&](auto&& fn) {
            std::vector<std::thread> th;
            struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{ th };
            th.reserve(lanes);
            for (int t = 0; t < lanes; ++t) {
                th.emplace_back([&, t]() noexcept {
                    int i = t;
                    try {
                        (void)hipSetDevice(h->device);
                        visfs_ba_handle local;            // per-thread error string; shares params / device
                        local.prm = h->prm; local.device = h->device; local.tuning = h->tuning;
                        for (; i < n; i += lanes) fn(local, i);
                        if (!local.err.empty()) errs[t] = local.err;
                    } catch (...) {                       // nothing may escape a thread: mark this lane's remaining windows failed
                        for (; i < n; i += lanes) { rcs[i] = VISFS_BA_ERR_DEVICE; need[i] = 0; r[i]->status = VISFS_BA_ERR_DEVICE; r[i]->n_poses_out = 0; }
                    }
                });
            }
        };
        parallel([&](visfs_ba_handle& local, int i) { need[i] = prepare_window(&local, *h->batch[i], w[i], r[i], h->batch[i]->pk); if (!need[i]) rcs[i] = r[i]->status; });
        // groups: lanes per landmark, PCG variant class, small-solve / fused flags must agree inside one batched launch
        const char* env = std::getenv("VISFS_BA_BATCH");
        const bool batching = !(env && env[0] == '0');
        std::map<std::array<int, 5>, std::vector<int>> groups;
        std::vector<int> singles;
        for (int i = 0; i < n; ++i) {
            if (!need[i]) continue;
            const Workspace& ws = *h->batch[i];
            // (the direct solver shares launches when the window's S is banded: k_band_chol, one workgroup per window)
            const bool band = h->prm.solver != 2 && !ws.small_solve && !ws.fused && ws.g.band_B >= 0;
            const bool batchable = batching && (h->prm.framework == 0 || ws.small_solve || band) && (h->prm.solver == 2 || ws.small_solve || ws.fused || band) && ws.g.Np <= MAX_STAGED_POSES && ws.g.Npf <= MAX_PCG_ONE_ROW_POSES;
            if (!batchable) { singles.push_back(i); continue; }
            const int cls = band ? 4 : ws.g.pcg_cu ? 3 : ws.g.Npf <= 64 ? 0 : ws.g.Npf <= 128 ? 1 : 2;
            groups[{ ws.g.group, cls, ws.small_solve ? 1 : 0, ws.fused ? 1 : 0, (ws.spec_fused ? 1 : 0) | (ws.g.n_runs > 0 ? 2 : 0) }].push_back(i);
        }
        std::vector<visfs_ba_stats> stats(n);
        int worst = VISFS_BA_OK;
        for (auto& kv : groups) {
            // the persistent PCG grid of one launch sequence must be resident as a whole: sized from the device's occupancy
            const std::vector<int>& all = kv.second;
            const int per = batch_members_per_launch(h, h->batch, all);
            for (size_t o = 0; o < all.size(); o += per) {
                std::vector<int> members(all.begin() + o, all.begin() + std::min(all.size(), o + per));
                const int rc = batch_optimize_group(h, members);
                for (int i : members) {
                    if (rc != VISFS_BA_OK) { rcs[i] = rc; worst = VISFS_BA_ERR_DEVICE; continue; }
                    fill_stats(*h->batch[i]->h_state, &stats[i]);
                    stats[i].solver_fallback = h->batch[i]->fell_back_last ? 1 : 0;
                    rcs[i] = h->batch[i]->h_state->status;
                }
            }
        }
        for (int i : singles) rcs[i] = ws_optimize(h, *h->batch[i], &stats[i]);
        parallel([&](visfs_ba_handle& local, int i) { if (need[i]) rcs[i] = finish_window(&local, *h->batch[i], w[i], r[i], h->batch[i]->pk, rcs[i], stats[i]); });
        for (int i = 0; i < n; ++i) if (rcs[i] == VISFS_BA_ERR_DEVICE) worst = VISFS_BA_ERR_DEVICE;
        for (int t = 0; t < lanes; ++t) if (!errs[t].empty()) h->err = errs[t];
        return worst;
    });
}

int visfs_ba_solve_batch_sharded(visfs_ba_handle* const* handles, int32_t n_handles, int32_t n, const visfs_ba_window* const* w, visfs_ba_result* const* r) {
    if (!handles || n_handles < 1 || n < 0 || (n > 0 && (!w || !r))) return VISFS_BA_ERR_BAD_ARGUMENT;
    for (int k = 0; k < n_handles; ++k) if (!handles[k]) return VISFS_BA_ERR_BAD_ARGUMENT;
    for (int k = 0; k < n_handles; ++k) for (int m = 0; m < k; ++m) if (handles[k] == handles[m]) return VISFS_BA_ERR_BAD_ARGUMENT;   // a handle is not thread-safe
    if (n == 0) return VISFS_BA_OK;
    const int per = (n + n_handles - 1) / n_handles;          // contiguous blocks: window i -> handle i / per (visfs_amd/dist.py:shard_windows)
    std::vector<int> rc(n_handles, VISFS_BA_OK);
    std::vector<std::thread> th;
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{ th };
    try {
        th.reserve(n_handles);
        for (int k = 0; k < n_handles; ++k) {
            const int lo = std::min(n, k * per), hi = std::min(n, lo + per);
            if (lo >= hi) continue;
            th.emplace_back([&, k, lo, hi]() noexcept { rc[k] = visfs_ba_solve_batch(handles[k], hi - lo, w + lo, r + lo); });
        }
    } catch (...) { return VISFS_BA_ERR_DEVICE; }              // thread creation failed: the joiner waits for what was started
    for (auto& x : th) x.join();
    int worst = VISFS_BA_OK;
    for (int k = 0; k < n_handles; ++k) if (rc[k] != VISFS_BA_OK) worst = rc[k];
    return worst;
}

// GRAPH layer for a batch: n graphs resident side by side, optimised by one sequence of batched launches (bench config 5).
int visfs_ba_batch_upload(visfs_ba_handle* h, int32_t n, const visfs_ba_graph* const* graphs) {
    if (!h || n < 0 || (n > 0 && !graphs)) return VISFS_BA_ERR_BAD_ARGUMENT;
    for (int i = 0; i < n; ++i) if (!graphs[i]) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (const char* why = framework_refusal(h->prm)) { h->err = why; return VISFS_BA_ERR_UNSUPPORTED; }
    return guarded(h, [&]() -> int {
        while ((int)h->batch.size() < n) { h->batch.push_back(new Workspace()); h->batch.back()->batch_member = true; }
        for (int i = 0; i < n; ++i) h->batch[i]->batch_hint = n;
        h->n_batch = 0;
        for (int i = 0; i < n; ++i) { const int rc = ws_upload(h, *h->batch[i], graphs[i]); if (rc != VISFS_BA_OK) return rc; }
        // the resident batch is reset and optimised on the handle's stream: drain the per-window upload streams once, here
        for (int i = 0; i < n; ++i) { HIP_TRY(h, hipStreamSynchronize(h->batch[i]->stream)); h->batch[i]->upload_in_flight = false; }
        // every graph once more as one array: visfs_ba_batch_reset is then a single launch
        BatchScratch& bs = h->scratch;
        if (bs.cap_all < (size_t)n) {
            if (bs.d_all) (void)hipFree(bs.d_all);
            bs.d_all = nullptr; bs.cap_all = 0;
            HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&bs.d_all), (size_t)std::max(n, 1) * sizeof(DeviceGraph)));
            bs.cap_all = (size_t)std::max(n, 1);
        }
        std::vector<DeviceGraph> hg(n);
        for (int i = 0; i < n; ++i) { hg[i] = h->batch[i]->g; bs.all_dims = i ? dims_max(bs.all_dims, dims_of(hg[i])) : dims_of(hg[i]); }
        if (n) HIP_TRY(h, hipMemcpy(bs.d_all, hg.data(), (size_t)n * sizeof(DeviceGraph), hipMemcpyHostToDevice));
        h->n_batch = n;
        return VISFS_BA_OK;
    });
}

int visfs_ba_batch_reset(visfs_ba_handle* h) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->n_batch > 0) launch_reset_batch(h->scratch.d_all, h->n_batch, h->scratch.all_dims, h->prm.framework == 1 ? h->prm.iterations : h->prm.iterations / 2, h->prm.trust_region == 1, 1, h->ws.stream);
    HIP_TRY(h, hipGetLastError());
    for (int i = 0; i < h->n_batch; ++i) h->batch[i]->pristine = true;
    return VISFS_BA_OK;
}

int visfs_ba_batch_optimize(visfs_ba_handle* h, visfs_ba_stats* stats) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    return guarded(h, [&]() -> int {
        const int n = h->n_batch;
        if (n == 0) { h->err = "no batch resident"; return VISFS_BA_ERR_NOT_LOADED; }
        std::map<std::array<int, 5>, std::vector<int>> groups;
        for (int i = 0; i < n; ++i) {
            const Workspace& ws = *h->batch[i];
            const bool band = h->prm.solver != 2 && !ws.small_solve && !ws.fused && ws.g.band_B >= 0;
            if (!(h->prm.solver == 2 || ws.small_solve || ws.fused || band)) { h->err = "batched launches need Optimizer/Solver=2, a banded reduced system (direct solver) or reduced systems <= 64 x 64"; return VISFS_BA_ERR_UNSUPPORTED; }
            if (ws.g.Np > MAX_STAGED_POSES || ws.g.Npf > MAX_PCG_ONE_ROW_POSES) { h->err = "windows of more than 840 poses / 256 free poses cannot share launches: solve them one by one"; return VISFS_BA_ERR_UNSUPPORTED; }
            const int cls = band ? 4 : ws.g.pcg_cu ? 3 : ws.g.Npf <= 64 ? 0 : ws.g.Npf <= 128 ? 1 : 2;
            groups[{ ws.g.group, cls, ws.small_solve ? 1 : 0, ws.fused ? 1 : 0, (ws.spec_fused ? 1 : 0) | (ws.g.n_runs > 0 ? 2 : 0) }].push_back(i);
        }
        int worst = VISFS_BA_OK;
        for (auto& kv : groups) {
            const std::vector<int>& all = kv.second;
            const int per = batch_members_per_launch(h, h->batch, all);
            for (size_t o = 0; o < all.size(); o += per) {
                std::vector<int> members(all.begin() + o, all.begin() + std::min(all.size(), o + per));
                const int rc = batch_optimize_group(h, members);
                if (rc != VISFS_BA_OK) return rc;
                for (int i : members) {
                    if (stats) { fill_stats(*h->batch[i]->h_state, &stats[i]); stats[i].solver_fallback = h->batch[i]->fell_back_last ? 1 : 0; }
                    if (h->batch[i]->h_state->status != VISFS_BA_OK) worst = h->batch[i]->h_state->status;
                }
            }
        }
        return worst;
    });
}

int visfs_ba_batch_download(visfs_ba_handle* h, int32_t index, double* pose_tq, double* point_xyz, uint8_t* obs_outlier, double* obs_chi2) {
    if (!h || index < 0 || index >= h->n_batch) return VISFS_BA_ERR_BAD_ARGUMENT;
    return guarded(h, [&]() { return ws_download(h, *h->batch[index], pose_tq, point_xyz, obs_outlier, obs_chi2); });
}

int visfs_ba_graph_describe(visfs_ba_handle* h, visfs_ba_graph_info* out) {
    if (!h || !out) return VISFS_BA_ERR_BAD_ARGUMENT;
    const Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    out->n_poses = w.g.Np; out->n_free_poses = w.g.Npf; out->n_points = w.g.Nl; out->n_obs = w.g.No; out->n_odo = w.g.Ne;
    out->n_blk = w.g.n_blk; out->n_pairs = w.n_pairs; out->lanes_per_landmark = w.g.group; out->n_schur_chunks = w.g.n_sch;
    out->schur_runs = w.g.n_runs; out->schur_run_landmarks = w.g.run_lr * w.g.run_m;
    out->device_bytes = (int64_t)w.device_bytes;
    out->fused_path = w.fused ? 1 : 0;
    out->band_blocks = (h->prm.solver != 2 && !w.small_solve) ? w.g.band_B : -1;
    out->graph_replayed = w.last_replayed ? 1 : 0;
    out->unit_form = w.spec_fused ? 2 : w.spec ? 1 : 0;
    out->solver_kernel = w.small_solve ? 5 : h->prm.solver != 2 ? (w.g.band_B >= 0 ? 7 : 6) : w.g.pcg_cu ? 4 : w.g.pcg1_code ? 1 : w.g.Npf > MAX_PCG_ONE_ROW_POSES ? 3 : 2;
    return VISFS_BA_OK;
}

int visfs_ba_profile_enable(visfs_ba_handle* h, uint32_t mask) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    h->ws.prof_mask = mask;
    return VISFS_BA_OK;
}

int visfs_ba_profile_read(visfs_ba_handle* h, visfs_ba_profile* out) {
    if (!h || !out) return VISFS_BA_ERR_BAD_ARGUMENT;
    Workspace& w = h->ws;
    if (w.stream) { HIP_TRY(h, hipStreamSynchronize(w.stream)); prof_harvest(w); }
    std::memset(out, 0, sizeof(*out));
    for (int k = 0; k < VISFS_BA_K_COUNT; ++k) {
        std::vector<float>& d = w.durs[k];
        std::sort(d.begin(), d.end(), [](float a, float b) { return a > b; });
        out->launches[k] = (int64_t)d.size();
        const int64_t na = std::min<int64_t>(w.active[k], (int64_t)d.size());
        out->active_launches[k] = na;
        for (size_t i = 0; i < d.size(); ++i) { out->total_ms[k] += d[i]; if ((int64_t)i < na) out->active_ms[k] += d[i]; }
        d.clear();
        w.active[k] = 0;
    }
    // what an event pair costs by itself: 15 empty pairs back to back, median
    if (w.stream) {
        float el[15];
        int n = 0;
        w.ev_used = 0;
        hipEvent_t ea[15], eb[15];
        for (; n < 15; ++n) {
            ea[n] = ProfScope::take(w); eb[n] = ProfScope::take(w);
            if (!ea[n] || !eb[n]) break;
            if (hipEventRecord(ea[n], w.stream) != hipSuccess || hipEventRecord(eb[n], w.stream) != hipSuccess) break;
        }
        if (n > 0 && hipStreamSynchronize(w.stream) == hipSuccess) {
            int m = 0;
            for (int i = 0; i < n; ++i) if (hipEventElapsedTime(&el[m], ea[i], eb[i]) == hipSuccess) ++m;
            if (m > 0) { std::sort(el, el + m); out->null_pair_ms = el[m / 2]; }
        }
        w.ev_used = 0;
    }
    return VISFS_BA_OK;
}

// ---------------------------------------------------------------- stage hooks
int visfs_ba_graph_free_poses(visfs_ba_handle* h) { return (h && h->ws.loaded) ? h->ws.g.Npf : -1; }

int visfs_ba_stage_linearize(visfs_ba_handle* h, double* robust_chi2, double* max_diag) {
    if (h && h->prm.framework != 0) { h->err = "the stage hooks step the g2o branch only (Optimizer/Framework=0)"; return VISFS_BA_ERR_UNSUPPORTED; }
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    DeviceGraph g = w.g;
    g.debug = 1;
    launch_stage_arm(g, 0.0, MODE_LIN, w.stream);
    launch_linearize(g, w.stream);
    launch_lin_finalize(g, 1, w.stream);
    HIP_TRY(h, hipGetLastError());
    int rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    if (robust_chi2) *robust_chi2 = w.h_state->current_chi;
    if (max_diag) *max_diag = w.h_state->max_diag;
    return VISFS_BA_OK;
}

int visfs_ba_stage_trial(visfs_ba_handle* h, double lambda, double* trial_chi2, double* scale, int32_t* pcg_iterations, int32_t* solver_ok) {
    if (h && h->prm.framework != 0) { h->err = "the stage hooks step the g2o branch only (Optimizer/Framework=0)"; return VISFS_BA_ERR_UNSUPPORTED; }
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    int rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    const int sel0 = w.h_state->sel;
    const double chi0 = w.h_state->current_chi;
    launch_stage_arm(w.g, lambda, MODE_TRIAL, w.stream);
    launch_schur_partial(w.g, w.stream);
    if (w.small_solve) launch_small_solve(w.g, h->prm.solver, w.stream);
    else {
        if (!(h->prm.solver == 2 && w.g.fin_pcg && w.g.pcg1_code != nullptr && !w.g.pcg_cu)) launch_schur_finalize(w.g, w.stream);     // (else: on board the PCG launch)
        if (h->prm.solver == 2) launch_pcg(w.g, w.stream); else launch_direct(w.g, w.stream);
    }
    launch_backsub(w.g, w.stream);
    HIP_TRY(h, hipGetLastError());
    rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    const int ok = !w.h_state->solver_failed && !w.h_state->pcg_timeout;
    if (pcg_iterations) *pcg_iterations = (h->prm.solver == 2) ? w.h_state->pcg_iter : 0;
    if (solver_ok) *solver_ok = ok;
    launch_decide(w.g, w.stream);                    // production K9 computes tempChi / scale ...
    HIP_TRY(h, hipGetLastError());
    rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    if (trial_chi2) *trial_chi2 = w.h_state->temp_chi;
    if (scale) *scale = w.h_state->scale - 1e-3;     // report computeScale() without the +1e-3 guard
    // ... and the hook then undoes the commit so the resident estimate is unchanged
    LmState st = *w.h_state;
    st.sel = sel0; st.current_chi = chi0; st.done = 0; st.mode = 0;
    *w.h_state = st;
    HIP_TRY(h, hipMemcpyAsync(w.g.st, w.h_state, sizeof(LmState), hipMemcpyHostToDevice, w.stream));
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    return VISFS_BA_OK;
}

static int stage_fetch_impl(visfs_ba_handle* h, int32_t which, double* dst, size_t n_doubles) {
    Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    const DeviceGraph& g = w.g;
    const size_t n6 = (size_t)w.n6();
    int rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    const int sel = w.h_state->sel;
    const double* src = nullptr; size_t m = 0;
    switch (which) {
        case VISFS_BA_BUF_OBS_ERR: src = g.obs_err; m = (size_t)g.No * 3; break;
        case VISFS_BA_BUF_OBS_CHI2: src = g.obs_chi2; m = g.No; break;
        case VISFS_BA_BUF_OBS_WEIGHT: src = g.lin[w.h_state->lin_sel & 1].obs_w; m = g.No; break;
        case VISFS_BA_BUF_HPL: src = g.W; m = (size_t)g.No * 18; break;
        case VISFS_BA_BUF_HLL: src = g.lin[w.h_state->lin_sel & 1].Hll; m = (size_t)g.Nl * 6; break;
        case VISFS_BA_BUF_BL: src = g.lin[w.h_state->lin_sel & 1].bl; m = (size_t)g.Nl * 3; break;
        case VISFS_BA_BUF_BP: src = g.bp; m = n6; break;
        case VISFS_BA_BUF_BS: src = g.bs; m = n6; break;
        case VISFS_BA_BUF_DX_POSE: src = g.x; m = n6; break;
        case VISFS_BA_BUF_DX_POINT: src = g.dxl; m = (size_t)g.Nl * 3; break;
        case VISFS_BA_BUF_POINT_TRIAL: src = g.pt[sel ^ 1]; m = (size_t)g.Nl * 3; break;
        case VISFS_BA_BUF_HPP: case VISFS_BA_BUF_S: m = n6 * n6; break;
        case VISFS_BA_BUF_POSE_TRIAL: m = (size_t)g.Np * 7; break;
        case 101: src = g.pcg_cu ? g.S_rows : nullptr; m = g.pcg_cu ? (size_t)g.pcg_max_row * 6 * g.cu_T : 0; if (!src) return bad(h, "not a k_pcg_cu window"); break;     // diagnostic: S by scalar row
        case 100: {   // diagnostic: raw PCG stamps (only meaningful in a -DVISFS_BA_STAMPS build)
            if (n_doubles < 128) return bad(h, "destination too small");
            HIP_TRY(h, hipMemcpyAsync(dst, g.stamps, 128 * 8, hipMemcpyDeviceToHost, w.stream));
            HIP_TRY(h, hipStreamSynchronize(w.stream));
            return VISFS_BA_OK;
        }
        default: return bad(h, "unknown buffer id");
    }
    if (n_doubles < m) return bad(h, "destination too small");
    if (src) {
        if (m) HIP_TRY(h, hipMemcpyAsync(dst, src, m * 8, hipMemcpyDeviceToHost, w.stream));
        HIP_TRY(h, hipStreamSynchronize(w.stream));
        return VISFS_BA_OK;
    }
    if (which == VISFS_BA_BUF_POSE_TRIAL) {
        std::vector<double> tmp((size_t)g.Np * POSE_STRIDE);
        HIP_TRY(h, hipMemcpyAsync(tmp.data(), g.pose[sel ^ 1], tmp.size() * 8, hipMemcpyDeviceToHost, w.stream));
        HIP_TRY(h, hipStreamSynchronize(w.stream));
        for (int i = 0; i < g.Np; ++i) for (int q = 0; q < 7; ++q) dst[7 * i + q] = tmp[POSE_STRIDE * i + q];
        return VISFS_BA_OK;
    }
    std::memset(dst, 0, m * 8);
    if (which == VISFS_BA_BUF_S) {
        std::vector<double> blk((size_t)g.n_blk * 36);
        HIP_TRY(h, hipMemcpyAsync(blk.data(), g.S, blk.size() * 8, hipMemcpyDeviceToHost, w.stream));
        HIP_TRY(h, hipStreamSynchronize(w.stream));
        for (int b = 0; b < g.n_blk; ++b)
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                const double v = blk[36 * (size_t)b + 6 * r + c];
                dst[(size_t)(6 * w.blk_i[b] + r) * n6 + 6 * w.blk_j[b] + c] = v;
                dst[(size_t)(6 * w.blk_j[b] + c) * n6 + 6 * w.blk_i[b] + r] = v;
            }
        return VISFS_BA_OK;
    }
    // HPP: diagonal blocks + odometry off-diagonal blocks
    std::vector<double> diag((size_t)std::max(g.Npf, 1) * 36), odo((size_t)std::max(g.Ne, 1) * 120);
    HIP_TRY(h, hipMemcpyAsync(diag.data(), g.Hpp, (size_t)g.Npf * 36 * 8, hipMemcpyDeviceToHost, w.stream));
    if (g.Ne) HIP_TRY(h, hipMemcpyAsync(odo.data(), g.lin[w.h_state->lin_sel & 1].odo_blk, (size_t)g.Ne * 120 * 8, hipMemcpyDeviceToHost, w.stream));
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    for (int a = 0; a < g.Npf; ++a)
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) dst[(size_t)(6 * a + r) * n6 + 6 * a + c] = diag[36 * (size_t)a + 6 * r + c];
    for (int e = 0; e < g.Ne; ++e) {
        const int a = w.pose_free[w.odo_i[e]], b = w.pose_free[w.odo_j[e]];
        if (a < 0 || b < 0) continue;
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
            const double v = odo[120 * (size_t)e + 72 + 6 * r + c];
            dst[(size_t)(6 * a + r) * n6 + 6 * b + c] += v;
            dst[(size_t)(6 * b + c) * n6 + 6 * a + r] += v;
        }
    }
    return VISFS_BA_OK;
}

// Stage hooks for stepping a solve by hand (tools/soak_diverge.py): commit the last trial state, start a phase
// (LinearSolverPCG::init()), run the outlier pass on the committed estimate.
int visfs_ba_stage_commit(visfs_ba_handle* h) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (h->prm.framework != 0) { h->err = "the stage hooks step the g2o branch only (Optimizer/Framework=0)"; return VISFS_BA_ERR_UNSUPPORTED; }
    return guarded(h, [&]() -> int {
    HIP_TRY(h, hipSetDevice(h->device));          // a process may hold handles on several GPUs
    Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    int rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    w.h_state->sel ^= 1;
    HIP_TRY(h, hipMemcpyAsync(w.g.st, w.h_state, sizeof(LmState), hipMemcpyHostToDevice, w.stream));
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    return (int)VISFS_BA_OK;
    });
}
int visfs_ba_stage_begin_phase(visfs_ba_handle* h) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (h->prm.framework != 0) { h->err = "the stage hooks step the g2o branch only (Optimizer/Framework=0)"; return VISFS_BA_ERR_UNSUPPORTED; }
    return guarded(h, [&]() -> int {
    HIP_TRY(h, hipSetDevice(h->device));          // a process may hold handles on several GPUs
    Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    int rc = ws_read_state(h, w);
    if (rc != VISFS_BA_OK) return rc;
    w.h_state->pcg_residual = -1.0; w.h_state->pcg_res_in = -1.0; w.h_state->status = 0;
    HIP_TRY(h, hipMemcpyAsync(w.g.st, w.h_state, sizeof(LmState), hipMemcpyHostToDevice, w.stream));
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    return (int)VISFS_BA_OK;
    });
}
int visfs_ba_stage_mark_outliers(visfs_ba_handle* h) {
    if (!h) return VISFS_BA_ERR_BAD_ARGUMENT;
    if (h->prm.framework != 0) { h->err = "the stage hooks step the g2o branch only (Optimizer/Framework=0)"; return VISFS_BA_ERR_UNSUPPORTED; }
    return guarded(h, [&]() -> int {
    HIP_TRY(h, hipSetDevice(h->device));          // a process may hold handles on several GPUs
    Workspace& w = h->ws;
    if (!w.loaded) { h->err = "no graph resident"; return VISFS_BA_ERR_NOT_LOADED; }
    launch_eval_mark(w.g, w.stream);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(w.stream));
    return (int)VISFS_BA_OK;
    });
}

int visfs_ba_hook_lm_script(int32_t gauss_newton, int32_t n_iter, double chi0, double max_diag0, int32_t n_trials,
                            const double* temp_chi, const double* scale, const int32_t* ok, visfs_ba_stats* stats) {
    if (!temp_chi || !scale || !ok || !stats || n_trials < 1) return VISFS_BA_ERR_BAD_ARGUMENT;
    LmState st;
    const int used = lm_script_host(gauss_newton, n_iter, chi0, max_diag0, n_trials, temp_chi, scale, ok, &st);
    fill_stats(st, stats);
    return used;
}

int visfs_ba_hook_ceres_script(int32_t max_iter, double cost0, double x_norm0, double grad_max0, int32_t n, const int32_t* ok, const double* model_cost_change,
                               const double* cand_cost, const double* step_norm, const double* grad_max, const double* x_norm, visfs_ba_stats* stats) {
    if (!ok || !model_cost_change || !cand_cost || !step_norm || !grad_max || !x_norm || !stats || n < 1) return VISFS_BA_ERR_BAD_ARGUMENT;
    LmState st;
    const int reason = ceres_script_host(max_iter, cost0, x_norm0, grad_max0, n, ok, model_cost_change, cand_cost, step_norm, grad_max, x_norm, &st);
    fill_stats(st, stats);
    return reason;
}

int visfs_ba_hook_dogleg_script(int32_t max_iter, double cost0, double x_norm0, double grad_max0, int32_t n, const int32_t* ok, const double* model_cost_change,
                                const double* cand_cost, const double* step_norm, const double* dogleg_step_norm, const double* grad_max, const double* x_norm,
                                visfs_ba_stats* stats, double* mu_trace) {
    if (!ok || !model_cost_change || !cand_cost || !step_norm || !dogleg_step_norm || !grad_max || !x_norm || !stats || !mu_trace || n < 1) return VISFS_BA_ERR_BAD_ARGUMENT;
    LmState st;
    const int reason = ceres_script_host(max_iter, cost0, x_norm0, grad_max0, n, ok, model_cost_change, cand_cost, step_norm, grad_max, x_norm, &st, dogleg_step_norm, mu_trace);
    fill_stats(st, stats);
    return reason;
}

int visfs_ba_hook_dogleg_combine(double s1, double s2, double s3, double jv2, double radius, double mu, double out[4]) {
    if (!out) return VISFS_BA_ERR_BAD_ARGUMENT;
    dogleg_combine(s1, s2, s3, jv2, radius, mu, out[0], out[1], out[2], out[3]);
    return VISFS_BA_OK;
}

int visfs_ba_stage_fetch(visfs_ba_handle* h, int32_t which, double* dst, size_t n_doubles) {
    if (!h || !dst) return VISFS_BA_ERR_BAD_ARGUMENT;
    return guarded(h, [&]() { return stage_fetch_impl(h, which, dst, n_doubles); });
}

}  // extern "C"
