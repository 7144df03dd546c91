// ba_device.hpp — HBM layout of one resident window and the device-side LM state.
//
// Layout rules:
//   * observations are LANDMARK-MAJOR (sorted by point, then pose) — the reference's edge
//     insertion order (Optimizer.cpp:156-221) — so a landmark's tiles are contiguous;
//   * a second POSE-MAJOR permutation (pose_obs) cut into chunks of <= LIN_CHUNK
//     observations feeds the Hpp/b_p reduction without atomics;
//   * per S block (i<=j) a list of (obs_a, obs_b) pairs sharing a landmark — g2o's
//     buildStructure analogue — feeds the Schur complement as a gather, no atomics;
//   * every accumulation is a fixed-order tree or serial sum: results are bitwise
//     reproducible run to run.
#pragma once

#include <stdint.h>

#include "ba_math.hpp"

namespace visfs_ba {

#ifndef VISFS_BA_POSE_SEEDS
#define VISFS_BA_POSE_SEEDS 1        // 0: A/B builds — the Schur gather reads the landmark-major seeds (round 1)
#endif
#ifndef VISFS_BA_RS_SWAP
#define VISFS_BA_RS_SWAP 1           // 0: A/B builds — the reduce-scatters select send / keep and move one of them (round 1)
#endif
constexpr int LIN_CHUNK = 256;        // observations per pose-major workgroup
constexpr int MAX_TRACE = 64;         // == VISFS_BA_MAX_TRACE
constexpr int POSE_STRIDE = 8;        // doubles per pose in HBM (7 used; 64-byte rows)
constexpr int MAX_STAGED_POSES = 840; // poses staged as R|t in LDS (12 doubles each; k_backsub stages two sets: 24 * 840 * 8 B + scratch <= 160 KiB);
                                      // larger windows take the kernels that read the poses from HBM (PoseSrc<false>)
constexpr int MAX_PCG_ONE_ROW_POSES = 256; // persistent PCG with one workgroup per block row: all co-resident (256 CUs, >= 1 workgroup each)
constexpr int MAX_PCG_FREE_POSES = 1024;   // beyond 256 free poses a workgroup owns several block rows (<= 256 workgroups) and an owner thread up to 4 blocks
constexpr int RUN_MAX_W = 64;        // k_schur_runs: widest pose span of a run of landmarks (slot table [landmarks][span], 16-bit entries)
constexpr int RUN_MAX_TILES = 416;   // ... observations of one sub-batch (21 doubles of LDS each: two workgroups per CU)
constexpr int RUN_TILE = 21;         // ... doubles per staged tile: Q = N D (9), N (9), Pc (3)
constexpr int SCH_CHUNK = 64;         // co-observation pairs per Schur wavefront and pass (DeviceGraph::sch_chunk = 64 x passes)
// fused single-workgroup path (k_small_optimize): limits of a "small" window
constexpr int SM_MAX_POSES = 16;      // R|t of every pose twice in LDS
constexpr int SM_MAX_N6 = 64;         // <= 10 free poses: the reduced camera system is solved in LDS
constexpr int SM_MAX_WCHUNKS = 192;   // 64-observation wave chunks of the pose-major pass (4 per LIN_CHUNK)
constexpr int SM_MAX_OBS = 8192;      // beyond this one CU's arithmetic costs more than the launches it saves
constexpr int SM_MAX_SCH = 1024;      // Schur chunks (<= 64 k co-observation pairs)

// LM / phase state machine, resident in HBM; every kernel of a "unit" reads its gate from here.
// Unit gate bits (LmState::mode).  Written ONLY by single-workgroup kernels (k_reset, k_decide, k_phase_end,
// k_stage_arm), so no multi-workgroup launch ever reads a gate that the same launch modifies.
constexpr int MODE_LIN = 1;     // this unit linearises (first unit of a phase, or the previous trial was accepted)
constexpr int MODE_TRIAL = 2;   // this unit runs one damped solve

struct LmState {
    double lambda, ni, current_chi, temp_chi, rho, scale, max_diag;
    double pcg_res_in;      // LinearSolverPCG::_residual as seen by the solve of this unit
    double pcg_residual;    // ... as left by the last solve (copied to pcg_res_in by k_schur_finalize)
    double chi2_initial, chi2_phase1, chi2_final;
    double trace_lambda[MAX_TRACE], trace_chi2[MAX_TRACE];
    int32_t sel;            // index of the committed estimate buffer (0/1)
    int32_t phase;          // 0 / 1
    int32_t max_iter;       // outer iterations allowed in this phase
    int32_t phase_iter;     // outer iterations completed in this phase
    int32_t trial_q;        // damped solves tried in the current outer iteration
    int32_t mode;           // MODE_* of the next unit; 0 = phase finished
    int32_t done;           // phase finished (max_iter reached or Terminate)
    int32_t solver_failed;  // direct solver: Cholesky hit a non-positive pivot in this unit
    int32_t pcg_iter;       // iterations of the last solve
    int32_t pcg_total;      // over all solves
    int32_t gauss_newton;   // trust_region == 1
    int32_t status;         // VISFS_BA_* (abort reasons)
    int32_t n_outliers;
    int32_t n_trace;
    int32_t iterations_run[2], trials_run[2];
    int32_t pcg_max;        // most PCG iterations any solve of this call needed
    int32_t pcg_timeout;    // a persistent-PCG workgroup gave up waiting (never expected; surfaces as ERR_DEVICE)
    int32_t n_active[4];    // units that did work: 0 linearise, 1 trial, 2 (unused), 3 back-substitution
    // speculative linearisation (k_linearize / k_odo_linearize with spec = 1 run beside the LM decision of the same launch):
    int32_t lin_sel;        // which of DeviceGraph::lin[2] holds the linearisation at the committed estimate
    int32_t spec_go;        // snapshot by k_backsub: this unit produced a trial state worth linearising
    int32_t spec_src;       // ... the estimate buffer holding it (sel ^ 1 at that time)
    int32_t spec_dst;       // ... the linearisation buffer to fill (lin_sel ^ 1 at that time)
    int32_t lin_b_pending;  // fused speculative unit: the accepted trial's landmark-major linearisation is complete (k_backsub<LINA>), its pose-major
                            // sums (hpp_part) are still to come — the role-B workgroups of the next k_schur_partial launch produce them
    int32_t ended;          // phase ends applied so far (0, 1, 2): k_eval / k_phase_end act only when the phase they close is done
    int32_t pcg_phase1;     // pcg_total when phase 1 ended
    int32_t n_edges_ok;     // stereo edges that can ever be active (not both ends fixed): the active set of phase 1
    // Optimizer/Framework=1 ([ceres-upstream] TrustRegionMinimizer + LevenbergMarquardtStrategy): lambda = 1 / tr_radius, ni = the
    // strategy's decrease factor, current_chi = 2 x cost
    double tr_radius, tr_x_norm;
    int32_t tr_invalid;     // consecutive invalid steps
    int32_t tr_reason;      // why the minimizer stopped: 1 max iterations, 2 gradient, 3 parameter, 4 function tolerance, 5 min radius, 6 invalid steps
    // Optimizer/Framework=1 with Optimizer/TrustRegion=1 ([ceres-upstream] DoglegStrategy, TRADITIONAL_DOGLEG): lambda = dl_mu, the multiplier of
    // the Gauss-Newton regularisation; the step of this unit is dl_A * v + dl_B * dn (v = g / m, dn = the regularised Gauss-Newton step)
    double dl_mu, dl_A, dl_B, dl_step_norm, dl_mcc;
    int32_t dogleg;
    uint32_t decide_epoch;  // tag of the last k_backsub launch that carried the LM decision; never reset (k_reset leaves it): stale
                            // hand-off words of an earlier launch or solve can then never match the tag a launch waits for
};

// The outputs of a linearisation that the Schur complement and the back-substitution consume.  Two sets: while the LM decision
// on a trial is taken, the trial state is already being linearised into the other set ("speculative linearise").
struct LinBuf {
    double* obs_w;              // [No]      rho' (0: inactive)
    double* obs_pcw;            // [No][4]   tile seed: Pc = R Pw + t (3) and the effective weight rho' / sigma^2 (0: no Hpl tile)
    double* pose_pcw;           // [n_pose_obs][4] the same seeds once more in POSE-MAJOR order (position in DeviceGraph::pose_obs): the pairs of a
                                //   Schur block (i, j) walk the landmarks common to poses i and j in ascending order, i.e. nearly consecutive
                                //   entries of both poses' lists — the gather's four seed loads per pair become coalesced (landmark-major they
                                //   hit 64 different cache lines per wave instruction; the kernel is bound by its loads, profiles/r02_tile_records_ab.log)
    double* Hll;                // [Nl][6]
    double* bl;                 // [Nl][3]
    double* hpp_part;           // [n_chunks][27] 21 upper + 6 b
    double* odo_blk;            // [Ne + 1][120] Aii(36) Ajj(36) Aij(36) bi(6) bj(6); slot Ne: the laser edges' aggregate
};

struct DeviceGraph {
    int32_t Np, Nl, No, Ne, Npf;
    int32_t n_chunks;       // pose-major chunks
    int32_t n_pose_obs;     // entries of pose_obs (observations of free poses)
    int32_t n_blk;          // stored S blocks (i <= j)
    int32_t n_sch;          // Schur chunks (<= sch_chunk co-observation pairs of one block each)
    int32_t sch_chunk;      // 64 x passes: pairs per chunk (a lane adds its pairs of the later passes serially)
    int32_t pcg_lds_minv;   // persistent PCG keeps all Minv blocks in LDS
    int32_t pcg_lds_srow;   // ... and its own block row of S
    int32_t pcg_max_row;    // longest block row of S (blocks)
    int32_t pcg_rows_per_wg; // block rows per PCG workgroup (1 up to 256 free poses)
    int32_t pcg_cu;         // 1: the reduced system is solved by ONE workgroup (k_pcg_cu: S in registers, no cross-workgroup hand-off)
    int32_t cu_T;           // ... its scalar rows rounded up to whole wavefronts: the row stride of S_rows
    int32_t pcg_lds_bytes;
    int32_t chol_np;        // padded order of the dense reduced camera matrix (direct solver)
    int32_t band_B;         // direct solver: block half-bandwidth of S (max j - i over the stored blocks) when the banded Cholesky
                            //   (k_band_chol) serves this window, -1: dense blocked Cholesky (k_chol_*)
    int32_t band_rows;      // ... block rows of the band resident in LDS at once (== Npf: the whole factor stays in LDS)
    int32_t band_lds_bytes; // ... dynamic LDS of k_band_chol
    int32_t n_lin_a;        // workgroups of the landmark-major role
    int32_t n_edges_ok;     // stereo edges whose two ends are not both fixed (the active set before the outlier pass)
    int32_t group;          // lanes per landmark (4/8/16/32/64)
    // Schur complement by RUNS OF LANDMARKS (k_schur_runs, round 4): n_runs > 0 selects it, 0 keeps the pair-list gather (k_schur_partial)
    int32_t n_runs;         // workgroups: run r covers landmarks [r * run_lr * run_m, ...), run_m sub-batches of run_lr landmarks each
    int32_t run_lr, run_m;
    int32_t run_cap;        // tile slots in LDS (>= the observations of any sub-batch)
    int32_t run_wmax;       // widest pose span of a run (<= RUN_MAX_W)
    int32_t run_lds_bytes;
    double fx, fy, cx, cy, bf;
    double inv_pixel_var, inv_odo_cov, huber_delta;
    // Optimizer/Framework=1 (Ceres branch, Optimizer.cpp:366-593): the residual is info * e with info = I / var, so the objective carries
    // 1 / var^2 (inv_pixel_var, inv_laser_cov hold the squares then) while the outlier test of :529-540 keeps e . (info e)
    int32_t ceres;
    int32_t dogleg;         // ... with its DOGLEG trust-region strategy (Optimizer/TrustRegion=1, Optimizer.cpp:515-519)
    double inv_pixel_var_out;
    // laser occupied-space edges (Optimizer.cpp:224-258): Nz unary edges on pose `laser_pose`; aggregated into slot Ne of odo_blk
    int32_t Nz, laser_pose;
    double inv_laser_cov;
    double Tcr[12];
    GridView grid;              // cost points into the static section
    const double* laser_xyz;    // [Nz][3]

    // ---- static graph ----
    const double* pose0;        // [Np][8]  initial Tcw (tx ty tz qx qy qz qw pad)
    const double* pt0;          // [Nl][3]
    const int32_t* pose_free;   // [Np]  free index or -1
    const int32_t* free_pose;   // [Npf] pose index
    const uint8_t* pt_fixed;    // [Nl]
    const int32_t* obs_pose;    // [No]
    const int32_t* obs_pt;      // [No]
    const int32_t* obs_ppos;    // [No] position of the observation in pose_obs (pose-major order), -1 for observations of fixed poses
    const double* obs_uvr;      // [No][3] (u_l, v_l, u_r) as Optimizer.cpp:187-188 forms them; window layer: written by k_index_count from obs_uvd
    const float* obs_uvd;       // [No][3] window layer only: the key-point (u, v) and the depth as localOptimize receives them (FeatureBA: floats) — 12
                                //   instead of 24 bytes per observation across the host's memory bus and PCIe; nullptr: obs_uvr came from the host
    double stereo_baseline;     // ... the baseline (getBaseLine(), a float) the disparity is formed with
    const uint8_t* obs_ok;      // [No] !(pose fixed && point fixed)
    const int32_t* lm_ptr;      // [Nl+1]
    const int32_t* chunk_pose;  // [n_chunks] free pose index
    const int32_t* chunk_ptr;   // [n_chunks+1] into pose_obs
    const int32_t* pose_obs;    // [sum] observation ids, pose-major (free poses only)
    const int32_t* pose_chunk_ptr; // [Npf+1] chunks of each free pose
    const int32_t* odo_i;       // [Ne] pose index of vertex 0
    const int32_t* odo_j;       // [Ne]
    const double* odo_tq;       // [Ne][7]
    const int32_t* pose_odo_ptr;// [Npf+1]
    const int32_t* pose_odo;    // [..] edge*2 + role (0: vertex 0 → Aii, 1: vertex 1 → Ajj)
    const int32_t* blk_i;       // [n_blk] free pose index (row)
    const int32_t* blk_j;       // [n_blk] (col), j >= i
    const int32_t* blk_ptr;     // [n_blk+1] into blk_pairs
    int32_t* pose_lm;           // [n_pose_obs] landmark of each entry of pose_obs (ascending within a pose); built on the device at upload
    struct PoseRec { int32_t k, l_ok; double u, v, ur; };   // observation id, landmark (bit 31: the edge is NOT usable: both ends fixed), the measurement
    PoseRec* pose_rec;          // [n_pose_obs] pose-major copy of what the pose-major role of the linearisation reads per observation (static: written by
                                // k_index_scatter): one 32-byte record instead of a chain pose_obs -> obs_pt / obs_ok / obs_uvr
    int4* blk_pairs;            // (tile of pose i, tile of pose j, landmark, 0) — tiles as pose-major positions: co-observations of one landmark, in landmark order
                                // per block; built on the device at upload (k_build_pairs) from the pose-major observation lists
    const int32_t* blk_chunk_ptr; // [n_blk+1] Schur chunks of each block
    const int4* sch_desc;       // [n_sch] (first pair, last pair + 1, pose index of i, pose index of j): ONE load gives a wave all it needs
    const int4* blk_desc;       // [n_blk][2]: (first Schur chunk, last + 1, first odometry entry, last + 1) — entries of blk_odo for an
                                //   off-diagonal block, of pose_odo for a diagonal one — and (i, j, first pose-major chunk of i, last + 1).
                                //   n_runs > 0: the first two are (first run | number of runs << 20, pose index of i | pose index of j << 16):
                                //   the runs whose pose span can hold this block
    const int32_t* run_k0;      // [n_runs * run_m + 1] first observation of every sub-batch of landmarks (the last entry: No)
    const int4* run_desc;       // [n_runs] (lowest pose index of the run's observations, pose span W, first slot of its W (W + 1) / 2 block
                                //   partials in sch_part, 0)
    const int32_t* blk_odo_ptr; // [n_blk+1]
    const int32_t* blk_odo;     // [..] edge*2 + transposed
    const int32_t* row_ptr;     // [Npf+1] adjacency of the block rows of S (for the mat-vec)
    const int32_t* row_col;     // [..] column block
    const int32_t* row_blk;     // [..] stored block id * 2 + transposed
    const int32_t* blk_slot;    // [n_blk] k_pcg_cu windows: position of stored block (i, j) in block row i's list | position in block row j's list << 8
    const int32_t* band_code;   // [Npf][band_B + 1] stored block id of S(I - d, I) (the lower block (I, I - d) is its transpose), -1: no such block
    const int32_t* pcg1_code;   // [Npf][Npf] stored block id * 2 + transposed of S(i, a), -1: no block — only for <= 64 free poses with PCG
                                //   (k_pcg1: one wavefront per block row); nullptr otherwise

    // ---- estimates ----
    double* pose[2];            // [Np][8]
    double* pt[2];              // [Nl][3]
    uint8_t* obs_level;         // [No] 0 active, 1 outlier (level 1)
    uint8_t* obs_outlier;       // [No] 1 iff moved to level 1 at Optimizer.cpp:285-286
    double* obs_chi2_out;       // [No] e^T Omega e at Optimizer.cpp:270

    // ---- linearisation products ----
    double* obs_err;            // [No][3]   (written only when debug != 0)
    double* obs_chi2;           // [No] (written under `debug` only: the stage hooks)
    double* W;                  // [No][18]  Hpl tiles, 6x3 row-major — written only for the stage hooks (debug)
    LinBuf lin[2];              // rho' weights, tile seeds, Hll, b_l, Hpp partials, odometry blocks: two sets (LmState::lin_sel)
    long long lin_stride;       // bytes from a member of lin[0] to the same member of lin[1]
    double* Hpp;                // [Npf][36]
    double* bp;                 // [Npf][6]
    int32_t* pose_pin;          // [Npf] 1: pose has no active edge (outside g2o's active set)
    double* lin_part;           // [n_lin_a + 1][2]  (robust chi2, max |Hll diag|) per workgroup; last = odometry

    // ---- per trial ----
    double* S;                  // [n_blk][36]
    double* S_rows;             // k_pcg_cu windows: S once more, BY SCALAR ROW — entry c of the k-th block of scalar row t at [(6 k + c) cu_T + t]: a
                                // wavefront's load of one entry is 512 contiguous bytes (k_schur_finalize writes both; unused slots stay zero from the upload)
    double* bs;                 // [Npf][6]
    double* Minv;               // [Npf][36]
    double* x;                  // [Npf][6]   pose increment
    double* sch_part;           // [n_sch][42] per-chunk partial sums (36 block entries + 6 of b_s)
    unsigned long long* granules; // [2][2*6Npf] {epoch:32 | half of a double:32} hand-off words of the persistent PCG
    uint32_t* fin_flag;         // [n_blk] fused finalisation + PCG launch (k_pcg1<FIN>): block b's S / b_s / Minv are complete when this holds the
                                //   unit's tag (damped solves of this optimise call so far + 1); zeroed by k_reset
    int32_t fin_pcg;            // 1: k_schur_finalize rides as the prologue of the k_pcg1 launch (one launch less per damped solve)
    uint32_t* fin_cnt;          // [n_blk] fin_arrive: arrivals at block b in the current launch of k_schur_partial (zero between launches)
    const int32_t* fin_exp;     // [n_blk] ... how many to expect: gather chunks of b | pose-major chunks of its pose << 16 (diagonal blocks)
    const int32_t* sch_blk;     // [n_sch] stored block of a gather chunk
    const int32_t* diag_blk;    // [Npf]   stored block id of (a, a)
    int32_t fin_arrive;         // 1: the LAST wavefront of k_schur_partial to arrive at a block finalises it (k_schur_finalize is not launched)
    double* dxl;                // [Nl][3]    landmark increment
    double* trial_part;         // [n_lin_a + 1][2]  (robust chi2 at trial state, scale contribution)
    // Optimizer/Framework=1: Jacobi scaling squared, fixed at iteration zero (k_ceres_lin_finalize): the damping of variable i is
    // lambda * clamp(H_ii s2_i, 1e-6, 1e32) / s2_i instead of lambda (damp_of)
    double* aux_part;           // [n_lin_a + 1] per-workgroup partials of the Ceres flavour: ||x||^2 shares (k_linearize), ||step||^2 shares (k_backsub)
    double* s2l;                // [Nl][3]
    double* s2p;                // [Npf][6]
    double* dl_part;            // [n_lin_a + 1][4] dogleg pass A per workgroup: ||J v||^2, ||g_s||^2, ||gn_s||^2, g_s . gn_s shares
    unsigned long long* trial_gran; // [n_lin_a + 1][4] the same two sums as {epoch:32 | half:32} hand-off words (k_backsub with the LM decision on board)
    double* chol_f;             // [chol_np][chol_np] the Cholesky factor L (direct solver), separate from the matrix being updated
    double* dense;              // [chol_np][chol_np] scratch of the direct solver (n = 6 Npf padded to a multiple of 32)
    double* chol_y;             // [chol_np]
    double* chol_linv;          // [2][32][32] inverse of the diagonal block of panel p in half p & 1
    double* band_L;             // [Npf][band_B + 1][36] the banded factor when it does not stay in LDS: block (I, I - d) at [I][d], d = 0: the packed Cholesky factor of the pivot block

    LmState* st;
    unsigned long long* stamps;  // [128] diagnostic build (-DVISFS_BA_STAMPS) only: real-time stamps of one PCG workgroup
    int32_t stamp_wg;
    int32_t debug;
    int32_t fault_pcg;      // test hook (VISFS_BA_FAULT_PCG_TIMEOUT=1 at upload): block row 0 of the persistent PCG withholds its first hand-off and every wait
                            // gives up after 2^10 instead of 2^22 polls — the path a GPU kept busy by another process takes (LmState::pcg_timeout)
};

// A wave-uniform pointer, pinned to scalar registers (the batched kernels read their graph from HBM with vector loads; without
// this the six pointers of a LinBuf stay live in VGPRs across the whole kernel).
template <class T>
#if defined(__HIPCC__)
__host__ __device__
#endif
inline T* sgpr_ptr(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
#else
    return p;
#endif
}

// Linearisation set k of a graph.  The two sets have the same internal layout at a constant byte distance (lin_stride), so the
// selection is one multiply-add per pointer on set 0 — selecting between two loaded pointers would keep both sets' pointers
// live (+20 VGPRs in the batched kernels, whose graph lives in HBM, cost k_linearize a wave of occupancy).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline LinBuf lin_of(const DeviceGraph& g, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    k = __builtin_amdgcn_readfirstlane(k);      // wave-uniform by construction
#endif
    const long long off = (long long)k * g.lin_stride;
    LinBuf L;
    L.obs_w = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].obs_w) + off));
    L.obs_pcw = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].obs_pcw) + off));
    L.pose_pcw = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].pose_pcw) + off));
    L.Hll = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].Hll) + off));
    L.bl = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].bl) + off));
    L.hpp_part = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].hpp_part) + off));
    L.odo_blk = sgpr_ptr(reinterpret_cast<double*>(reinterpret_cast<char*>(g.lin[0].odo_blk) + off));
    return L;
}

}  // namespace visfs_ba
