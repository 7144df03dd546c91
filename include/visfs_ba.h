/*
 * visfs_ba.h — C ABI of the MI355X sliding-window bundle-adjustment backend.
 *
 * This is the drop-in boundary for VISFS's `Optimizer::localOptimize`
 * (reference: corelib/include/Optimizer/Optimizer.h:29-73, implementation
 * corelib/src/Optimizer/Optimizer.cpp:58-364, g2o branch).  The reference has no
 * FFI of its own; every entry point below cites the reference code it replaces.
 * Plain pointers and sizes only: no C++/Eigen/OpenCV/torch types cross this line.
 *
 * Two layers are exported:
 *   1. the WINDOW layer  — same inputs/outputs as `localOptimize`, flattened
 *      (`visfs_ba_solve_window`, `visfs_ba_solve_batch`);
 *   2. the GRAPH layer   — the factor graph the reference builds at
 *      Optimizer.cpp:100-223 as flat arrays, kept resident in HBM
 *      (`visfs_ba_graph_*`, `visfs_ba_optimize`), plus stage hooks used by the
 *      parity tests to compare every intermediate against the CPU oracle.
 *
 * Threading: a handle is NOT thread-safe and owns one HIP stream, matching the
 * reference's single-caller contract (Estimator thread only, Estimator.cpp:254).
 * No exceptions cross the ABI.  All floating point is IEEE fp64 unless stated.
 */
#ifndef VISFS_BA_H
#define VISFS_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VISFS_BA_ABI_VERSION 8

/* ---- status codes ------------------------------------------------------- */
/* The reference signals failure by returning an EMPTY pose map
 * (Optimizer.cpp:143-147, 272-280, 315-318, 330-334).  The C++ shim maps every
 * status != VISFS_BA_OK (except PASSTHROUGH) to that empty map. */
enum {
    VISFS_BA_OK = 0,
    VISFS_BA_PASSTHROUGH = 1,      /* poses.size()==1 || iterations<=0 → input poses returned (Optimizer.cpp:360-361) */
    VISFS_BA_ERR_TOO_FEW_POSES = 2,/* "should be called at least with 1 pose" (Optimizer.cpp:362-364), or first id == 0 (:74) */
    VISFS_BA_ERR_NAN_CHI2 = 3,     /* Optimizer.cpp:272-275 */
    VISFS_BA_ERR_HUGE_CHI2_1 = 4,  /* Optimizer.cpp:277-280 */
    VISFS_BA_ERR_HUGE_CHI2_2 = 5,  /* Optimizer.cpp:315-318 */
    VISFS_BA_ERR_BAD_ARGUMENT = 6,
    VISFS_BA_ERR_UNSUPPORTED = 7,  /* an unknown Optimizer/Framework, a size beyond a kernel's limits, ... */
    VISFS_BA_ERR_DEVICE = 8,       /* HIP runtime error / no MI355X present */
    VISFS_BA_ERR_NOT_LOADED = 9    /* no graph resident in the handle */
};

/* ---- parameters: the eight Optimizer keys (Parameters.h:184-191, read at Optimizer.cpp:37-54) */
typedef struct visfs_ba_params {
    int32_t framework;            /* Optimizer/Framework: 0 = the g2o branch (Optimizer.cpp:72-364), 1 = the Ceres branch (:366-593;
                                   * trust_region picks its strategy: 0 LEVENBERG_MARQUARDT, 1 DOGLEG (:515-519); solver is ignored —
                                   * every linear_solver_type the branch selects is an exact dense solve) */
    int32_t solver;               /* Optimizer/Solver: 0 csparse, 1 cholmod, 3 eigen → direct Cholesky of S; 2 → block-Jacobi PCG */
    int32_t trust_region;         /* Optimizer/TrustRegion: 0 Levenberg, 1 GaussNewton (framework 1: 0 LM, 1 DOGLEG) */
    int32_t iterations;           /* Optimizer/Iterations (run as iterations/2 + iterations/2) */
    double  pixel_variance;       /* Optimizer/PixelVariance      (default 1.5)  */
    double  odometry_covariance;  /* Optimizer/OdometryCovariance (default 5e-5) */
    double  laser_covariance;     /* Optimizer/LaserCovariance    (default 0.1, unused: laser factor out of scope) */
    double  robust_kernel_delta;  /* Optimizer/RobustKernelDelta  (default 8.0; <=0 disables Huber and phase 2) */
} visfs_ba_params;

/* Fills the reference defaults (Parameters.h:184-191). */
void visfs_ba_default_params(visfs_ba_params* p);

/* ---- WINDOW layer: the arguments of localOptimize, flattened -------------- */
/* 3x4 transforms are ROW-MAJOR [r00 r01 r02 tx  r10 r11 r12 ty  r20 r21 r22 tz]
 * (the top three rows of Eigen::Isometry3d::matrix()). */
/* Map::Grid2D as EdgeOccupiedObservation reads it through GridArrayAdapter (TypeOccupiedSpace2D.h:22-48):
 * limits (MapLimits.h:25-37) and getCorrespondenceCost(Array2i(x, y)) (Grid2d.h:33-36) for every cell, i.e. the
 * float the reference obtains from its uint16 cells via the value table (unknown cells carry the table's entry 0). */
typedef struct visfs_ba_grid {
    double  resolution;           /* limits().resolution() */
    double  max_x, max_y;         /* limits().max() */
    int32_t num_x_cells, num_y_cells;   /* limits().cellLimits() */
    const float* correspondence_cost;   /* [num_y_cells][num_x_cells], flat index num_x_cells * y + x (Grid2d.h:93-95) */
} visfs_ba_grid;

typedef struct visfs_ba_window {
    uint64_t root_id;             /* _rootId: pose fixed iff id == root_id (Optimizer.cpp:111) */

    int32_t  n_poses;             /* _poses, in std::map (ascending id) order */
    const uint64_t* pose_ids;     /* [n_poses] */
    const double*   pose_Twr;     /* [n_poses][12] robot pose in world */

    int32_t  n_links;             /* _links, in std::map order */
    const uint64_t* link_from;    /* [n_links] */
    const uint64_t* link_to;      /* [n_links] */
    const double*   link_T;       /* [n_links][12] T_r1r2 */

    int32_t  n_cameras;           /* _cameraModels.size(); baseline is used only if > 1 (Optimizer.cpp:181-183) */
    double   fx, fy, cx, cy;      /* cameraModels.front()->eigenKdouble() (Optimizer.cpp:176,191-194) */
    float    baseline;            /* getBaseLine() — float in the reference (PinholeModel.cpp:75-77) */
    double   Trc[12];             /* getTansformImageToRobot() (GeometricCamera.h:15-19) */

    int32_t  n_points;            /* _points3D, ascending feature id */
    const uint64_t* point_ids;    /* [n_points] */
    double*         point_xyz;    /* [n_points][3]  IN/OUT (Optimizer.cpp:343-358) */
    const uint8_t*  point_fixed;  /* [n_points] fixSymbol */

    int32_t  n_refs;              /* _wordReferences leaves, feature-major then pose-major (nested std::map order) */
    const uint64_t* ref_feature;  /* [n_refs] */
    const uint64_t* ref_pose;     /* [n_refs] */
    const float*    ref_u;        /* [n_refs] kpt.pt.x */
    const float*    ref_v;        /* [n_refs] kpt.pt.y */
    const float*    ref_depth;    /* [n_refs] FeatureBA::depth */

    /* laser occupied-space factor (Optimizer.cpp:224-258): active iff n_laser_points > 0 AND grid != NULL
     * (the reference's `!_pointClouds.empty() && _submap != nullptr`). */
    int32_t  n_laser_points;      /* all points of all _pointClouds, concatenated in iteration order */
    const double* laser_xyz;      /* [n_laser_points][3] RangefinderPoint::position (robot frame of the newest pose) */
    const visfs_ba_grid* grid;    /* _submap->getGrid() */
} visfs_ba_window;

typedef struct visfs_ba_result {
    int32_t  status;              /* VISFS_BA_* */
    int32_t  n_poses_out;         /* 0 on failure (the reference's empty map) */
    uint64_t* pose_ids_out;       /* caller-allocated [n_poses] */
    double*   pose_Twr_out;       /* caller-allocated [n_poses][12] */
    int32_t  outlier_capacity;    /* size of the two arrays below (>= n_refs is always enough) */
    int32_t  n_outliers;          /* appended pairs (featureId, poseId), reference order (Optimizer.cpp:284-302) */
    uint64_t* outlier_feature;    /* caller-allocated */
    uint64_t* outlier_pose;       /* caller-allocated */
    int32_t  iterations_run[2];   /* outer LM iterations executed in phase 1 / phase 2 */
    double   chi2_initial;        /* activeRobustChi2 before the first iteration */
    double   chi2_phase1;         /* Optimizer.cpp:271 */
    double   chi2_final;          /* Optimizer.cpp:315 */
    int32_t  warn_mono_skipped;   /* observations that would take the reference's uninitialised mono branch (Optimizer.cpp:197-210): skipped */
    int32_t  solver_fallback;     /* ABI 7: 1 = Optimizer/Solver=2's persistent PCG could not keep its workgroups resident (another process on the
                                   * GPU) and this window was solved again, in the same call, on the direct solver — see visfs_ba_stats */
} visfs_ba_result;

/* ---- GRAPH layer: the factor graph of Optimizer.cpp:100-223 as flat arrays ---- */
/* Pose state is the reference's CameraPose (OptimizeTypeDefine.h:16-86):
 * Tcw as [tx ty tz qx qy qz qw], w >= 0, unit norm.  Observations are sorted by
 * (point, pose) — the insertion order of the reference's edges. */
typedef struct visfs_ba_graph {
    int32_t n_poses, n_points, n_obs, n_odo;
    const double*  pose_tq;       /* [n_poses][7] */
    const uint8_t* pose_fixed;    /* [n_poses] */
    const double*  point_xyz;     /* [n_points][3] */
    const uint8_t* point_fixed;   /* [n_points] */
    const int32_t* obs_point;     /* [n_obs] non-decreasing */
    const int32_t* obs_pose;      /* [n_obs] */
    const double*  obs_uvr;       /* [n_obs][3] (u_l, v_l, u_r) as built at Optimizer.cpp:187-188 */
    const int32_t* odo_from;      /* [n_odo] vertex(0) (Optimizer.cpp:136-139) */
    const int32_t* odo_to;        /* [n_odo] vertex(1) */
    const double*  odo_tq;        /* [n_odo][7] T_c1c2 = Trc^-1 T_r1r2 Trc as SE3Quat (Optimizer.cpp:131-140) */
    double fx, fy, cx, cy, bf;    /* EdgeStereo intrinsics (Optimizer.cpp:191-195) */
    /* EdgeOccupiedObservation edges (Optimizer.cpp:224-258): unary on pose `laser_pose` (the newest), information
     * 1 / laserCovariance, no robust kernel; inactive when that pose is fixed (allVerticesFixed). */
    int32_t n_laser;
    int32_t laser_pose;           /* index of _poses.rbegin() */
    const double* laser_xyz;      /* [n_laser][3] */
    const visfs_ba_grid* grid;    /* NULL iff n_laser == 0 */
    double Tcr[12];               /* transformRobotToImage_ = getTansformImageToRobot().inverse() (TypeOccupiedSpace2D.h:81-82) */
} visfs_ba_graph;

/* Per-solve statistics of the two optimise phases (Optimizer.cpp:261-318).
 * Optimizer/Framework=1 (the Ceres branch, one ceres::Solve): iterations_run[0] = passes of the minimizer loop (successful, unsuccessful
 * and invalid steps alike, as Solver::Summary::iterations counts them), trials_run[0] = linear solves, [1] = 0; chi2_* = 2 x cost (the sum
 * of rho over the residual blocks), chi2_phase1 = chi2_final; trace_lambda = the trust-region radius after each iteration. */
#define VISFS_BA_MAX_TRACE 64
typedef struct visfs_ba_stats {
    int32_t status;
    int32_t iterations_run[2];
    int32_t trials_run[2];        /* damped solves incl. rejected ones */
    int32_t pcg_iterations;       /* total over all solves (solver == 2) */
    int32_t n_outliers;
    double  chi2_initial, chi2_phase1, chi2_final;
    int32_t n_trace;              /* entries used below (outer iterations, both phases) */
    double  trace_lambda[VISFS_BA_MAX_TRACE];   /* lambda of the accepted (or last) trial */
    double  trace_chi2[VISFS_BA_MAX_TRACE];     /* robust chi2 after the iteration */
    /* ABI 3: what the two phases actually worked on (measurement: a phase-2 launch touches only the edges left at level 0) */
    int32_t n_active_edges[2];    /* stereo edges in the active set of phase 1 / phase 2 (level 0, not both ends fixed); [1] = [0] - n_outliers */
    int32_t pcg_iterations_phase[2];   /* PCG iterations of the solves of phase 1 / phase 2 (sum = pcg_iterations) */
    /* ABI 7: 1 = the persistent PCG's hand-off timed out (its grid could not stay resident: a GPU shared with another process) and the
     * solve was re-run from the uploaded estimates on the direct solver (k_band_chol / the dense blocked Cholesky) — the reference's
     * linear solver cannot fail for lack of residency (Optimizer.cpp:76-91).  The result is then the exact-solve one (pcg_iterations = 0). */
    int32_t solver_fallback;
    int32_t reserved;
} visfs_ba_stats;

typedef struct visfs_ba_handle visfs_ba_handle;

/* Replaces `Optimizer::Optimizer(const ParametersMap&)` (Optimizer.cpp:37-56).
 * device_index: HIP device ordinal.  Returns VISFS_BA_ERR_DEVICE when no gfx950
 * device can be opened — there is NO CPU fallback behind this ABI. */
int visfs_ba_create(const visfs_ba_params* params, int device_index, visfs_ba_handle** out);
void visfs_ba_destroy(visfs_ba_handle* h);
const char* visfs_ba_last_error(const visfs_ba_handle* h);
/* Why the calling thread's last visfs_ba_create failed ("" after a success): missing runtime or device, index out of range,
 * wrong architecture, or a stream / pinned-memory allocation failure — with the HIP error text. */
const char* visfs_ba_create_error(void);
int visfs_ba_abi_version(void);

/* What a handle is tuned for (ABI 8).  The reference has no counterpart: one VISFS::Optimizer::Optimizer serves one Estimator
 * (corelib/include/Optimizer/Optimizer.h:29-56), one window per call.  LATENCY (the default): a window on its own is solved as fast as it can
 * be — reduced systems of up to 64 free poses run the PCG as one wavefront per block row on as many compute units.  THROUGHPUT: for a handle
 * that is mostly given BATCHES (visfs_ba_solve_batch, visfs_ba_batch_*; BASELINE config 5), windows of up to 56 free poses run their PCG in ONE
 * workgroup each: +7 % at 8 resident 50-key-frame windows, +13 % at 16, -4 % for a lone window.  The choice belongs to the handle, never to the
 * size of a batch: a window's result is the same bytes whether it is solved alone or with others THROUGH THE SAME HANDLE; the two tunings agree
 * to rounding with identical iteration counts.  Applies to the uploads that follow (resident graphs keep what they were uploaded with).
 * The environment variable VISFS_BA_PCG_CU=0|1, when set, overrides every handle (diagnostics). */
#define VISFS_BA_TUNE_LATENCY    0
#define VISFS_BA_TUNE_THROUGHPUT 1
int visfs_ba_set_tuning(visfs_ba_handle* h, int32_t tuning);

/* Replaces one call of `Optimizer::localOptimize` (Optimizer.cpp:58-364): pack
 * (graph build :100-223), upload, both optimise phases on the GPU (:261-318),
 * download, write-back (:320-358).  Host pointers in, host pointers out. */
int visfs_ba_solve_window(visfs_ba_handle* h, const visfs_ba_window* w, visfs_ba_result* r);

/* BASELINE config 5: n independent windows solved concurrently on this handle's
 * device (one stream per in-flight window).  No reference counterpart — the
 * reference solves one window per frame. */
int visfs_ba_solve_batch(visfs_ba_handle* h, int32_t n, const visfs_ba_window* const* w, visfs_ba_result* const* r);

/* BASELINE config 5 inside ONE process: n independent windows over n_handles handles — normally one per GPU of the node.  The
 * windows are dealt in contiguous blocks, window i to handle i / ceil(n / n_handles) (the partition of SURVEY §8e and of
 * visfs_amd/dist.py's one-process-per-GPU form); every handle solves its block with visfs_ba_solve_batch on its own host thread.
 * No data-path exchange between devices: results land in the caller's buffers.  Returns the worst status of the blocks. */
int visfs_ba_solve_batch_sharded(visfs_ba_handle* const* handles, int32_t n_handles, int32_t n,
                                 const visfs_ba_window* const* w, visfs_ba_result* const* r);

/* Host-only graph build, exported so that CPU tests can check it without a GPU
 * (Optimizer.cpp:100-223: Twr→Tcw :104-109, link → T_c1c2 :131-140, float
 * disparity :187-188).  Caller allocates: pose_tq[n_poses*7], pose_fixed[n_poses],
 * point_used[n_points] (1 iff the point became a vertex, :158), obs_* sized n_refs,
 * odo_* sized n_links.  On return *g points into those buffers. */
int visfs_ba_pack_window(const visfs_ba_params* params, const visfs_ba_window* w,
                         double* pose_tq, uint8_t* pose_fixed, uint8_t* point_used,
                         int32_t* obs_point, int32_t* obs_pose, double* obs_uvr, int32_t* obs_ref,
                         int32_t* odo_from, int32_t* odo_to, double* odo_tq,
                         visfs_ba_graph* g, int32_t* n_mono_skipped);
/* Host-only write-back (Optimizer.cpp:320-329): Twr = Tcw^-1 * Trc^-1. */
void visfs_ba_unpack_pose(const double* tq, const double* Trc, double* Twr_out);

/* Upload a graph and build the device-side index structures (the analogue of
 * graph construction + g2o buildStructure).  The graph stays resident. */
int visfs_ba_graph_upload(visfs_ba_handle* h, const visfs_ba_graph* g);
/* Restore the uploaded initial estimates and re-activate all edges (device→device). */
int visfs_ba_graph_reset(visfs_ba_handle* h);
/* Both optimise phases + outlier marking on the resident graph
 * (Optimizer.cpp:261-318).  Inputs are already in HBM: this is the timed region
 * of bench.py. */
int visfs_ba_optimize(visfs_ba_handle* h, visfs_ba_stats* stats);
/* Any of the output pointers may be NULL.  obs_outlier[i] = 1 iff edge i was
 * moved to level 1 (Optimizer.cpp:285-286); obs_chi2 = e^T Omega e at :270. */
int visfs_ba_graph_download(visfs_ba_handle* h, double* pose_tq, double* point_xyz,
                            uint8_t* obs_outlier, double* obs_chi2);

/* ---- stage hooks (parity tests; each runs the production kernels) -------- */
enum {
    VISFS_BA_BUF_OBS_ERR = 0,     /* [n_obs][3]  e = z - pi(R Pw + t)          (OptimizeTypeDefine.h:121-126) */
    VISFS_BA_BUF_OBS_CHI2 = 1,    /* [n_obs]     e^T Omega e                    */
    VISFS_BA_BUF_OBS_WEIGHT = 2,  /* [n_obs]     Huber rho'(chi2), 0 if inactive */
    VISFS_BA_BUF_HPL = 3,         /* [n_obs][18] J_pose^T (rho' Omega) J_point, 6x3 row-major; 0 if an end is fixed */
    VISFS_BA_BUF_HLL = 4,         /* [n_points][6] xx xy xz yy yz zz            */
    VISFS_BA_BUF_BL = 5,          /* [n_points][3]                              */
    VISFS_BA_BUF_HPP = 6,         /* [6*npf][6*npf] dense row-major, free poses in index order (visual + odometry) */
    VISFS_BA_BUF_BP = 7,          /* [6*npf]                                    */
    VISFS_BA_BUF_S = 8,           /* [6*npf][6*npf] reduced camera matrix incl. lambda */
    VISFS_BA_BUF_BS = 9,          /* [6*npf]                                    */
    VISFS_BA_BUF_DX_POSE = 10,    /* [6*npf]                                    */
    VISFS_BA_BUF_DX_POINT = 11,   /* [n_points][3], 0 for fixed points          */
    VISFS_BA_BUF_POSE_TRIAL = 12, /* [n_poses][7]  state after the update       */
    VISFS_BA_BUF_POINT_TRIAL = 13 /* [n_points][3]                              */
};
/* number of free (non-fixed) poses of the resident graph */
int visfs_ba_graph_free_poses(visfs_ba_handle* h);
/* Linearise at the current estimate (K1,K2,K3,K4): fills ERR/CHI2/WEIGHT/HPL/HLL/BL/HPP/BP.
 * Returns the robust chi2 (activeRobustChi2) and the max |diag H| used for lambda init. */
int visfs_ba_stage_linearize(visfs_ba_handle* h, double* robust_chi2, double* max_diag);
/* One damped solve at the given lambda (K5,K6,K7,K8 + chi2 at the trial state).
 * solver follows the handle's params.  Does not commit the trial state. */
int visfs_ba_stage_trial(visfs_ba_handle* h, double lambda, double* trial_chi2, double* scale,
                         int32_t* pcg_iterations, int32_t* solver_ok);
/* Copies a stage buffer to host memory as fp64 in the layout documented above. */
int visfs_ba_stage_fetch(visfs_ba_handle* h, int32_t which, double* dst, size_t n_doubles);

/* Stepping a solve by hand (tools/soak_diverge.py): make the last trial state the estimate (discardTop); start a phase
 * (LinearSolverPCG::init(): the carried residual is forgotten); the outlier pass of Optimizer.cpp:283-303 on the estimate
 * (marks edges with chi2 > delta as level 1, visible through visfs_ba_graph_download's obs_outlier). */
int visfs_ba_stage_commit(visfs_ba_handle* h);
int visfs_ba_stage_begin_phase(visfs_ba_handle* h);
int visfs_ba_stage_mark_outliers(visfs_ba_handle* h);
/* Host-only hook (no device needed): ONE optimize(n_iter) phase of the LM / Gauss-Newton control (K9, the device-side state
 * machine's own functions compiled for the host) driven by scripted trial outcomes — trial t returns (temp_chi[t], scale[t] =
 * computeScale() without the +1e-3, ok[t]); the last entry repeats if the schedule asks for more.  Fills stats->trace_*,
 * iterations_run[0], trials_run[0], chi2_final (committed chi2); returns the trials consumed.  Checks the schedule of
 * [g2o-upstream] OptimizationAlgorithmLevenberg::solve, incl. its failure paths, against the CPU checker. */
int visfs_ba_hook_lm_script(int32_t gauss_newton, int32_t n_iter, double chi0, double max_diag0, int32_t n_trials,
                            const double* temp_chi, const double* scale, const int32_t* ok, visfs_ba_stats* stats);
/* Host-only hook: the control of the Ceres branch (Optimizer/Framework=1; [ceres-upstream] TrustRegionMinimizer + LevenbergMarquardtStrategy,
 * the device-side state machine's own functions compiled for the host) on scripted outcomes: iteration t's linear solve reports
 * (ok[t], model_cost_change[t], cand_cost[t], step_norm[t]); a step that is taken then reports (grad_max[t], x_norm[t]) of the new
 * linearisation; the last entry repeats.  Fills stats->trace_lambda (trust-region radius after each iteration), trace_chi2 (2 x cost),
 * iterations_run[0], chi2_final; returns the termination reason (1 max iterations, 2 gradient, 3 parameter, 4 function tolerance,
 * 5 minimum radius, 6 consecutive invalid steps). */
int visfs_ba_hook_ceres_script(int32_t max_iter, double cost0, double x_norm0, double grad_max0, int32_t n, const int32_t* ok,
                               const double* model_cost_change, const double* cand_cost, const double* step_norm,
                               const double* grad_max, const double* x_norm, visfs_ba_stats* stats);
/* The same with the DOGLEG strategy (Optimizer/TrustRegion=1; [ceres-upstream] DoglegStrategy's radius and mu rules): iteration t's step
 * had the scaled length dogleg_step_norm[t]; mu_trace[i] (VISFS_BA_MAX_TRACE doubles) = the regularisation mu after iteration i. */
int visfs_ba_hook_dogleg_script(int32_t max_iter, double cost0, double x_norm0, double grad_max0, int32_t n, const int32_t* ok,
                                const double* model_cost_change, const double* cand_cost, const double* step_norm,
                                const double* dogleg_step_norm, const double* grad_max, const double* x_norm, visfs_ba_stats* stats,
                                double* mu_trace);
/* Host-only hook: the point on the dogleg path (the function k_dogleg_mid calls, compiled for the host) from the inner products of the
 * scaled space: s1 = ||g_s||^2, s2 = ||gn_s||^2, s3 = g_s . gn_s, jv2 = ||J (g / m)||^2.  out = { A, B, step norm, model cost change }:
 * the step is A (g / m) + B dn in the unscaled variables. */
int visfs_ba_hook_dogleg_combine(double s1, double s2, double s3, double jv2, double radius, double mu, double out[4]);

/* ---- measurement hooks (bench.py) ------------------------------------------ */
/* Sizes of the resident graph and of the index structures built at upload. */
typedef struct visfs_ba_graph_info {
    int32_t n_poses, n_free_poses, n_points, n_obs, n_odo;
    int32_t n_blk;                /* stored 6x6 blocks of the reduced camera matrix (upper triangle incl. diagonal) */
    int64_t n_pairs;              /* co-observation pairs feeding the Schur gather */
    int32_t lanes_per_landmark;   /* wavefront sub-group size of the landmark-major kernels */
    int32_t n_schur_chunks;       /* wavefronts of the Schur gather (<= 64 pairs each) */
    int64_t device_bytes;         /* HBM footprint of the window */
    int32_t fused_path;           /* 1: the window runs on the fused single-workgroup kernel (small windows, opt-in with VISFS_BA_FUSED=1) */
    int32_t solver_kernel;        /* which kernel solves the reduced system (the symbol a kernel trace shows for the VISFS_BA_K_PCG / _DIRECT class):
                                   * 1 k_pcg1 (PCG, one wavefront per block row), 2 k_pcg (PCG, four waves per row), 3 k_pcg, several rows per
                                   * workgroup, 4 k_pcg_cu (PCG, one workgroup), 5 k_small_solve, 6 k_chol_* (blocked dense Cholesky),
                                   * 7 k_band_chol (block-banded Cholesky factorisation in one workgroup) */
    int32_t band_blocks;          /* direct solver: block half-bandwidth of S when k_band_chol serves the window, -1 otherwise */
    int32_t graph_replayed;       /* 1: the LAST visfs_ba_optimize of this resident graph ran as a hipGraph replay of its launch sequence (only
                                   * re-optimisations of the same resident graph are ever replayed; a per-frame visfs_ba_solve_window never is) */
    int32_t unit_form;            /* how a damped solve is launched: 0 the gated unit (k_linearize when a trial was accepted ... k_backsub + decision),
                                   * 1 the speculative unit in two launches (k_backsub, then k_linearize of the trial beside the decision),
                                   * 2 the fused speculative unit (k_backsub also carries the decision and the landmark-major half of the
                                   *   trial's linearisation; the pose-major half rides behind the next k_schur_partial): what the kernel
                                   *   classes of the profile hooks contain depends on it */
    int32_t schur_runs;           /* ABI 7: > 0: the Schur complement is formed by k_schur_runs, that many workgroups each owning a run of
                                   * schur_run_landmarks consecutive landmarks (tiles staged in LDS); 0: the pair-list gather (k_schur_partial,
                                   * n_schur_chunks wavefronts) */
    int32_t schur_run_landmarks;
} visfs_ba_graph_info;
/* GRAPH layer for a batch of independent windows (BASELINE config 5): n graphs resident side by side; one optimise call runs
 * them through ONE sequence of launches (blockIdx.y = window, each window gated by its own LM state).  Needs
 * Optimizer/Solver = 2 or reduced systems <= 64 x 64.  stats: caller-allocated [n] or NULL. */
int visfs_ba_batch_upload(visfs_ba_handle* h, int32_t n, const visfs_ba_graph* const* graphs);
int visfs_ba_batch_reset(visfs_ba_handle* h);
int visfs_ba_batch_optimize(visfs_ba_handle* h, visfs_ba_stats* stats);
int visfs_ba_batch_download(visfs_ba_handle* h, int32_t index, double* pose_tq, double* point_xyz, uint8_t* obs_outlier, double* obs_chi2);

int visfs_ba_graph_describe(visfs_ba_handle* h, visfs_ba_graph_info* out);

enum {
    VISFS_BA_K_LINEARIZE = 0, VISFS_BA_K_LIN_FINALIZE = 1, VISFS_BA_K_SCHUR = 2, VISFS_BA_K_SCHUR_FINALIZE = 3,
    VISFS_BA_K_PCG = 4, VISFS_BA_K_DIRECT = 5, VISFS_BA_K_BACKSUB = 6, VISFS_BA_K_DECIDE = 7,
    VISFS_BA_K_PHASE_END = 8, VISFS_BA_K_RESET = 9, VISFS_BA_K_SMALL = 10, VISFS_BA_K_COUNT = 11
};
typedef struct visfs_ba_profile {
    double  total_ms[VISFS_BA_K_COUNT];        /* sum of hipEventElapsedTime over all launches of the class */
    int64_t launches[VISFS_BA_K_COUNT];
    double  active_ms[VISFS_BA_K_COUNT];       /* the `active_launches` longest launches (gated-off launches return at once) */
    int64_t active_launches[VISFS_BA_K_COUNT]; /* launches that did work, counted on the device */
    double  null_pair_ms;                      /* median elapsed time of an EMPTY event pair on the stream: what the event mechanism
                                                  itself adds to every duration above (rocprofv3 kernel durations do not carry it) */
} visfs_ba_profile;
/* mask: bit k enables hipEvent pairs around every launch of kernel class k on the handle's own stream. */
int visfs_ba_profile_enable(visfs_ba_handle* h, uint32_t mask);
/* Returns and clears the accumulated profile. */
int visfs_ba_profile_read(visfs_ba_handle* h, visfs_ba_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* VISFS_BA_H */
