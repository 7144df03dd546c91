/*
 * visfs_ba_oracle.h — CPU oracle for the sliding-window BA hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product
 * (visfs_amd/csrc → libvisfs_ba_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (supersaiyajinggod/VISFS) holds no test, golden
 * vector or fixture for Optimizer::localOptimize and its
 * arithmetic lives in g2o, an un-vendored, un-pinned dependency (API usage bounds it
 * to releases 20201223_git … pre-2023) that cannot be built in this image (no
 * Eigen/g2o/OpenCV).  This file restates (a) VISFS's own vertex/edge arithmetic,
 * citing file:line, and (b) g2o's published Levenberg/Schur/PCG algorithm
 * ([g2o-upstream] marks: g2o/core/optimization_algorithm_levenberg.cpp,
 * block_solver.hpp, base_binary_edge.hpp, robust_kernel_impl.cpp,
 * solvers/pcg/linear_solver_pcg.hpp).  It is pinned only by the known-answer
 * vectors, finite-difference checks and fixed-point tests under tests/; the single
 * piece checked against the reference's own test vectors is the grid addressing of
 * the laser factor (tests/golden/ref_map2d_cell_index.json).
 */
#ifndef VISFS_BA_ORACLE_H
#define VISFS_BA_ORACLE_H

#include "../include/visfs_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- VISFS-owned arithmetic (K1,K2,K3,K8) -------------------------------- */
/* CameraPose(R,t): Eigen::Quaterniond(R) + normalizeRotation (OptimizeTypeDefine.h:30-41). R row-major. */
void oracle_pose_from_Rt(const double R[9], const double t[3], double tq[7]);
void oracle_pose_to_Rt(const double tq[7], double R[9], double t[3]);
/* CameraPose::update (OptimizeTypeDefine.cpp:7-14) with deltaQ (Math.h:277-287). */
void oracle_pose_update(double tq[7], const double delta[6]);
/* EdgeStereo::computeError + linearizeOplus (OptimizeTypeDefine.h:121-187).
 * intr = {fx,fy,cx,cy,bf}; Jpoint 3x3 row-major, Jpose 3x6 row-major. */
void oracle_stereo_edge(const double tq[7], const double pw[3], const double uvr[3],
                        const double intr[5], double e[3], double Jpoint[9], double Jpose[18]);
/* EdgePoseConstraint::computeError + linearizeOplus (OptimizeTypeDefine.cpp:35-88). Ji,Jj 6x6 row-major. */
void oracle_odo_edge(const double tq1[7], const double tq2[7], const double meas_tq[7],
                     double e[6], double Ji[36], double Jj[36]);
/* EdgeOccupiedObservation::computeError / linearizeOplus (TypeOccupiedSpace2D.h:126-179) incl. the reference's
 * autodiff parameter aliasing (see the .c file); [ceres-upstream] bicubic interpolation.  J is 1x6, either may be NULL. */
void oracle_laser_edge(const double tq[7], const double Tcr[12], const double P[3], const visfs_ba_grid* g, double* e, double J[6]);
void oracle_bicubic(const visfs_ba_grid* g, double r, double c, double* f, double* dfdr, double* dfdc);
/* RobustKernelHuber::robustify [g2o-upstream]: rho[0]=rho(e2), rho[1]=rho'(e2). */
void oracle_huber(double e2, double delta, double rho[2]);

/* ---- graph build / write-back (Optimizer.cpp:100-223, 320-358) ----------- */
/* Same contract as visfs_ba_pack_window in include/visfs_ba.h. */
int oracle_pack_window(const visfs_ba_params* params, const visfs_ba_window* w,
                       double* pose_tq, uint8_t* pose_fixed, uint8_t* point_used,
                       int32_t* obs_point, int32_t* obs_pose, double* obs_uvr, int32_t* obs_ref,
                       int32_t* odo_from, int32_t* odo_to, double* odo_tq,
                       visfs_ba_graph* g, int32_t* n_mono_skipped);
void oracle_unpack_pose(const double* tq, const double* Trc, double* Twr_out);

/* ---- the solver ----------------------------------------------------------- */
typedef struct oracle_sys oracle_sys;
/* num_threads: 1 = scalar port (g2o's default build has OpenMP off); >1 uses OpenMP
 * over edges / landmarks when the oracle is compiled with -fopenmp. */
oracle_sys* oracle_sys_create(const visfs_ba_params* params, const visfs_ba_graph* g, int num_threads);
void oracle_sys_destroy(oracle_sys* s);
int oracle_sys_free_poses(const oracle_sys* s);
/* computeActiveErrors + buildSystem at the current estimate. */
void oracle_sys_linearize(oracle_sys* s, double* robust_chi2, double* max_diag);
/* setLambda + Schur solve + update + computeActiveErrors at the trial state; trial is NOT committed. */
void oracle_sys_trial(oracle_sys* s, double lambda, double* trial_chi2, double* scale,
                      int32_t* pcg_iterations, int32_t* solver_ok);
/* Optimizer/Framework=1 systems: one DOGLEG step (radius, mu) from the current linearisation; step -> DX buffers, trial state -> TRIAL
 * buffers, out = { model cost change, scaled step norm, trial cost }.  Returns 0 when the factorisation fails. */
int oracle_sys_dogleg_trial(oracle_sys* s, double radius, double mu, double out[3]);
/* Same buffer ids and layouts as visfs_ba_stage_fetch. */
int oracle_sys_fetch(oracle_sys* s, int32_t which, double* dst, size_t n_doubles);
/* Both phases + outlier marking (Optimizer.cpp:261-318). Returns status; seconds = wall time of that region. */
int oracle_sys_optimize(oracle_sys* s, visfs_ba_stats* stats, double* seconds);
void oracle_sys_download(oracle_sys* s, double* pose_tq, double* point_xyz, uint8_t* obs_outlier, double* obs_chi2);
void oracle_sys_reset(oracle_sys* s);

/* Stepping a solve by hand: discardTop (the trial becomes the estimate), algorithm->init() (PCG residual forgotten), the outlier
 * pass of Optimizer.cpp:270-303 at the estimate (returns the number of edges moved to level 1). */
void oracle_sys_commit(oracle_sys* s);
void oracle_sys_begin_phase(oracle_sys* s);
int oracle_sys_mark_outliers(oracle_sys* s);

/* The LM / Gauss-Newton schedule ([g2o-upstream] OptimizationAlgorithmLevenberg::solve inside SparseOptimizer::optimize) run on
 * SCRIPTED trial outcomes: trial t of one optimize(n_iter) call returns (temp_chi[t], scale[t] = computeScale() without the
 * +1e-3, ok[t]); linearise reports the committed chi2 (chi0 at the start) and max_diag0.  Fills stats->trace_*, iterations_run[0],
 * trials_run[0], chi2_final (committed chi2); returns the number of trials consumed.  Same control-flow function as the solver. */
int oracle_lm_script(int gauss_newton, int n_iter, double chi0, double max_diag0, int n_trials, const double* temp_chi,
                     const double* scale, const int32_t* ok, visfs_ba_stats* stats);

/* Optimizer/Framework=1: [ceres-upstream] TrustRegionMinimizer + LevenbergMarquardtStrategy on scripted outcomes (see the .c file). */
int oracle_ceres_script(int max_iter, double cost0, double x_norm0, double grad_max0, int n, const int32_t* ok, const double* mcc,
                        const double* cand_cost, const double* step_norm, const double* grad_max, const double* x_norm, visfs_ba_stats* stats);
/* ... with [ceres-upstream] DoglegStrategy (Optimizer/TrustRegion=1): the scaled step lengths come with the script, mu_trace goes out. */
int oracle_dogleg_script(int max_iter, double cost0, double x_norm0, double grad_max0, int n, const int32_t* ok, const double* mcc,
                         const double* cand_cost, const double* step_norm, const double* dogleg_step_norm, const double* grad_max, const double* x_norm,
                         visfs_ba_stats* stats, double* mu_trace);
/* The point on the (traditional) dogleg path from the inner products of the scaled space: out = { A, B, step norm, model cost change }. */
void oracle_dogleg_combine(double S1, double S2, double S3, double JV2, double radius, double mu, double out[4]);

/* localOptimize-equivalent on host buffers (pack → optimise → write-back). */
int oracle_solve_window(const visfs_ba_params* params, const visfs_ba_window* w, visfs_ba_result* r, int num_threads);

/* bench.py's cpu_baseline leg only: bind thread t of the OpenMP team to cpus[t] (returns how many were bound; 0 in the serial build);
 * oracle_omp_unpin gives the calling thread its own affinity mask back. */
int oracle_omp_pin(const int32_t* cpus, int n);
void oracle_omp_unpin(void);

#ifdef __cplusplus
}
#endif
#endif
