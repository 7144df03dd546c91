// ba_kernels.hpp — launch wrappers of the gfx950 BA kernels (ba_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "ba_device.hpp"
#include "ba_math.hpp"

namespace visfs_ba {

// Launch geometry of one window, or the element-wise maximum over a batch of windows (same lanes-per-landmark group).
struct LaunchDims { int group, np, chunks, lin_blocks, backsub_blocks, run_wgs, run_lds, sch_wgs, sch_multi, fin_wgs, pcg_rows, pcg_lds, eval_blocks, reset_blocks, has_odo, pcg_one_wave, pcg_cu, band, band_lds, ceres, dogleg, fin_pcg, fin_pcg_wgs; };
LaunchDims dims_of(const DeviceGraph& g);
LaunchDims dims_max(const LaunchDims& a, const LaunchDims& b);

int configure_kernels(const DeviceGraph& g);
int pcg_resident_capacity(const LaunchDims& d, bool many, int device);   // workgroups of the persistent PCG that fit on the device at once (0: query failed)
void arm_launch_events(hipEvent_t start, hipEvent_t stop);             // measurement: attach an event pair to the next timed launch (this thread)
bool launch_events_pending();                                        // ... still armed: no timed launch has consumed it
void launch_build_index(const DeviceGraph& g, int32_t* hist, hipStream_t s);   // upload: lm_ptr, obs_ok, pose_obs / obs_ppos from the primary arrays (hist: [index_blocks(No)][Npf] scratch)
int index_blocks(int No);
void launch_build_pairs(const DeviceGraph& g, hipStream_t s);        // upload: co-observation pair lists of the S blocks
void launch_reset(const DeviceGraph& g, int max_iter, int gauss_newton, int restore, hipStream_t s);
void launch_linearize(const DeviceGraph& g, hipStream_t s);          // k_linearize (+ k_odo_linearize when the window has odometry edges)
void launch_linearize_decide(const DeviceGraph& g, hipStream_t s);   // last launch of a unit: linearise the trial state beside the LM decision (k_decide role)
void launch_lin_finalize(const DeviceGraph& g, int force, hipStream_t s);
void launch_schur_partial(const DeviceGraph& g, hipStream_t s);
void launch_schur_finalize(const DeviceGraph& g, hipStream_t s);
void launch_pcg(const DeviceGraph& g, hipStream_t s);                // persistent block-Jacobi PCG, one launch per damped solve
void launch_direct(const DeviceGraph& g, hipStream_t s);             // dense assemble + Cholesky
void launch_backsub(const DeviceGraph& g, hipStream_t s);
void launch_backsub_odospec(const DeviceGraph& g, hipStream_t s);   // speculative unit with odometry / laser edges: also linearises them at the trial poses
void launch_ceres_lin_finalize(const DeviceGraph& g, hipStream_t s);  // Optimizer/Framework=1: cost, ||g||_inf, ||x||, Jacobi scaling after every linearisation
void launch_backsub_lin_decide(const DeviceGraph& g, hipStream_t s); // fused speculative unit: back-substitution + LM decision + role A of the trial's linearisation
void launch_schur_partial_roleb(const DeviceGraph& g, hipStream_t s);  // ... its Schur gather, with the pending pose-major role of that linearisation behind it
void launch_backsub_decide(const DeviceGraph& g, hipStream_t s);    // gated unit: the launch also takes the LM decision (no k_decide)
void launch_decide(const DeviceGraph& g, hipStream_t s);
void launch_backsub_dogleg(const DeviceGraph& g, int pass, hipStream_t s);   // Optimizer/Framework=1 + TrustRegion=1: the two landmark passes of the dogleg step
void launch_dogleg_mid(const DeviceGraph& g, hipStream_t s);                  // ... the combination between them (coefficients, model cost change, trial poses)
void launch_phase_end(const DeviceGraph& g, int phase_just_done, int mark, int next_max_iter, hipStream_t s);
size_t band_lds_bytes(int npf, int B, int rows);                       // dynamic LDS of the banded direct solver (k_band_chol) with `rows` block rows resident
constexpr int BAND_LDS_BUDGET = 156 * 1024;
bool band_plan(int npf, int B, int* rows, int* lds_bytes);             // false: the band is too wide for k_band_chol (dense blocked Cholesky instead)
size_t pcg_cu_lds_bytes(int npf, int max_row);                       // dynamic LDS of k_pcg_cu: the slices of S that do not fit the registers
bool pcg_cu_fits(int npf, int max_row);                               // the reduced system fits the single-workgroup PCG (k_pcg_cu)
bool small_solve_fits(const DeviceGraph& g);                         // 6 Npf <= 64: S is finalised and solved by one workgroup
void launch_small_solve(const DeviceGraph& g, int solver, hipStream_t s);   // k_schur_finalize + solver + K8 in one launch
bool small_path_fits(const DeviceGraph& g);                          // the window qualifies for the fused single-workgroup path
void launch_small_optimize(const DeviceGraph& g, int solver, int half, hipStream_t s);   // both phases + outlier pass in one launch
// batches of independent windows: gs = B DeviceGraphs in HBM, blockIdx.y = window
void launch_reset_batch(const DeviceGraph* gs, int B, const LaunchDims& d, int max_iter, int gauss_newton, int restore, hipStream_t s);
// st1 (B == 1 only): the window's LM state at a fixed address, handed to the kernels beside the graph pointer (see Many::st1)
void launch_unit_batch(const DeviceGraph* gs, int B, const LaunchDims& d, bool first, bool small_solve, int solver, bool fused_decide, bool spec_fused, hipStream_t s, LmState* st1 = nullptr);
void launch_phase_end_batch(const DeviceGraph* gs, int B, const LaunchDims& d, int phase_just_done, int mark, int next_max_iter, hipStream_t s, LmState* st1 = nullptr);
LaunchDims dims_class(const LaunchDims& d);                           // launch geometry rounded up to size classes (per-frame launch sequences replayed across uploads)
bool dims_equal(const LaunchDims& a, const LaunchDims& b);
void launch_small_optimize_batch(const DeviceGraph* gs, int B, int solver, int half, hipStream_t s);
void launch_gather_lm(const DeviceGraph* gs, int B, LmState* out, hipStream_t s);
void launch_stage_arm(const DeviceGraph& g, double lambda, int mode, hipStream_t s);
void launch_eval_mark(const DeviceGraph& g, hipStream_t s);           // stage hook: outlier pass on the committed estimate, ungated
// test hook: one phase of the LM state machine on the host, through the functions the kernels run, on scripted trial outcomes
int ceres_script_host(int max_iter, double cost0, double x_norm0, double grad_max0, int n, const int32_t* ok, const double* mcc, const double* cand_cost,
                      const double* step_norm, const double* grad_max, const double* x_norm, LmState* st,
                      const double* dogleg_step_norm = nullptr, double* mu_trace = nullptr);
int lm_script_host(int gauss_newton, int n_iter, double chi0, double max_diag0, int n_trials, const double* temp_chi, const double* scale, const int32_t* ok, LmState* st);

}  // namespace visfs_ba
