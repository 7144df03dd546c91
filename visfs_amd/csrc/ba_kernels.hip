// ba_kernels.hip — hand-written gfx950 kernels of the sliding-window BA hot path.
//
// One "unit" of the device-side LM state machine is the launch sequence
//   k_linearize [k_odo_linearize] [k_lin_finalize]  →  k_schur_partial → k_schur_finalize
//   → k_pcg (persistent, one launch per damped solve) | k_dense_assemble + k_cholesky → k_backsub → k_decide
// Every kernel reads its gate (LmState::mode) from HBM and returns at once when it has nothing to do, so a
// whole optimise phase is enqueued without a host round trip.  `mode` is written only by single-workgroup
// kernels, so no multi-workgroup launch reads a gate that the same launch modifies.
//
// Mapping to the reference / g2o:
//   K1,K2  EdgeStereo::computeError / linearizeOplus          → k_linearize (role A landmark-major, role B pose-major), k_backsub, k_eval
//   K3     EdgePoseConstraint                                 → k_odo_linearize, k_backsub / k_eval (odometry role)
//   K4     constructQuadraticForm + RobustKernelHuber          → k_linearize + k_schur_finalize (Hpp/b_p sums)
//   K5     BlockSolver Schur complement                        → k_schur_partial (gather over co-observation pairs) + k_schur_finalize
//   K6     LinearSolverPCG / sparse Cholesky                   → k_pcg (persistent) / k_dense_assemble + k_cholesky
//   K7,K8  back-substitution, oplus                            → k_backsub (+ pose oplus in the solver epilogue)
//   K9     OptimizationAlgorithmLevenberg control              → k_lin_finalize (lambda init) + k_decide
//   K10    outlier marking (Optimizer.cpp:283-303)             → k_eval + k_phase_end
//   K11    EdgeOccupiedObservation (laser)                     → laser role of k_odo_linearize, k_backsub / k_eval
// Small windows: k_small_solve (finalise + solve reduced systems <= 64 x 64 in one workgroup); k_small_optimize (whole
// optimise in one workgroup, opt-in).  Every kernel is templated on the graph source: One (a window by value) or Many
// (blockIdx.y selects one of several independent windows sharing the launch).
//
// All reductions are fixed-order (halving butterflies + serial tails): no floating-point atomics, results are
// bitwise reproducible.  Wavefront = 64 everywhere.
#include "ba_kernels.hpp"

#include <float.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace visfs_ba {

// ---------------------------------------------------------------- wave helpers
// Wave-wide sum, every lane gets the result.  DPP lane moves (quad_perm, row_ror) build the four 16-lane row sums in
// the VALU — no LDS-crossbar ds_bpermute round trips — and four readlanes combine them in a fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);     // row_ror:4
    v += dpp_mov<0x128>(v);     // row_ror:8  → every lane holds the sum of its row of 16
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}
// Value of lane (lane ^ M) without the LDS crossbar: quad_perm / row_ror DPP moves inside a row of 16, v_permlane16_swap /
// v_permlane32_swap (gfx950) across rows — ds_bpermute costs an LDS round trip per 32 bits and the reductions below
// issue dozens per wave (tools/shfl_selftest.hip checks every variant against __shfl_xor on the device).
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <int M>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned v) {
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);            // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);       // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {                                                                     // row_ror:4 into banks 1,3; row_ror:12 into banks 0,2
        const unsigned d = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xA, false);
        return __builtin_amdgcn_update_dpp(d, v, 0x12C, 0xF, 0x5, false);
    }
    else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);      // row_ror:8
    else if constexpr (M == 16) { const v2u_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (threadIdx.x & 16) ? r.x : r.y; }
    else { static_assert(M == 32, "lane mask"); const v2u_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (threadIdx.x & 32) ? r.x : r.y; }
}
template <int M>
__device__ __forceinline__ double xor_lane(double v) {
    // across rows the permlane swaps measured slightly SLOWER than ds_bpermute in the Schur gather (two swaps + two selects per
    // double against two bpermutes whose latency other waves hide): DPP inside a row, the LDS crossbar across rows
    if constexpr (M >= 16) return __shfl_xor(v, M, 64);
    else {
        const unsigned lo = xor_lane_u32<M>((unsigned)__double2loint(v)), hi = xor_lane_u32<M>((unsigned)__double2hiint(v));
        return __hiloint2double((int)hi, (int)lo);
    }
}
template <int M> struct XorTree {          // butterflies over the masks M, M/2, ..., 1
    static __device__ __forceinline__ double sum(double v) { v += xor_lane<M>(v); return XorTree<M / 2>::sum(v); }
    static __device__ __forceinline__ double max(double v) { v = fmax(v, xor_lane<M>(v)); return XorTree<M / 2>::max(v); }
};
template <> struct XorTree<0> {
    static __device__ __forceinline__ double sum(double v) { return v; }
    static __device__ __forceinline__ double max(double v) { return v; }
};
__device__ __forceinline__ double wave_max(double v) { return XorTree<32>::max(v); }
template <int G>
__device__ __forceinline__ double group_sum(double v) { return XorTree<G / 2>::sum(v); }
template <int G>
__device__ __forceinline__ double group_max(double v) { return XorTree<G / 2>::max(v); }

// 1 / sqrt(x) for x > 0: v_rsq_f64 seed + two Newton steps (the IEEE sqrt + division pair costs ~500 dependent cycles, and
// this sits on the critical path of every eliminated column).
__device__ __forceinline__ double fast_rsqrt(const double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// Halving reduce-scatter: sums N per-lane values across an aligned group of W lanes with sum_k ceil(N/2^k)
// shuffles instead of N*log2(W).  At stage M a lane with bit M clear keeps the low half of its array and
// receives its partner's low half; a lane with the bit set does the same with the high halves.  On return
// a[j] holds the group sum of original element off + j for j < len (other slots are padding).
// One exchange of the halving reduce-scatter for the partner lane ^ M:  (up ? hi : lo) + the partner's (up ? hi : lo), where
// up = this lane has bit M set.  The textbook form selects send / keep (four v_cndmask per double) and moves `send` across;
// across rows the gfx950 swaps do the selecting themselves — v_permlane32_swap (v_permlane16_swap) exchanges the upper half
// (odd rows) of its first operand with the lower half (even rows) of its second, so after swap(lo, hi) a lower lane holds
// {own lo, partner's lo} and an upper lane {partner's hi, own hi}: the sum of the two registers is the result in every lane, with
// no select and no LDS crossbar trip.  Inside a row (M = 8, 4) two bank-masked DPP moves per word build the same two registers.
// The two addends are the same as in the select form (IEEE addition commutes): results are bit-identical to it.
template <int M>
__device__ __forceinline__ double halves_exchange_sum(const double lo, const double hi, const bool up) {
    if constexpr (VISFS_BA_RS_SWAP && (M == 32 || M == 16)) {
        v2u_t w0, w1;
        if constexpr (M == 32) {
            w0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
            w1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
        } else {
            w0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(lo), (unsigned)__double2loint(hi), false, false);
            w1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(lo), (unsigned)__double2hiint(hi), false, false);
        }
        return __hiloint2double((int)w1.x, (int)w0.x) + __hiloint2double((int)w1.y, (int)w0.y);
    } else if constexpr (VISFS_BA_RS_SWAP && (M == 8 || M == 4)) {
        // X: upper lanes take the partner's hi, lower lanes keep their lo;  Y: lower lanes take the partner's lo, upper keep their hi
        constexpr int UP_CTRL = (M == 8) ? 0x128 : 0x124, LO_CTRL = (M == 8) ? 0x128 : 0x12C;     // row_ror:8 | row_ror:4 / row_ror:12
        constexpr int UP_BANKS = (M == 8) ? 0xC : 0xA, LO_BANKS = (M == 8) ? 0x3 : 0x5;
        const unsigned l0 = (unsigned)__double2loint(lo), l1 = (unsigned)__double2hiint(lo), h0 = (unsigned)__double2loint(hi), h1 = (unsigned)__double2hiint(hi);
        const unsigned x0 = __builtin_amdgcn_update_dpp(l0, h0, UP_CTRL, 0xF, UP_BANKS, false), x1 = __builtin_amdgcn_update_dpp(l1, h1, UP_CTRL, 0xF, UP_BANKS, false);
        const unsigned y0 = __builtin_amdgcn_update_dpp(h0, l0, LO_CTRL, 0xF, LO_BANKS, false), y1 = __builtin_amdgcn_update_dpp(h1, l1, LO_CTRL, 0xF, LO_BANKS, false);
        return __hiloint2double((int)x1, (int)x0) + __hiloint2double((int)y1, (int)y0);
    } else {
        const double send = up ? lo : hi;
        const double keep = up ? hi : lo;
        return keep + xor_lane<M>(send);
    }
}

template <int N, int M>
struct ReduceScatter {
    static __device__ __forceinline__ void run(double* a, int lane, int& off, int& len) {
        constexpr int H = (N + 1) / 2;
        const bool up = (lane & M) != 0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const double lo = a[j];
            const double hi = (j + H < N) ? a[j + H] : 0.0;
            a[j] = halves_exchange_sum<M>(lo, hi, up);
        }
        if (up) { off += H; len = len > H ? len - H : 0; } else { len = len < H ? len : H; }
        ReduceScatter<H, M / 2>::run(a, lane, off, len);
    }
};
template <int N>
struct ReduceScatter<N, 0> {
    static __device__ __forceinline__ void run(double*, int, int&, int&) {}
};
// number of array slots a lane still holds after reducing N values over W lanes
constexpr int rs_slots(int N, int W) { return W <= 1 ? N : rs_slots((N + 1) / 2, W / 2); }

// Sum (or max) one value per thread over a 256-thread workgroup; every thread gets the result.
// red: LDS scratch of >= 4 doubles.  Fixed order: wave butterfly, then waves 0..3 serially.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double block_max_256(double v, double* red) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// Stage every pose of the window as R|t (12 doubles) in LDS: the observation tiles gather from here.
__device__ __forceinline__ void stage_poses(const double* __restrict__ pose, int Np, double* sRt) {
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const Rt T = pose_to_Rt(pose + POSE_STRIDE * i);
        double* o = sRt + 12 * i;
        o[0] = T.R.m00; o[1] = T.R.m01; o[2] = T.R.m02; o[3] = T.R.m10; o[4] = T.R.m11; o[5] = T.R.m12;
        o[6] = T.R.m20; o[7] = T.R.m21; o[8] = T.R.m22; o[9] = T.t.x; o[10] = T.t.y; o[11] = T.t.z;
    }
}
__device__ __forceinline__ Rt load_Rt(const double* sRt, int i) {
    const double* o = sRt + 12 * i;
    Rt T;
    T.R = Mat3{ o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8] };
    T.t = Vec3{ o[9], o[10], o[11] };
    return T;
}

// Where a kernel finds pose i as R|t: staged in LDS by stage_poses (12 doubles per pose: the observation tiles gather from there), or —
// windows with more poses than the LDS holds (MAX_STAGED_POSES) — converted on the fly from the estimate [tx ty tz qx qy qz qw] in HBM.
template <bool STG> struct PoseSrc {
    const double* p;
    __device__ __forceinline__ Rt get(const int i) const {
        if constexpr (STG) return load_Rt(p, i);
        else return pose_to_Rt(p + POSE_STRIDE * i);
    }
};

__device__ __forceinline__ Intrinsics intr_of(const DeviceGraph& g) { return Intrinsics{ g.fx, g.fy, g.cx, g.cy, g.bf }; }

// Where a kernel finds its window.  One: the DeviceGraph travels by value in the kernel arguments (single window).
// Many: independent windows solved side by side — blockIdx.y selects the window from an array in HBM,
// every window is gated by its own LmState, so the same launch serves windows at different points of their LM loops.
struct One { DeviceGraph g; static constexpr bool batched = false; };
// st1 (a launch that serves ONE window whose LM state sits at a fixed address — the per-frame path, whose captured launch sequence is
// replayed across uploads): the gate of every kernel is then read through a pointer that travels in the kernel arguments, in parallel
// with the first fields of the graph, instead of behind them; nullptr for batches.
struct Many { const DeviceGraph* gs; LmState* st1; static constexpr bool batched = true; };
// host side of the launchers: the by-value graph of a single window (a batch never reaches the callers of this: see staged())
inline const DeviceGraph& graph_of_host(const One& s) { return s.g; }
inline const DeviceGraph& graph_of_host(const Many&) { static const DeviceGraph none{}; return none; }
__device__ __forceinline__ const DeviceGraph& graph_of(const One& s) { return s.g; }
__device__ __forceinline__ LmState* state_of(const One&, const DeviceGraph& g) { return g.st; }
__device__ __forceinline__ LmState* state_of(const Many& s, const DeviceGraph& g) { return s.st1 ? s.st1 : g.st; }
// The graph array of a batch is written by the host before the launches and never by a kernel: reading it through the constant
// address space lets the compiler fetch the members with scalar loads into SGPRs (as it does for the by-value graph of One)
// instead of keeping vector-loaded copies live in VGPRs.
__device__ __forceinline__ const DeviceGraph& graph_of(const Many& s) {
    typedef const __attribute__((address_space(4))) DeviceGraph* ConstGraphPtr;
    ConstGraphPtr p = (ConstGraphPtr)(s.gs + blockIdx.y);
    return *(const DeviceGraph*)p;
}

// chi2() = e . (Omega e), Omega = I3 / pixelVariance (Optimizer.cpp:153)
// How a kernel gets its linearisation set: chosen at run time (LmState::lin_sel; lin_of: seven pointers in SGPRs).  Round 4: the
// windows of a batched launch (Many) run the fused speculative unit too, so they select between their two sets like a window on its
// own (round 1-3: always lin[0], read in place — computing the pointers up front had cost the batched k_linearize / k_backsub 9-13 %
// while the graph of a batch was still read with vector loads; it now arrives through the constant address space, in SGPRs).
template <class Src> struct LinSel {
    static constexpr bool two_sets = true;
    LinBuf L;
    __device__ __forceinline__ LinSel(const DeviceGraph& g, const int k) : L(lin_of(g, k)) {}
    __device__ __forceinline__ const LinBuf& get() const { return L; }
};

__device__ __forceinline__ double chi2_of(const Vec3& e, double iv) { return e.x * (iv * e.x) + e.y * (iv * e.y) + e.z * (iv * e.z); }
// The robust kernel of a stereo term: g2o's Huber when delta > 0 (Optimizer.cpp:212-216), Ceres' HuberLoss always (:370,469).
__device__ __forceinline__ void robustify(const DeviceGraph& g, const double c2, const double delta, double& rho0, double& rho1) {
    if (g.ceres) huber_ceres(c2, delta, rho0, rho1);
    else if (delta > 0.0) huber(c2, delta, rho0, rho1);
}
// Damping added to diagonal entry H_ii of variable idx (s2: the Jacobi scaling squared of its block): lambda for g2o's
// (H + lambda I); for the Ceres flavour lambda * clamp(H_ii s2, 1e-6, 1e32) / s2 — LevenbergMarquardtStrategy's diagonal
// sqrt(clamp(diag(J'^T J')) / radius) of the column-scaled Jacobian J' = J S, written in the unscaled variables.
__device__ __forceinline__ double damp_of(const DeviceGraph& g, const double lambda, const double hii, const double* __restrict__ s2, const size_t idx) {
    if (!g.ceres) return lambda;
    const double q = s2[idx];
    return lambda * (fmin(fmax(hii * q, 1e-6), 1e32) / q);
}

// Upper-triangle index of (r,c), r <= c, in the 21-entry packing used by role B.
__device__ __forceinline__ int upper_idx(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

// Entry q (< 36: Hpp(r,c); 36..41: b_p) of a free pose: fixed-order sum of its pose-major chunk partials [c0, c1)
// and of the odometry edges incident to it (entries [o0, o1) of pose_odo).
// (COH: the partials were written by other wavefronts of THIS launch — write-through stores — and are read past this CU's caches)
__device__ __forceinline__ double ld_coherent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <bool COH = false>
__device__ __forceinline__ double hpp_entry_r(const DeviceGraph& g, const LinBuf& L, int q, int c0, int c1, int o0, int o1) {
    double v = 0.0;
    if (q < 36) {
        const int r = q / 6, c = q % 6;
        const int u = r <= c ? upper_idx(r, c) : upper_idx(c, r);
#pragma unroll 4
        for (int ch = c0; ch < c1; ++ch) v += COH ? ld_coherent(L.hpp_part + 27 * (size_t)ch + u) : L.hpp_part[27 * (size_t)ch + u];
        for (int n = o0; n < o1; ++n) {
            const int code = g.pose_odo[n];
            v += L.odo_blk[120 * (size_t)(code >> 1) + ((code & 1) ? 36 : 0) + q];
        }
    } else {
        const int r = q - 36;
#pragma unroll 4
        for (int ch = c0; ch < c1; ++ch) v += COH ? ld_coherent(L.hpp_part + 27 * (size_t)ch + 21 + r) : L.hpp_part[27 * (size_t)ch + 21 + r];
        for (int n = o0; n < o1; ++n) {
            const int code = g.pose_odo[n];
            v += L.odo_blk[120 * (size_t)(code >> 1) + ((code & 1) ? 114 : 108) + r];
        }
    }
    return v;
}
__device__ __forceinline__ double hpp_entry(const DeviceGraph& g, const LinBuf& L, int a, int q) {
    return hpp_entry_r(g, L, q, g.pose_chunk_ptr[a], g.pose_chunk_ptr[a + 1], g.pose_odo_ptr[a], g.pose_odo_ptr[a + 1]);
}

// Role B, one observation of a free pose: upper triangle of Jx^T (rho' Omega) Jx and -Jx^T (rho' Omega) e into acc[27]
// (left untouched for an inactive edge).  Shared by k_linearize and the fused small-window kernel.
__device__ __forceinline__ void pose_obs_terms(const DeviceGraph& g, const int k, const Rt& T, const double* __restrict__ pt,
                                               const Intrinsics& K, const double iv, const double delta, double acc[27]) {
    const bool active = (g.obs_level[k] == 0) && g.obs_ok[k];
    if (!active) return;
    const int l = g.obs_pt[k];
    const Vec3 pw{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] };
    Vec3 pc;
    const Vec3 e = stereo_error(T, pw, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
    const double c2 = chi2_of(e, iv);
    double rho0 = c2, rho1 = 1.0;
    robustify(g, c2, delta, rho0, rho1);
    const double wo = rho1 * iv;
    double Jx[18];
    stereo_jacobian_pose(pc, K, Jx);
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = r; cc < 6; ++cc, ++q)
            acc[q] = Jx[r] * wo * Jx[cc] + Jx[6 + r] * wo * Jx[6 + cc] + Jx[12 + r] * wo * Jx[12 + cc];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] = -(Jx[r] * wo * e.x + Jx[6 + r] * wo * e.y + Jx[12 + r] * wo * e.z);
}

// The same terms from the pose-major record of the observation (DeviceGraph::pose_rec): the record does not depend on the LM state, so
// a kernel can have it in flight before its gate; what is left behind the gate is obs_level (the outlier pass moves edges to level 1)
// and the landmark.  Same arithmetic as pose_obs_terms on the same values.
__device__ __forceinline__ void pose_obs_terms_rec(const DeviceGraph& g, const DeviceGraph::PoseRec& rec, const Rt& T, const double* __restrict__ pt,
                                                   const Intrinsics& K, const double iv, const double delta, double acc[27]) {
    const bool active = (g.obs_level[rec.k] == 0) && rec.l_ok >= 0;
    if (!active) return;
    const int l = rec.l_ok;
    const Vec3 pw{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] };
    Vec3 pc;
    const Vec3 e = stereo_error(T, pw, rec.u, rec.v, rec.ur, K, pc);
    const double c2 = chi2_of(e, iv);
    double rho0 = c2, rho1 = 1.0;
    robustify(g, c2, delta, rho0, rho1);
    const double wo = rho1 * iv;
    double Jx[18];
    stereo_jacobian_pose(pc, K, Jx);
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = r; cc < 6; ++cc, ++q)
            acc[q] = Jx[r] * wo * Jx[cc] + Jx[6 + r] * wo * Jx[6 + cc] + Jx[12 + r] * wo * Jx[12 + cc];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] = -(Jx[r] * wo * e.x + Jx[6 + r] * wo * e.y + Jx[12 + r] * wo * e.z);
}

// Role A for one landmark handled by G lanes (sub = lane within the group): weights, chi2, tile seeds, Hll, b_l.
// Shared by k_linearize<G> and the fused small-window kernel (G = 1: one thread per landmark).
template <int G, bool STG = true>
__device__ __forceinline__ void lin_landmark(const DeviceGraph& g, const LinBuf& L, const int l, const bool lvalid, const int sub, const PoseSrc<STG> P,
                                             const double* __restrict__ pt, const Intrinsics& K, const double iv, const double delta,
                                             double& chi_acc, double& md, double* xn_acc = nullptr, const Vec3* pw_reg = nullptr) {
    int k0 = 0, k1 = 0;
    Vec3 pw{ 0, 0, 0 };
    bool lfree = false;
    if (lvalid) {
        k0 = g.lm_ptr[l]; k1 = g.lm_ptr[l + 1];
        // (pw_reg: the caller has the landmark in registers — k_backsub<LINA> linearises the trial landmark its own lanes have just formed;
        // reading it back from pt would race with the one lane of the group that stores it)
        pw = pw_reg ? *pw_reg : Vec3{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] };
        lfree = !g.pt_fixed[l];
    }
    // Optimizer/Framework=1: this pass also feeds the minimizer's bookkeeping (k_ceres_lin_finalize only adds the partials up): md
    // collects ||b_l||_inf instead of max |diag H_ll| (there is no lambda to initialise), xn_acc the landmark's share of ||x||^2, and at
    // iteration zero the Jacobi scaling of its three columns is written
    const bool ceres = g.ceres != 0;
    const bool ceres_first = ceres && g.st->phase_iter == 0;
    if (ceres && xn_acc && sub == 0 && lfree && k1 > k0) *xn_acc += pw.x * pw.x + pw.y * pw.y + pw.z * pw.z;
    double hb[9];                          // Hll (xx xy xz yy yz zz) then b_l
#pragma unroll
    for (int q = 0; q < 9; ++q) hb[q] = 0.0;
    for (int k = k0 + sub; k < k1; k += G) {
        const int ip = g.obs_pose[k];
        const Rt T = P.get(ip);
        const double u = g.obs_uvr[3 * k], v = g.obs_uvr[3 * k + 1], ur = g.obs_uvr[3 * k + 2];
        Vec3 pc;
        const Vec3 e = stereo_error(T, pw, u, v, ur, K, pc);
        const double c2 = chi2_of(e, iv);
        const bool active = (g.obs_level[k] == 0) && g.obs_ok[k];
        double rho0 = c2, rho1 = 1.0;
        robustify(g, c2, delta, rho0, rho1);
        L.obs_w[k] = active ? rho1 : 0.0;
        if (g.debug) { g.obs_chi2[k] = active ? c2 : 0.0; g.obs_err[3 * k] = active ? e.x : 0.0; g.obs_err[3 * k + 1] = active ? e.y : 0.0; g.obs_err[3 * k + 2] = active ? e.z : 0.0; }
        const bool pfree = g.pose_free[ip] >= 0;
        double wo_tile = 0.0;
        if (active) {
            chi_acc += rho0;
            const double wo = rho1 * iv;        // weightedOmega = rho' * Omega
            double Jp[9], Jx[18];
            stereo_jacobians(T, pc, K, Jp, Jx);
            if (lfree) {
                hb[0] += Jp[0] * wo * Jp[0] + Jp[3] * wo * Jp[3] + Jp[6] * wo * Jp[6];
                hb[1] += Jp[0] * wo * Jp[1] + Jp[3] * wo * Jp[4] + Jp[6] * wo * Jp[7];
                hb[2] += Jp[0] * wo * Jp[2] + Jp[3] * wo * Jp[5] + Jp[6] * wo * Jp[8];
                hb[3] += Jp[1] * wo * Jp[1] + Jp[4] * wo * Jp[4] + Jp[7] * wo * Jp[7];
                hb[4] += Jp[1] * wo * Jp[2] + Jp[4] * wo * Jp[5] + Jp[7] * wo * Jp[8];
                hb[5] += Jp[2] * wo * Jp[2] + Jp[5] * wo * Jp[5] + Jp[8] * wo * Jp[8];
                hb[6] -= Jp[0] * wo * e.x + Jp[3] * wo * e.y + Jp[6] * wo * e.z;
                hb[7] -= Jp[1] * wo * e.x + Jp[4] * wo * e.y + Jp[7] * wo * e.z;
                hb[8] -= Jp[2] * wo * e.x + Jp[5] * wo * e.y + Jp[8] * wo * e.z;
            }
            if (pfree && lfree) wo_tile = wo;
        }
        // the 32-byte tile seed (Hpl is rebuilt from it where it is consumed)
        double2* seed = reinterpret_cast<double2*>(L.obs_pcw + 4 * (size_t)k);
        seed[0] = make_double2(pc.x, pc.y);
        seed[1] = make_double2(pc.z, wo_tile);
        if (VISFS_BA_POSE_SEEDS && g.n_runs == 0) {   // (k_schur_runs reads the landmark-major seeds)
            const int pp = g.obs_ppos[k];             // the pose-major copy the Schur gather reads (coalesced there)
            if (pp >= 0) {
                double2* ps = reinterpret_cast<double2*>(L.pose_pcw + 4 * (size_t)pp);
                ps[0] = make_double2(pc.x, pc.y);
                ps[1] = make_double2(pc.z, wo_tile);
            }
        }
        if (g.debug) {
            double Wv[18];
            hpl_tile(T, pc, wo_tile, K, Wv);
            double2* Wk = reinterpret_cast<double2*>(g.W + 18 * (size_t)k);
#pragma unroll
            for (int q = 0; q < 9; ++q) Wk[q] = make_double2(Wv[2 * q], Wv[2 * q + 1]);
        }
    }
    // 9 sums over the G lanes of the landmark; each lane ends up owning rs_slots(9,G) of them
    int off = 0, len = 9;
    ReduceScatter<9, G / 2>::run(hb, sub, off, len);
    constexpr int SL = rs_slots(9, G);
    if (lvalid) {
#pragma unroll
        for (int j = 0; j < SL; ++j) {
            const int idx = off + j;
            if (j < len) {
                if (idx < 6) {
                    L.Hll[6 * (size_t)l + idx] = hb[j];
                    if (lfree && (idx == 0 || idx == 3 || idx == 5)) {
                        if (!ceres) md = fmax(md, fabs(hb[j]));
                        else if (ceres_first) { const double q = 1.0 / (1.0 + sqrt(hb[j])); g.s2l[3 * (size_t)l + (idx == 0 ? 0 : idx == 3 ? 1 : 2)] = q * q; }
                    }
                } else {
                    L.bl[3 * (size_t)l + (idx - 6)] = hb[j];
                    if (ceres && lfree) md = fmax(md, fabs(hb[j]));
                }
            }
        }
    }
}

// ================================================================= K9: Levenberg-Marquardt control
// [g2o-upstream] OptimizationAlgorithmLevenberg::solve, second half; one workgroup.  Sets the gate of the next unit.
// The scalar half of [g2o-upstream] OptimizationAlgorithmLevenberg::solve (and the Gauss-Newton variant): one thread.
// ok = the linear solve succeeded; chi / sc = robust chi2 at the trial state and computeScale's sum.
// spec: the accepted trial's linearisation is already in the other buffer set (speculative linearise): flip lin_sel with sel.
__host__ __device__ __forceinline__ void lm_decide(LmState* st, const bool ok, const double lambda, const double chi, const double sc, const int spec) {
    st->lin_b_pending = 0;
    const int ph = st->phase;
    st->trials_run[ph] += 1;
    st->solver_failed = 0;
    if (ok) st->n_active[3] += 1;
    if (st->pcg_timeout) { st->status = 8; st->done = 1; st->mode = 0; return; }     // VISFS_BA_ERR_DEVICE
    if (st->gauss_newton) {
        // OptimizationAlgorithmGaussNewton: always take the step; Fail ends the phase
        if (ok) { st->sel ^= 1; if (spec) st->lin_sel ^= 1; if (spec == 2) st->lin_b_pending = 1; }
        if (st->n_trace < MAX_TRACE) { st->trace_lambda[st->n_trace] = 0.0; st->trace_chi2[st->n_trace] = st->current_chi; st->n_trace++; }
        if (ok) st->current_chi = chi;
        st->phase_iter += 1; st->iterations_run[ph] = st->phase_iter;
        if (!ok || st->phase_iter >= st->max_iter) { st->done = 1; st->mode = 0; } else st->mode = MODE_LIN | MODE_TRIAL;
        return;
    }
    const double tempChi = ok ? chi : DBL_MAX;
    const double scale = (ok ? sc : 0.0) + 1e-3;
    const double rho = (st->current_chi - tempChi) / scale;
    st->temp_chi = tempChi; st->scale = scale; st->rho = rho;
    bool iteration_over = false, terminate = false;
    if (rho > 0.0 && tempChi <= DBL_MAX && tempChi == tempChi) {
        double alpha = 1.0 - pow(2.0 * rho - 1.0, 3.0);
        alpha = fmin(alpha, 2.0 / 3.0);
        const double scaleFactor = fmax(1.0 / 3.0, alpha);
        st->lambda = lambda * scaleFactor;
        st->ni = 2.0;
        st->current_chi = tempChi;
        st->sel ^= 1;                               // discardTop: the trial becomes the estimate
        if (spec) st->lin_sel ^= 1;
        if (spec == 2) st->lin_b_pending = 1;
        st->trial_q += 1;
        iteration_over = true;
    } else {
        const double nl = lambda * st->ni;
        st->lambda = nl;
        st->ni *= 2.0;                              // pop: estimate unchanged
        if (!(fabs(nl) <= DBL_MAX)) { iteration_over = true; terminate = true; }   // !isfinite(lambda): break before qmax++, and solve() returns Terminate
        else {
            st->trial_q += 1;
            if (!(rho < 0.0) || st->trial_q >= 10) iteration_over = true;   // loop runs while rho < 0 && qmax < 10
        }
    }
    if (iteration_over) {
        if (st->trial_q == 10 || rho == 0.0) terminate = true;
        if (st->n_trace < MAX_TRACE) { st->trace_lambda[st->n_trace] = st->lambda; st->trace_chi2[st->n_trace] = st->current_chi; st->n_trace++; }
        st->phase_iter += 1; st->iterations_run[ph] = st->phase_iter;
        st->trial_q = 0;
        if (terminate || st->phase_iter >= st->max_iter) { st->done = 1; st->mode = 0; }
        else st->mode = MODE_LIN | MODE_TRIAL;      // next unit linearises at the (possibly unchanged) estimate
    } else {
        st->mode = MODE_TRIAL;                      // same linearisation, larger lambda
    }
}

// ---- Optimizer/Framework=1: [ceres-upstream] TrustRegionMinimizer (Ceres 2.0 / 2.1 control flow) with LevenbergMarquardtStrategy, default
// Solver::Options except max_num_iterations (Optimizer.cpp:504-527).  One unit of the device state machine = one iteration of the
// minimizer loop; current_chi = 2 x cost.  solve_ok: the strategy produced a finite step; sc = step^T (D step + b) = 2 x
// model_cost_change; chi = sum of rho at the candidate; step_norm = ||x - candidate||.  The same function runs on the host under the
// scripted known-answer tests (visfs_ba_hook_ceres_script).
__host__ __device__ __forceinline__ void ceres_decide(LmState* st, const bool solve_ok, const double chi, const double sc, const double step_norm) {
    const int ph = st->phase;
    st->trials_run[ph] += 1;
    st->solver_failed = 0;
    if (solve_ok) st->n_active[3] += 1;
    st->phase_iter += 1; st->iterations_run[ph] = st->phase_iter;
    const double cost = 0.5 * st->current_chi, mcc = 0.5 * sc;
    bool stop = false, accepted = false;
    if (!solve_ok || !(mcc > 0.0)) {                                       // step_is_valid = model_cost_change > 0
        st->tr_invalid += 1;
        if (st->tr_invalid >= 5) { stop = true; st->tr_reason = 6; }         // max_num_consecutive_invalid_steps
        else if (st->dogleg) st->dl_mu *= 10.0;                             // DoglegStrategy::StepIsInvalid: mu *= mu_increase_factor
        else st->tr_radius *= 0.5;                                          // LevenbergMarquardtStrategy::StepIsInvalid
    } else {
        st->tr_invalid = 0;
        double cand = 0.5 * chi;
        if (!(fabs(cand) <= DBL_MAX)) cand = DBL_MAX;                      // the candidate could not be evaluated
        const double cost_change = cost - cand;
        if (step_norm <= 1e-8 * (st->tr_x_norm + 1e-8)) { stop = true; st->tr_reason = 3; }            // ParameterToleranceReached
        else if (fabs(cost_change) <= 1e-6 * cost) { stop = true; st->tr_reason = 4; }                 // FunctionToleranceReached: the step is not taken
        else {
            const double rho = cost_change / mcc;
            st->rho = rho; st->temp_chi = 2.0 * cand; st->scale = sc;
            if (rho > 1e-3) {                                                // min_relative_decrease
                accepted = true;
                st->current_chi = 2.0 * cand;
                st->sel ^= 1;                                               // the candidate becomes x
                if (st->dogleg) {
                    // [ceres-upstream] DoglegStrategy::StepAccepted: decrease_threshold 0.25, increase_threshold 0.75
                    if (rho < 0.25) st->tr_radius *= 0.5;
                    if (rho > 0.75) st->tr_radius = fmax(st->tr_radius, 3.0 * st->dl_step_norm);
                    st->dl_mu = fmax(1e-8, 2.0 * st->dl_mu / 10.0);        // back towards a pure Gauss-Newton solve
                } else {
                    st->tr_radius = st->tr_radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3.0));
                    st->tr_radius = fmin(1e16, st->tr_radius);
                    st->ni = 2.0;
                }
            } else if (st->dogleg) {
                st->tr_radius *= 0.5;                                       // DoglegStrategy::StepRejected (Ceres then reuses the Gauss-Newton step: same vectors)
            } else {
                st->tr_radius = st->tr_radius / st->ni;
                st->ni *= 2.0;
            }
        }
    }
    st->lambda = st->dogleg ? st->dl_mu : 1.0 / st->tr_radius;         // what the next unit's solve is damped with
    if (st->n_trace < MAX_TRACE) { st->trace_lambda[st->n_trace] = st->tr_radius; st->trace_chi2[st->n_trace] = st->current_chi; st->n_trace++; }
    // FinalizeIterationAndCheckIfMinimizerCanContinue (the gradient test of an accepted step follows its linearisation: k_ceres_lin_finalize)
    if (!stop && st->phase_iter >= st->max_iter) { stop = true; st->tr_reason = 1; }
    if (!stop && st->tr_radius < 1e-32) { stop = true; st->tr_reason = 5; }
    if (stop) { st->done = 1; st->mode = 0; }
    else st->mode = accepted ? (MODE_LIN | MODE_TRIAL) : MODE_TRIAL;
}
// A fresh linearisation (iteration zero, or the x an accepted step has produced): cost, ||x||, ||g||_inf; the trust region starts here.
__host__ __device__ __forceinline__ void ceres_lin_update(LmState* st, const double chi_total, const double grad_max, const double x_norm) {
    st->current_chi = chi_total;
    st->tr_x_norm = x_norm;
    st->max_diag = grad_max;
    if (st->phase_iter == 0) {
        st->chi2_initial = chi_total;
        st->tr_radius = 1e4; st->ni = 2.0; st->tr_invalid = 0; st->tr_reason = 0;     // initial_trust_region_radius
        st->dl_mu = 1e-8;                                                              // DoglegStrategy: kMinMu
        st->lambda = st->dogleg ? st->dl_mu : 1.0 / st->tr_radius;
    }
    if (grad_max <= 1e-10) { st->done = 1; st->mode = 0; st->tr_reason = 2; }          // GradientToleranceReached
}
// Does pose ip / landmark l belong to the reduced program (a non-constant parameter block that appears in a residual block)?
__device__ __forceinline__ bool ceres_pose_in_x(const DeviceGraph& g, const int ip) {
    const int a = g.pose_free[ip];
    if (a < 0) return false;
    return g.chunk_ptr[g.pose_chunk_ptr[a + 1]] > g.chunk_ptr[g.pose_chunk_ptr[a]] || (g.Nz > 0 && g.laser_pose == ip);
}
// The k_decide role of the Ceres flavour (256 threads): ||x - candidate|| over the reduced program, then the state machine.
__device__ __forceinline__ void ceres_decide_role(const DeviceGraph& g, LmState* st, const bool ok, const double chi, const double sc, double* red) {
    const int tid = threadIdx.x;
    double n2 = 0.0;
    if (ok) {
        for (int w = tid; w < g.n_lin_a; w += 256) n2 += g.aux_part[w];                  // the landmarks' shares, summed per workgroup by k_backsub
        const double* pa = g.pose[st->sel];
        const double* pb = g.pose[st->sel ^ 1];
        for (int t = tid; t < POSE_STRIDE * g.Np; t += 256) {
            const int ip = t / POSE_STRIDE, c = t % POSE_STRIDE;
            if (c < 7 && ceres_pose_in_x(g, ip)) { const double d = pb[t] - pa[t]; n2 += d * d; }
        }
    }
    n2 = block_sum_256(n2, red);
    if (tid != 0) return;
    const bool finite_step = fabs(n2) <= DBL_MAX;                           // IsArrayValid(step)
    ceres_decide(st, ok && finite_step, chi, sc, sqrt(n2));
}

__device__ __noinline__ void lm_decide_call(LmState* st, const bool ok, const double lambda, const double chi, const double sc) { lm_decide(st, ok, lambda, chi, sc, 0); }

// The k_decide role: 256 threads sum the trial's chi2 / computeScale partials, thread 0 steps the LM state machine.
__device__ __forceinline__ void decide_role(const DeviceGraph& g, LmState* st, double* red, const bool spec) {
    if (!(st->mode & MODE_TRIAL)) return;
    const int tid = threadIdx.x;
    const bool ok = !st->solver_failed && !st->pcg_timeout;
    const double lambda = st->lambda;
    double chi = 0.0, sc = 0.0;
    if (ok) {
        for (int w = tid; w < g.n_lin_a + 1; w += 256) { chi += g.trial_part[2 * w]; sc += g.trial_part[2 * w + 1]; }
        for (int t = tid; t < 6 * g.Npf; t += 256) { const double x = g.x[t]; sc += x * (damp_of(g, lambda, g.Hpp[36 * (size_t)(t / 6) + 7 * (t % 6)], g.s2p, t) * x + g.bp[t]); }
    }
    chi = block_sum_256(chi, red);
    sc = block_sum_256(sc, red);
    if (g.ceres) { ceres_decide_role(g, st, ok, chi, g.dogleg ? 2.0 * st->dl_mcc : sc, red); return; }   // (dogleg: the model cost change of k_dogleg_mid)
    if (tid != 0) return;
    lm_decide(st, ok, lambda, chi, sc, spec ? 1 : 0);
}

// The decision on board k_backsub (DEC): every workgroup publishes its two partial sums as hand-off words {tag:32 | half of a
// double:32} with write-through stores; one more workgroup per window — the LAST ones of the launch in dispatch order, so they
// start when the work is running out — reads the words with sc1 loads until every tag is there and takes the decision.  The data
// is the flag (cdna_hip_programming.md §6 Guideline 16, form R2, as in the persistent PCG): no fence, no counter (1875 relaxed
// atomic adds on one word cost the C4 launch 34 us: same-address atomics serialise at ~18 ns), no second launch.
// tag = the launch's number in LmState::decide_epoch, which only the decider advances, at the very end.
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p);
__device__ __forceinline__ void st_granule(unsigned long long* p, unsigned long long v);
__device__ __forceinline__ void publish_trial(const DeviceGraph& g, const int w, const unsigned ep, const double chi, const double sc) {
    unsigned long long* o = g.trial_gran + 4 * (size_t)w;
    const unsigned long long e = (unsigned long long)ep << 32;
    st_granule(o, e | (unsigned)__double2loint(chi)); st_granule(o + 1, e | (unsigned)__double2hiint(chi));
    st_granule(o + 2, e | (unsigned)__double2loint(sc)); st_granule(o + 3, e | (unsigned)__double2hiint(sc));
}
// The deciding workgroup (256 threads): fetch the partials of every workgroup of the window — four workgroups' words in flight per
// thread, fetched again after a pause while a tag is missing — add them in decide_role's order and step the LM state
// machine.  LmState is written only here, after every workgroup has published, i.e. after every workgroup has read its gate and
// its lambda / sel.  A wait that never ends (never expected) surfaces like a PCG hand-off time-out: VISFS_BA_ERR_DEVICE.
__device__ __forceinline__ void decide_gather_role(const DeviceGraph& g, LmState* st, const unsigned ep, const bool ok, double* red, const int spec = 0) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const double lambda = st->lambda;
    const int n = g.n_lin_a + 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.trial_gran, 0, (int)(n * 32), 0x00020000);
    double chi = 0.0, sc = 0.0;
    int bad = 0;
    // (kept small on purpose: the role shares the kernel's register budget with the landmark role — an eight-deep version with a
    // poll loop per word needed 256 VGPRs + 56 AGPRs and cost EVERY workgroup of k_backsub its occupancy: 16 -> 46 us at C4)
    constexpr int U = 4;
    for (int w0 = tid; w0 < n; w0 += 256 * U) {
        v4u_t a[U], b[U];
        for (int spin = 0;; ++spin) {
            bool ready = true;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int w = min(w0 + 256 * u, n - 1);             // (clamped: a thread's surplus slots re-read the last workgroup's words)
                a[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * w, 0, 16); b[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * w + 16, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) ready = ready && a[u].y == ep && a[u].w == ep && b[u].y == ep && b[u].w == ep;
            if (ready) break;
            if (spin > (1 << 20)) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(8);                 // ~0.2 us: the pollers must not crowd the memory system the workers are bound by
        }
        if (ok) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (w0 + 256 * u < n) { chi += __hiloint2double((int)a[u].z, (int)a[u].x); sc += __hiloint2double((int)b[u].z, (int)b[u].x); }
            }
        }
    }
    if (ok) for (int t = tid; t < 6 * g.Npf; t += 256) { const double x = g.x[t]; sc += x * (lambda * x + g.bp[t]); }
    chi = block_sum_256(chi, red);
    sc = block_sum_256(sc, red);
    bad = __syncthreads_or(bad);
    if (tid != 0) return;
    if (bad) st->pcg_timeout = 1;
    lm_decide(st, ok && !bad, lambda, chi, sc, spec);
    st->decide_epoch = ep;
}
// The deciders of a launch sit in its last grid row, behind the working workgroups: gridDim.x = working workgroups + gridDim.y,
// workgroup (working + j, gridDim.y - 1) decides for window j.  Returns the window, or -1 for a working (or idle) workgroup.
__device__ __forceinline__ int decider_window() {
    const int working = (int)gridDim.x - (int)gridDim.y;
    if ((int)blockIdx.x < working) return -1;
    return (blockIdx.y == gridDim.y - 1) ? (int)blockIdx.x - working : -2;          // -2: an idle filler of the rectangular grid
}
// The decider of window j.
__device__ __forceinline__ void decider_run(const DeviceGraph& gd, double* red, const int spec = 0) {
    LmState* sd = gd.st;
    if (sd->mode & MODE_TRIAL) decide_gather_role(gd, sd, sd->decide_epoch + 1u, !sd->solver_failed && !sd->pcg_timeout, red, spec);
}
__device__ __forceinline__ void decider_of(const One& s, int, double* red, const int spec = 0) { decider_run(s.g, red, spec); }
__device__ __forceinline__ void decider_of(const Many& s, const int j, double* red, const int spec = 0) {
    typedef const __attribute__((address_space(4))) DeviceGraph* ConstGraphPtr;
    decider_run(*(const DeviceGraph*)(ConstGraphPtr)(s.gs + j), red, spec);
}

// ================================================================= K1/K2/K4: linearise the stereo edges
// spec = 0: linearise the committed estimate (first unit of a phase, stage hooks, large windows) when the gate says so.
// spec = 1 ("speculative linearise", the last launch of a unit): the trial state that k_backsub has just completed is
// linearised into the OTHER buffer set while one extra workgroup takes the LM decision on that trial (the k_decide role): an
// accepted trial — the common case — finds its linearisation ready and only flips LmState::lin_sel; a rejected one leaves the
// current set untouched.  The linearising workgroups read the snapshot k_backsub left (spec_go / spec_src / spec_dst), never a
// field the decision writes, so the launch has no intra-kernel race.
template <int G, class Src, bool SPEC, bool STG = true>
__global__ __launch_bounds__(256) void k_linearize(const Src src) {
    constexpr bool spec = SPEC;
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sRt = smem;                       // [Np][12] (STG)
    double* red = smem + (STG ? 12 * g.Np : 0);   // [4 * 27]
    int sel, ls;
    if (spec) {
        if ((int)blockIdx.x == g.n_lin_a + g.n_chunks) { decide_role(g, st, red, true); return; }
        if (!st->spec_go) return;
        sel = st->spec_src; ls = st->spec_dst;
    } else {
        if (!(st->mode & MODE_LIN)) return;
        sel = st->sel; ls = st->lin_sel;
    }
    sel = __builtin_amdgcn_readfirstlane(sel);          // wave-uniform: keeps the selected estimate pointers in SGPRs
    const double* __restrict__ pose = g.pose[sel];
    const double* __restrict__ pt = g.pt[sel];
    if (STG) { stage_poses(pose, g.Np, sRt); __syncthreads(); }
    const PoseSrc<STG> P{ STG ? sRt : pose };
    const LinSel<Src> lsel(g, ls); const LinBuf& L = lsel.get();                     // after the staging: its pointer loads overlap the pose loads
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta;
    const int bid = blockIdx.x, tid = threadIdx.x;

    if (bid < g.n_lin_a) {
        // ---- role A: landmark-major, G lanes per landmark: Hpl tiles, Hll, b_l, weights, robust chi2
        constexpr int LPW = 256 / G;
        const int l = bid * LPW + tid / G, sub = tid % G;
        const bool lvalid = l < g.Nl;
        double chi_acc = 0.0, md = 0.0, xn = 0.0;
        lin_landmark<G, STG>(g, L, l, lvalid, sub, P, pt, K, iv, delta, chi_acc, md, &xn);
        const double chi_tot = block_sum_256(chi_acc, red);
        const double md_tot = block_max_256(md, red);
        if (tid == 0) { g.lin_part[2 * bid] = chi_tot; g.lin_part[2 * bid + 1] = md_tot; }
        if (g.ceres) { const double xn_tot = block_sum_256(xn, red); if (tid == 0) g.aux_part[bid] = xn_tot; }
    } else {
        // ---- role B: pose-major chunk: upper triangle of Jx^T (rho' Omega) Jx and -Jx^T (rho' Omega) e
        const int c = bid - g.n_lin_a;
        if (c >= g.n_chunks) return;                 // a batched launch is sized for the largest window
        const int a = g.chunk_pose[c];
        const int begin = g.chunk_ptr[c], end = g.chunk_ptr[c + 1];
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        if (begin + tid < end) pose_obs_terms_rec(g, g.pose_rec[begin + tid], P.get(g.free_pose[a]), pt, K, iv, delta, acc);   // (one coalesced 32-byte record per observation)
        const int wave = tid >> 6, lane = tid & 63;
        int off = 0, len = 27;
        ReduceScatter<27, 32>::run(acc, lane, off, len);
        if (len >= 1) red[wave * 27 + off] = acc[0];
        __syncthreads();
        if (tid < 27) L.hpp_part[27 * (size_t)c + tid] = red[tid] + red[27 + tid] + red[54 + tid] + red[81 + tid];
    }
}

// One wheel-odometry edge: its five 6x6 / 6x1 contributions into odo_blk[e_] (120 doubles); returns its chi2
// (0 and zero blocks when both poses are fixed: allVerticesFixed).  Shared with the fused small-window kernel.
__device__ __forceinline__ double odo_edge_blocks(const DeviceGraph& g, const LinBuf& L, const int e_, const double* __restrict__ pose, const double ic) {
    const int i = g.odo_i[e_], j = g.odo_j[e_];
    const bool fi = g.pose_free[i] >= 0, fj = g.pose_free[j] >= 0;
    double* o = L.odo_blk + 120 * (size_t)e_;
    if (!fi && !fj) { for (int q = 0; q < 120; ++q) o[q] = 0.0; return 0.0; }
    double e[6], Ji[36], Jj[36];
    odo_linearize(pose + POSE_STRIDE * i, pose + POSE_STRIDE * j, g.odo_tq + 7 * e_, e, Ji, Jj);
    double c2 = 0.0;
#pragma unroll
    for (int d = 0; d < 6; ++d) c2 += e[d] * (ic * e[d]);
    for (int r = 0; r < 6; ++r) {
        for (int cc = 0; cc < 6; ++cc) {
            double aii = 0, ajj = 0, aij = 0;
            for (int d = 0; d < 6; ++d) {
                aii += Ji[d * 6 + r] * ic * Ji[d * 6 + cc];
                ajj += Jj[d * 6 + r] * ic * Jj[d * 6 + cc];
                aij += Ji[d * 6 + r] * ic * Jj[d * 6 + cc];
            }
            o[r * 6 + cc] = fi ? aii : 0.0;
            o[36 + r * 6 + cc] = fj ? ajj : 0.0;
            o[72 + r * 6 + cc] = (fi && fj) ? aij : 0.0;
        }
        double bi = 0, bj = 0;
        for (int d = 0; d < 6; ++d) { bi += Ji[d * 6 + r] * ic * e[d]; bj += Jj[d * 6 + r] * ic * e[d]; }
        o[108 + r] = fi ? -bi : 0.0;
        o[114 + r] = fj ? -bj : 0.0;
    }
    return c2;
}

// One range point: J^T Omega J (upper triangle) and -J^T Omega e added to acc[27]; returns its chi2.
__device__ __forceinline__ double laser_point_terms(const DeviceGraph& g, const double* tq, const int z, const double il, double acc[27]) {
    const Vec3 P{ g.laser_xyz[3 * z], g.laser_xyz[3 * z + 1], g.laser_xyz[3 * z + 2] };
    const double e = laser_error(tq, g.Tcr, P, g.grid);
    double J[6];
    laser_jacobian(tq, g.Tcr, P, g.grid, J, g.ceres != 0);
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = r; cc < 6; ++cc, ++q) acc[q] += J[r] * il * J[cc];
#pragma unroll
    for (int r = 0; r < 6; ++r) acc[21 + r] -= J[r] * il * e;
    return e * (il * e);
}
// Entry t (< 27) of the reduced laser sums into the pseudo-edge slot Ne of odo_blk (full symmetric 6x6 + b).
__device__ __forceinline__ void laser_store_slot(const DeviceGraph& g, const LinBuf& L, const int t, const double v) {
    double* o = L.odo_blk + 120 * (size_t)g.Ne;
    if (t < 21) {
        int r = 0, base = 0;
        while (t >= base + (6 - r)) { base += 6 - r; ++r; }
        const int cc = r + (t - base);
        o[r * 6 + cc] = v; o[cc * 6 + r] = v;
    } else o[108 + (t - 21)] = v;
}

// ================================================================= K3: wheel-odometry edges
// EdgePoseConstraint, Omega = I6 / odometryCovariance (Optimizer.cpp:117-121), no robust kernel.  One workgroup.
// The role as a function (256 threads; red4: 4 doubles, redz: 4 * 27 doubles of LDS): linearises every odometry edge and the laser
// edges at `pose` into set L, returns the workgroup's chi2 sum.  Run by k_odo_linearize and — for the speculative unit — by the
// odometry workgroup of k_backsub, which has the trial poses in hand a whole launch earlier.
__device__ __forceinline__ double odo_role(const DeviceGraph& g, const LinBuf& L, const double* __restrict__ pose, double* red4, double* redz) {
    const double ic = g.inv_odo_cov;
    const int tid = threadIdx.x;
    double chi_acc = 0.0;
    for (int e_ = tid; e_ < g.Ne; e_ += 256) {
        chi_acc += odo_edge_blocks(g, L, e_, pose, ic);
    }
    // laser occupied-space edges (EdgeOccupiedObservation, Omega = 1 / laserCovariance, Optimizer.cpp:232-249, no kernel):
    // all on one pose, so the workgroup reduces J^T Omega J (upper triangle) and -J^T Omega e into slot Ne of odo_blk.
    if (g.Nz > 0) {
        const double il = g.inv_laser_cov;
        const double* tq = pose + POSE_STRIDE * g.laser_pose;
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        for (int z = tid; z < g.Nz; z += 256) {
            chi_acc += laser_point_terms(g, tq, z, il, acc);
        }
        const int wave = tid >> 6, lane = tid & 63;
        int off = 0, len = 27;
        ReduceScatter<27, 32>::run(acc, lane, off, len);
        if (len >= 1) redz[wave * 27 + off] = acc[0];
        __syncthreads();
        if (tid < 27) {
            laser_store_slot(g, L, tid, redz[tid] + redz[27 + tid] + redz[54 + tid] + redz[81 + tid]);
        }
    }
    return block_sum_256(chi_acc, red4);
}

template <class Src>
__global__ __launch_bounds__(256) void k_odo_linearize(const Src src, const int spec_arg) {
    const bool spec = !Src::batched && spec_arg;
    const DeviceGraph& g = graph_of(src);
    const LmState* st = state_of(src, g);
    int sel, ls;
    if (spec) { if (!st->spec_go) return; sel = st->spec_src; ls = st->spec_dst; }
    else { if (!(st->mode & MODE_LIN)) return; sel = st->sel; ls = st->lin_sel; }
    const LinSel<Src> lsel(g, ls); const LinBuf& L = lsel.get();
    __shared__ double red[4];
    __shared__ double redz[4 * 27];
    const double chi_tot = odo_role(g, L, g.pose[sel], red, redz);
    if (threadIdx.x == 0) { g.lin_part[2 * g.n_lin_a] = chi_tot; g.lin_part[2 * g.n_lin_a + 1] = 0.0; }
}

// One thread: chi2 / max|diag H| of a fresh linearisation; computeLambdaInit on the first iteration of a phase ([g2o-upstream] tau = 1e-5).
__host__ __device__ __noinline__ void lin_finalize_update(LmState* st, const double chi_total, const double md_total) {
    st->current_chi = chi_total;
    st->max_diag = md_total;
    if (st->phase_iter == 0) {
        if (st->phase == 0) st->chi2_initial = chi_total;
        st->lambda = st->gauss_newton ? 0.0 : 1e-5 * md_total;
        st->ni = 2.0;
    }
}

// Test hook (visfs_ba_hook_lm_script): the LM state machine of ONE phase stepped on the host by the very functions the kernels
// run (lin_finalize_update, lm_decide — compiled for both sides), on scripted trial outcomes.  No solve happens here: this checks
// the control flow (accept / reject, lambda schedule, Terminate rules, failure paths) against the checker's restatement of
// [g2o-upstream] OptimizationAlgorithmLevenberg::solve without a device.
int lm_script_host(const int gauss_newton, const int n_iter, const double chi0, const double max_diag0, const int n_trials,
                   const double* temp_chi, const double* scale, const int32_t* ok, LmState* st) {
    *st = LmState{};
    st->ni = 2.0; st->pcg_res_in = -1.0; st->pcg_residual = -1.0; st->spec_dst = 1;
    st->max_iter = n_iter; st->gauss_newton = gauss_newton;
    st->done = n_iter <= 0 ? 1 : 0; st->mode = st->done ? 0 : (MODE_LIN | MODE_TRIAL);
    double committed = chi0;
    int pos = 0;
    for (int guard = 0; st->mode != 0 && guard < 100000; ++guard) {
        // first unit of a phase: k_lin_finalize; later units take current_chi from the accepted trial (as the kernels do)
        if ((st->mode & MODE_LIN) && st->phase_iter == 0) lin_finalize_update(st, committed, max_diag0);
        const int t = pos < n_trials ? pos : n_trials - 1;
        ++pos;
        const int sel_before = st->sel;
        lm_decide(st, ok[t] != 0, st->lambda, temp_chi[t], scale[t], false);
        if (st->sel != sel_before) committed = temp_chi[t];
    }
    st->chi2_final = committed;
    return pos;
}

// Test hook (visfs_ba_hook_ceres_script): the Ceres-flavour state machine stepped on the host by the functions the kernels run
// (ceres_lin_update, ceres_decide), on scripted outcomes: iteration t's solve reports (ok, model_cost_change, candidate cost,
// ||step||); an accepted step then reports (||g||_inf, ||x||) of its linearisation.
int ceres_script_host(const int max_iter, const double cost0, const double x_norm0, const double grad_max0, const int n, const int32_t* ok,
                      const double* mcc, const double* cand_cost, const double* step_norm, const double* grad_max, const double* x_norm, LmState* st,
                      const double* dogleg_step_norm, double* mu_trace) {
    *st = LmState{};
    st->dogleg = dogleg_step_norm ? 1 : 0;                     // the DOGLEG strategy: iteration t's step had the scaled length dogleg_step_norm[t]
    st->ni = 2.0; st->max_iter = max_iter; st->current_chi = 2.0 * cost0;
    st->done = max_iter <= 0 ? 1 : 0; st->mode = st->done ? 0 : (MODE_LIN | MODE_TRIAL);
    if (st->done) st->tr_reason = 1;
    if (st->mode & MODE_LIN) ceres_lin_update(st, 2.0 * cost0, grad_max0, x_norm0);
    int pos = 0;
    for (int guard = 0; st->mode != 0 && guard < 100000; ++guard) {
        const int q = pos < n ? pos : n - 1;
        ++pos;
        if (dogleg_step_norm) st->dl_step_norm = dogleg_step_norm[q];
        ceres_decide(st, ok[q] != 0, 2.0 * cand_cost[q], 2.0 * mcc[q], step_norm[q]);
        if (mu_trace && st->n_trace >= 1 && st->n_trace <= MAX_TRACE) mu_trace[st->n_trace - 1] = st->dl_mu;
        if (st->mode & MODE_LIN) ceres_lin_update(st, st->current_chi, grad_max[q], x_norm[q]);
    }
    st->chi2_final = st->current_chi;
    return st->tr_reason;
}

// Single workgroup, launched in the FIRST unit of a phase (and by the stage hook): sums Hpp/b_p, reduces the
// robust chi2 and max|diag H| of the linearisation and does computeLambdaInit ([g2o-upstream] tau = 1e-5).
// Later units take current_chi from the accepted trial and lambda from k_decide.
template <class Src>
__global__ __launch_bounds__(1024) void k_lin_finalize(const Src src, const int force) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    if (!(st->mode & MODE_LIN)) return;
    if (!force && st->phase_iter != 0) return;
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    __shared__ double red[32];
    const int tid = threadIdx.x;
    double md = 0.0;
    for (int t = tid; t < g.Npf * 42; t += 1024) {
        const int a = t / 42, q = t % 42;
        const double v = hpp_entry(g, L, a, q);
        if (q < 36) { g.Hpp[36 * (size_t)a + q] = v; if (q % 7 == 0) md = fmax(md, fabs(v)); }
        else g.bp[6 * (size_t)a + (q - 36)] = v;
    }
    double chi = 0.0;
    const int nparts = g.n_lin_a + 1;
    for (int w = tid; w < nparts; w += 1024) { chi += g.lin_part[2 * w]; md = fmax(md, g.lin_part[2 * w + 1]); }
    // two workgroup reductions through the wave butterflies and 16 per-wave partials each (fixed order): ONE barrier instead of the
    // twenty of two log-step LDS trees
    const double wchi = wave_sum(chi), wmd = wave_max(md);
    if ((tid & 63) == 0) { red[tid >> 6] = wchi; red[16 + (tid >> 6)] = wmd; }
    __syncthreads();
    if (tid == 0) {
        double chi_total = red[0], md_total = red[16];
        for (int w = 1; w < 16; ++w) { chi_total += red[w]; md_total = fmax(md_total, red[16 + w]); }
        lin_finalize_update(st, chi_total, md_total);
    }
}

// Optimizer/Framework=1: the tail of EVERY linearisation (iteration zero and after each accepted step).  One workgroup: sums Hpp / b_p,
// the cost, ||g||_inf and ||x|| of the reduced program; at iteration zero also the Jacobi scaling 1 / (1 + sqrt(H_ii)) of every variable
// ([ceres-upstream] TrustRegionMinimizer::IterationZero / EstimateScale).
template <class Src>
__global__ __launch_bounds__(1024) void k_ceres_lin_finalize(const Src src) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    if (!(st->mode & MODE_LIN)) return;
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    __shared__ double red[48];
    const int tid = threadIdx.x;
    const bool first = st->phase_iter == 0;
    double gm = 0.0, chi = 0.0, xn = 0.0;
    for (int t = tid; t < g.Npf * 42; t += 1024) {
        const int a = t / 42, q = t % 42;
        const double v = hpp_entry(g, L, a, q);
        if (q < 36) { g.Hpp[36 * (size_t)a + q] = v; if (first && q % 7 == 0) { const double s = 1.0 / (1.0 + sqrt(v)); g.s2p[6 * (size_t)a + q / 7] = s * s; } }
        else { g.bp[6 * (size_t)a + (q - 36)] = v; gm = fmax(gm, fabs(v)); }
    }
    // (the landmark part — ||b_l||_inf, the landmarks' share of ||x||^2, their Jacobi scaling — was done by k_linearize's landmark pass)
    for (int w = tid; w < g.n_lin_a; w += 1024) { gm = fmax(gm, g.lin_part[2 * w + 1]); xn += g.aux_part[w]; }
    for (int t = tid; t < POSE_STRIDE * g.Np; t += 1024) {
        const int ip = t / POSE_STRIDE, c = t % POSE_STRIDE;
        if (c < 7 && ceres_pose_in_x(g, ip)) { const double v = g.pose[st->sel][t]; xn += v * v; }
    }
    const int nparts = g.n_lin_a + 1;
    for (int w = tid; w < nparts; w += 1024) chi += g.lin_part[2 * w];
    const int wave = tid >> 6, lane = tid & 63;
    chi = wave_sum(chi); xn = wave_sum(xn); gm = wave_max(gm);
    if (lane == 0) { red[wave] = chi; red[16 + wave] = gm; red[32 + wave] = xn; }
    __syncthreads();
    if (tid == 0) {
        double chi_total = red[0], gm_total = red[16], xn_total = red[32];
        for (int w = 1; w < 16; ++w) { chi_total += red[w]; gm_total = fmax(gm_total, red[16 + w]); xn_total += red[32 + w]; }
        ceres_lin_update(st, chi_total, gm_total, sqrt(xn_total));
    }
}

// ================================================================= upload: co-observation pair lists of the S blocks
// g2o's buildStructure analogue for the gather-form Schur complement: block (i <= j) of S lists, in landmark order, the pairs
// (observation of pose i, observation of pose j, landmark) of the free landmarks both poses observe.  One wavefront per block
// intersects the two poses' observation lists (pose-major, hence sorted by landmark; a pose sees a landmark at most once): a
// lane per observation of pose i, a binary search in pose j's list (pose_lm: the landmark of every pose-major entry), a ballot
// prefix for the slot — no atomics, so the order
// (and with it every later summation order) is fixed.  The host has sized blk_ptr from the same rule (counts only).
// ---- the O(N_obs) index arrays, built on the device from the primary arrays (obs_pt, obs_pose, pose_free, pt_fixed): the host only
// sends what localOptimize's inputs carry and the block-level structure of S; nothing of size N_obs is computed or copied twice.
//   lm_ptr    CSR over the landmark-major observations (observations arrive sorted by (point, pose): boundaries of obs_pt);
//   obs_ok    !(pose fixed && point fixed);
//   pose_obs  the pose-major permutation (free poses only, ascending observation id inside a pose), obs_ppos its inverse.
// The permutation is a stable multi-split over blocks of IDX_T observations: inside a wavefront one ballot per DISTINCT free pose gives
// every lane its rank among the equal-pose lanes below it and the wavefront's count of that pose (k_index_count sums the four
// wavefronts' counts per block, k_index_scan turns the block counts into offsets, k_index_scatter places) — no atomics, order fixed.
constexpr int IDX_T = 256;
__global__ __launch_bounds__(256) void k_index_count(const DeviceGraph g, int32_t* __restrict__ hist) {
    __shared__ int sfree[IDX_T];
    const int k0 = blockIdx.x * IDX_T, tid = threadIdx.x, k = k0 + tid;
    const int n = min(IDX_T, g.No - k0);
    int32_t* lm_ptr = const_cast<int32_t*>(g.lm_ptr);
    if (tid < n) {
        const int c = g.obs_pose[k], l = g.obs_pt[k];
        const int a = g.pose_free[c];
        sfree[tid] = a;
        if (g.obs_uvd) {
            // the stereo measurement exactly as Optimizer.cpp:187-188 forms it: disparity = float(baseLine * fx / depth) in double, then
            // float; u_r = u - disparity in FLOAT arithmetic, promoted (IEEE double division and round-to-nearest conversions on both sides)
            const float u = g.obs_uvd[3 * (size_t)k], v = g.obs_uvd[3 * (size_t)k + 1], dpt = g.obs_uvd[3 * (size_t)k + 2];
            const float disparity = static_cast<float>(g.stereo_baseline * g.fx / (double)dpt);
            double* uvr = const_cast<double*>(g.obs_uvr) + 3 * (size_t)k;
            uvr[0] = (double)u; uvr[1] = (double)v; uvr[2] = (double)(u - disparity);
        }
        const_cast<uint8_t*>(g.obs_ok)[k] = !(a < 0 && g.pt_fixed[l]);
        if (a < 0) const_cast<int32_t*>(g.obs_ppos)[k] = -1;
        const int lprev = k > 0 ? g.obs_pt[k - 1] : -1;
        for (int q = lprev + 1; q <= l; ++q) lm_ptr[q] = k;                   // landmarks without observations get empty ranges
        if (k == g.No - 1) for (int q = l + 1; q <= g.Nl; ++q) lm_ptr[q] = g.No;
    }
    // per-wavefront counts of every free pose among the wave's 64 observations (one ballot per DISTINCT pose present), summed over the
    // four wavefronts: thread a scanning all 256 entries was 256 dependent LDS reads per thread
    extern __shared__ int wcnt[];                                             // [4][Npf]
    for (int q = tid; q < 4 * g.Npf; q += 256) wcnt[q] = 0;
    __syncthreads();
    {
        const int a = tid < n ? sfree[tid] : -1;
        const int lane = tid & 63, wave = tid >> 6;
        unsigned long long todo = __ballot(a >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int key = __shfl(a, leader);
            const unsigned long long m = __ballot(a == key);
            if (lane == leader) wcnt[wave * g.Npf + key] = __popcll(m);
            todo &= ~m;
        }
    }
    __syncthreads();
    for (int a = tid; a < g.Npf; a += 256) hist[(size_t)blockIdx.x * g.Npf + a] = wcnt[a] + wcnt[g.Npf + a] + wcnt[2 * g.Npf + a] + wcnt[3 * g.Npf + a];
}
// exclusive scan of the per-block counts over the blocks, per pose, offset by where the pose's range starts in pose_obs.
// One workgroup per free pose: thread t sums a contiguous run of blocks, the 256 run sums are scanned in LDS, each thread writes its
// run's exclusive prefixes back (one thread per pose walking all blocks was 10.7 us at C2's 196 blocks, a latency chain).
__global__ __launch_bounds__(256) void k_index_scan(const DeviceGraph g, int32_t* __restrict__ hist, const int nblocks) {
    if (g.No == 0 && threadIdx.x == 0 && blockIdx.x == 0) { int32_t* lm_ptr = const_cast<int32_t*>(g.lm_ptr); for (int q = 0; q <= g.Nl; ++q) lm_ptr[q] = 0; }
    const int a = blockIdx.x, tid = threadIdx.x;
    if (a >= g.Npf) return;
    __shared__ int ssum[256];
    const int per = (nblocks + 255) / 256;
    const int b0 = min(tid * per, nblocks), b1 = min(b0 + per, nblocks);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += hist[(size_t)b * g.Npf + a];
    ssum[tid] = sum;
    __syncthreads();
    // inclusive scan over the 256 run sums (Hillis-Steele, 8 steps)
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? ssum[tid - off] : 0;
        __syncthreads();
        ssum[tid] += v;
        __syncthreads();
    }
    int run = g.chunk_ptr[g.pose_chunk_ptr[a]] + ssum[tid] - sum;
    for (int b = b0; b < b1; ++b) { const int c = hist[(size_t)b * g.Npf + a]; hist[(size_t)b * g.Npf + a] = run; run += c; }
}
// Stable multi-split, second half: observation k of free pose a goes to base[block][a] + (observations of a in the block's earlier
// wavefronts) + (its rank among the equal-pose lanes below it in its own wavefront) — ballots, one per distinct pose of the wavefront.
// Also writes the pose-major landmark list (pose_lm) the pair builder searches.
__global__ __launch_bounds__(256) void k_index_scatter(const DeviceGraph g, const int32_t* __restrict__ base) {
    extern __shared__ int wcnt[];                                             // [4][Npf]
    const int k0 = blockIdx.x * IDX_T, tid = threadIdx.x, k = k0 + tid;
    const int n = min(IDX_T, g.No - k0);
    for (int q = tid; q < 4 * g.Npf; q += 256) wcnt[q] = 0;
    __syncthreads();
    const int a = tid < n ? g.pose_free[g.obs_pose[k]] : -1;
    const int lane = tid & 63, wave = tid >> 6;
    int rank = 0;
    {
        unsigned long long todo = __ballot(a >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int key = __shfl(a, leader);
            const unsigned long long m = __ballot(a == key);
            if (a == key) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == leader) wcnt[wave * g.Npf + key] = __popcll(m);
            todo &= ~m;
        }
    }
    __syncthreads();
    if (a >= 0) {
        int pos = base[(size_t)blockIdx.x * g.Npf + a] + rank;
        for (int w = 0; w < wave; ++w) pos += wcnt[w * g.Npf + a];
        const_cast<int32_t*>(g.pose_obs)[pos] = k;
        const_cast<int32_t*>(g.obs_ppos)[k] = pos;
        const int l = g.obs_pt[k];
        g.pose_lm[pos] = l;
        // (obs_ok and obs_uvr of this observation were written by k_index_count, the launch before the scan)
        DeviceGraph::PoseRec rec;
        rec.k = k; rec.l_ok = g.obs_ok[k] ? l : (l | (int)0x80000000);
        rec.u = g.obs_uvr[3 * (size_t)k]; rec.v = g.obs_uvr[3 * (size_t)k + 1]; rec.ur = g.obs_uvr[3 * (size_t)k + 2];
        g.pose_rec[pos] = rec;
    }
}

// Co-observation pairs of every block (i <= j) of S: the observations of pose i whose (free) landmark pose j sees too, in pose i's order.
// One wavefront per block.  Pose j's landmark list (ascending: observations are landmark-major) is staged in LDS once and every
// observation of pose i runs a branch-free lower-bound search in it, PAIR_U searches in flight per lane — with the list in HBM the ten
// dependent loads of a search were the whole 44 us of this kernel at C2 (one wave per block: latency, not bandwidth).  Lists longer
// than PAIR_LDS entries are searched in HBM as before.
constexpr int PAIR_LDS = 4096;        // entries of pose j's list per wavefront (16 KB; four wavefronts per workgroup)
constexpr int PAIR_U = 8;
__global__ __launch_bounds__(256) void k_build_pairs(const DeviceGraph g) {
    __shared__ int slist[4 * PAIR_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= g.n_blk) return;
    const int i = g.blk_i[b], j = g.blk_j[b];
    const int sA = g.chunk_ptr[g.pose_chunk_ptr[i]], eA = g.chunk_ptr[g.pose_chunk_ptr[i + 1]];
    const int sB = g.chunk_ptr[g.pose_chunk_ptr[j]], eB = g.chunk_ptr[g.pose_chunk_ptr[j + 1]];
    int base = g.blk_ptr[b];
    const int end = g.blk_ptr[b + 1];
    if (base == end) return;                              // a block that exists for its odometry edge only
    const int nB = eB - sB;
    int top = 1;                                          // largest power of two <= |B| (uniform trip count of the search)
    while (2 * top <= nB) top *= 2;
    int* const mine = slist + wave * PAIR_LDS;
    const bool in_lds = (i != j) && nB <= PAIR_LDS;
    if (in_lds) {
        for (int q = lane; q < nB; q += 64) mine[q] = g.pose_lm[sB + q];
        __builtin_amdgcn_s_waitcnt(0);                    // (one wavefront reads what it wrote: no barrier, the stores only have to have left)
        __builtin_amdgcn_wave_barrier();
    }
    constexpr int U = PAIR_U;
    for (int t0 = sA; t0 < eA; t0 += 64 * U) {
        int k1[U], l[U], lo[U];
        bool cand[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 64 * u + lane;
            k1[u] = 0; l[u] = 0; lo[u] = 0;
            if (t < eA) { k1[u] = VISFS_BA_POSE_SEEDS ? t : g.pose_obs[t]; l[u] = g.pose_lm[t]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cand[u] = (t0 + 64 * u + lane < eA) && !g.pt_fixed[l[u]];
        // lower bound of l in pose j's landmark list, branch-free: lo ends at the first entry >= l
        if (i != j) {
            if (in_lds) {
                for (int step = top; step >= 1; step >>= 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int probe = lo[u] + step - 1;
                        const int v = mine[probe < nB ? probe : nB - 1];
                        if (cand[u] && probe < nB && v < l[u]) lo[u] += step;
                    }
                }
            } else {
                for (int step = top; step >= 1; step >>= 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int probe = lo[u] + step - 1;
                        if (cand[u] && probe < nB && g.pose_lm[sB + probe] < l[u]) lo[u] += step;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool m = cand[u];
            int k2 = k1[u];
            if (i != j) {
                m = m && lo[u] < nB && (in_lds ? mine[lo[u] < nB ? lo[u] : 0] : g.pose_lm[sB + (lo[u] < nB ? lo[u] : 0)]) == l[u];
                if (m) k2 = VISFS_BA_POSE_SEEDS ? sB + lo[u] : g.pose_obs[sB + lo[u]];
            }
            const unsigned long long mask = __ballot(m);
            if (m) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (pos < end) g.blk_pairs[pos] = make_int4(k1[u], k2, l[u], 0);
            }
            base += __popcll(mask);
        }
    }
}

// ================================================================= K5: Schur complement (gather form)
// k_schur_partial: one wavefront per chunk of co-observation pairs of ONE block (i <= j) of the reduced camera matrix, one
// pair per lane and pass:  Hpl_il (Hll_l + lambda I)^-1 Hpl_jl^T  (+ the b_s term on diagonal blocks), reduce-scattered over
// the wave into 42 partial sums.
// The contribution Wa D Wb^T of ONE pair (tile of pose i, tile of pose j, landmark), through the structure of the tiles:
// W = [N ; [Pc]x N] (tile_core), so with P = Na D Nb^T (3x3) the 6x6 result is [P, P Xb^T ; Xa P, Xa P Xb^T] — rows of P
// crossed with Pc_b, columns with Pc_a — and the b_s term is [v ; Pc_a x v] with v = Na (D b_l): about half the fp64 work of
// forming both 6x3 tiles.  A lane without a pair (have = false) produces exact zeros.
// ACC: ADD the contribution to G / gb (a lane's later pairs of a multi-pass chunk) instead of writing it — only the pair's own top
// half (18 values, which its bottom half is built from) is live beside the accumulators, not a second 36 + 6.
template <bool ACC, bool CERES = false>
__device__ __forceinline__ void schur_pair(const DeviceGraph& g, const LinBuf& L, const int4 pr, const bool have, const bool diag, const Rt& Ti, const Rt& Tj,
                                           const double lambda, double G[36], double gb[6]) {
    if (!ACC) {
#pragma unroll
        for (int r = 0; r < 6; ++r) gb[r] = 0.0;
    }
    const double* H = L.Hll + 6 * (size_t)pr.z;
    const double* seeds = VISFS_BA_POSE_SEEDS ? L.pose_pcw : L.obs_pcw;            // pair tiles are pose-major positions / observation ids
    const double2* sa = reinterpret_cast<const double2*>(seeds + 4 * (size_t)pr.x);
    const double2* sb = reinterpret_cast<const double2*>(seeds + 4 * (size_t)pr.y);
    const double2 a0 = have ? sa[0] : make_double2(0.0, 0.0), a1 = have ? sa[1] : make_double2(1.0, 0.0);
    const double2 b0 = (have && !diag) ? sb[0] : a0, b1 = (have && !diag) ? sb[1] : a1;
    const Intrinsics K = intr_of(g);
    const Vec3 pa{ a0.x, a0.y, a1.x }, pb{ b0.x, b0.y, b1.x };
    double Na[9], Nb[9];
    tile_core(Ti.R, pa, a1.y, K, Na);
    tile_core(Tj.R, pb, b1.y, K, Nb);
    double h[6] = { 1.0, 0.0, 0.0, 1.0, 0.0, 1.0 }, B[3] = { 0.0, 0.0, 0.0 };
    if (have) {
        // (the flavour is a template parameter here: as a run-time branch the three dampings cost the VALU-bound gather 3-4 %)
        double a0 = lambda, a1 = lambda, a2 = lambda;
        if (CERES) { a0 = damp_of(g, lambda, H[0], g.s2l, 3 * (size_t)pr.z); a1 = damp_of(g, lambda, H[3], g.s2l, 3 * (size_t)pr.z + 1); a2 = damp_of(g, lambda, H[5], g.s2l, 3 * (size_t)pr.z + 2); }
        h[0] = H[0] + a0; h[1] = H[1]; h[2] = H[2]; h[3] = H[3] + a1; h[4] = H[4]; h[5] = H[5] + a2;
        if (diag) { const double* Bl = L.bl + 3 * (size_t)pr.z; B[0] = Bl[0]; B[1] = Bl[1]; B[2] = Bl[2]; }
    }
    double D[6];
    sym3_inverse(h, D);
    // a landmark without any active edge has Hll = 0 and (Gauss-Newton, lambda = 0) a singular block: its tiles are
    // all zero, so drop the term instead of multiplying 0 by inf
    const bool okD = (D[0] == D[0]) && (fabs(D[0]) <= DBL_MAX) && (D[3] == D[3]) && (fabs(D[3]) <= DBL_MAX) && (D[5] == D[5]) && (fabs(D[5]) <= DBL_MAX);
    if (!okD) { D[0] = D[1] = D[2] = D[3] = D[4] = D[5] = 0.0; }
    double Q[9], P[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {                       // Q = Na D
        Q[3 * r + 0] = Na[3 * r] * D[0] + Na[3 * r + 1] * D[1] + Na[3 * r + 2] * D[2];
        Q[3 * r + 1] = Na[3 * r] * D[1] + Na[3 * r + 1] * D[3] + Na[3 * r + 2] * D[4];
        Q[3 * r + 2] = Na[3 * r] * D[2] + Na[3 * r + 1] * D[4] + Na[3 * r + 2] * D[5];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) P[3 * r + c] = Q[3 * r] * Nb[3 * c] + Q[3 * r + 1] * Nb[3 * c + 1] + Q[3 * r + 2] * Nb[3 * c + 2];   // P = Q Nb^T
    // top half: [P | rows of P crossed with Pc_b]
    double T[18];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double p0 = P[3 * r], p1 = P[3 * r + 1], p2 = P[3 * r + 2];
        T[6 * r + 0] = p0; T[6 * r + 1] = p1; T[6 * r + 2] = p2;
        T[6 * r + 3] = pb.y * p2 - pb.z * p1; T[6 * r + 4] = pb.z * p0 - pb.x * p2; T[6 * r + 5] = pb.x * p1 - pb.y * p0;
    }
#pragma unroll
    for (int q = 0; q < 18; ++q) { if (ACC) G[q] += T[q]; else G[q] = T[q]; }
    // bottom half: Pc_a crossed with the columns of the top half
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double t0 = T[c], t1 = T[6 + c], t2 = T[12 + c];
        const double u0 = pa.y * t2 - pa.z * t1, u1 = pa.z * t0 - pa.x * t2, u2 = pa.x * t1 - pa.y * t0;
        if (ACC) { G[18 + c] += u0; G[24 + c] += u1; G[30 + c] += u2; } else { G[18 + c] = u0; G[24 + c] = u1; G[30 + c] = u2; }
    }
    if (diag) {
        const double v0 = Q[0] * B[0] + Q[1] * B[1] + Q[2] * B[2], v1 = Q[3] * B[0] + Q[4] * B[1] + Q[5] * B[2], v2 = Q[6] * B[0] + Q[7] * B[1] + Q[8] * B[2];
        const double w0 = pa.y * v2 - pa.z * v1, w1 = pa.z * v0 - pa.x * v2, w2 = pa.x * v1 - pa.y * v0;
        if (ACC) { gb[0] += v0; gb[1] += v1; gb[2] += v2; gb[3] += w0; gb[4] += w1; gb[5] += w2; }
        else { gb[0] = v0; gb[1] = v1; gb[2] = v2; gb[3] = w0; gb[4] = w1; gb[5] = w2; }
    }
}

// One wavefront, one chunk (<= 64 pairs of one block, MULTI: <= DeviceGraph::sch_chunk): shared by k_schur_partial and the
// fused small-window kernel.
__device__ __forceinline__ void st_pub(double* p, const double v);
// fin_arrive (round 4, VERDICT r03 item 9 as asked): the wavefront that completes a block's partials finalises the block — defined behind schur_block
template <bool ROLEB>
__device__ __forceinline__ void fin_arrive_at(const DeviceGraph& g, const LinBuf& L, LmState* st, const int b, const int lane);
template <bool MULTI, bool CERES = false>
__device__ __forceinline__ void schur_chunk(const DeviceGraph& g, const LinBuf& L, const int ch, const int lane, const double lambda, const double* __restrict__ pose,
                                            const int4 dsc, const int4 pr) {
    // (dsc = sch_desc[ch] and pr = the lane's first pair come from the caller: neither depends on the LM state, so a kernel can
    // have them in flight while its gate is still being read)
    const int e = dsc.x + lane, e_end = dsc.y;
    const bool diag = (dsc.z == dsc.w);
    const Rt Ti = pose_to_Rt(pose + POSE_STRIDE * dsc.z);
    const Rt Tj = pose_to_Rt(pose + POSE_STRIDE * dsc.w);
    double acc[21];
    double keep0 = 0.0, keep1 = 0.0;
    int off0 = 0, len0 = 21, off1 = 0, len1 = 21;
    const bool have = e < e_end;
    // pair = (tile of pose i, tile of pose j, landmark): every load below depends only on this one
    double G[36], gb[6];
    schur_pair<false, CERES>(g, L, pr, have, diag, Ti, Tj, lambda, G, gb);
    if (MULTI) {
        // chunks of more than 64 pairs: the lane adds its later pairs (e + 64, e + 128, ...) serially, in that fixed order, so one
        // reduce-scatter serves the whole chunk (the cross-lane reduction costs about as much VALU time as a pair product)
        for (int e2 = e + 64; e2 - lane < e_end; e2 += 64) {
            const bool have2 = e2 < e_end;
            const int4 pr2 = have2 ? g.blk_pairs[e2] : make_int4(0, 0, 0, 0);
            schur_pair<true, CERES>(g, L, pr2, have2, diag, Ti, Tj, lambda, G, gb);
        }
    }
    // two halves of 21 sums (block rows 0-2 + b_s 0-2, block rows 3-5 + b_s 3-5): halves the live accumulator registers
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int q = 0; q < 18; ++q) acc[q] = G[18 * half + q];
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[18 + r] = gb[3 * half + r];
        int off = 0, len = 21;
        ReduceScatter<21, 32>::run(acc, lane, off, len);
        if (half == 0) { keep0 = acc[0]; off0 = off; len0 = len; } else { keep1 = acc[0]; off1 = off; len1 = len; }
    }
    // element index inside a half: 0..17 → block entry (row 3*half + idx/6, col idx%6); 18..20 → b_s entry 3*half + (idx-18)
    double* out = g.sch_part + 42 * (size_t)ch;
    if (g.fin_arrive) {       // (read by another wavefront of this launch: write-through)
        if (len0 >= 1) st_pub(out + (off0 < 18 ? off0 : 36 + (off0 - 18)), keep0);
        if (len1 >= 1) st_pub(out + (off1 < 18 ? 18 + off1 : 39 + (off1 - 18)), keep1);
    } else {
        if (len0 >= 1) out[off0 < 18 ? off0 : 36 + (off0 - 18)] = keep0;
        if (len1 >= 1) out[off1 < 18 ? 18 + off1 : 39 + (off1 - 18)] = keep1;
    }
}

template <bool MULTI>
__device__ __forceinline__ void schur_chunk(const DeviceGraph& g, const LinBuf& L, const int ch, const int lane, const double lambda, const double* __restrict__ pose) {
    const int4 dsc = g.sch_desc[ch];
    const int e = dsc.x + lane;
    schur_chunk<MULTI>(g, L, ch, lane, lambda, pose, dsc, e < dsc.y ? g.blk_pairs[e] : make_int4(0, 0, 0, 0));
}

// The pose-major role of the linearisation as a workgroup behind a Schur launch (ROLEB): chunk c of a pose's observations.
template <class Src>
__device__ __forceinline__ void roleb_chunk(const DeviceGraph& g, const LmState* st, const int c, double* redb) {
    if (c >= g.n_chunks) return;
    // chunk -> pose, range, record: none of it depends on the LM state — in flight before the gate
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int a = g.chunk_pose[c];
    const int begin = g.chunk_ptr[c], end = g.chunk_ptr[c + 1];
    const bool mine = begin + tid < end;
    DeviceGraph::PoseRec rec;
    rec.k = 0; rec.l_ok = -1; rec.u = rec.v = rec.ur = 0.0;
    if (mine) rec = g.pose_rec[begin + tid];
    const int ipose = g.free_pose[a];
    if (!st->lin_b_pending || !(st->mode & MODE_TRIAL)) return;
    const int sel = st->sel;
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
    if (mine) pose_obs_terms_rec(g, rec, pose_to_Rt(g.pose[sel] + POSE_STRIDE * ipose), g.pt[sel], intr_of(g), g.inv_pixel_var, g.huber_delta, acc);
    int off = 0, len = 27;
    ReduceScatter<27, 32>::run(acc, lane, off, len);
    if (len >= 1) redb[wave * 27 + off] = acc[0];
    __syncthreads();
    if (Src::batched || !g.fin_arrive) {
        if (tid < 27) L.hpp_part[27 * (size_t)c + tid] = redb[tid] + redb[27 + tid] + redb[54 + tid] + redb[81 + tid];
        return;
    }
    if (wave != 0) return;
    if (tid < 27) st_pub(L.hpp_part + 27 * (size_t)c + tid, redb[tid] + redb[27 + tid] + redb[54 + tid] + redb[81 + tid]);
    fin_arrive_at<true>(g, L, const_cast<LmState*>(st), g.diag_blk[a], lane);
}

// ROLEB (single window, fused speculative unit): the workgroups behind this window's share of the chunk list are the pose-major role of
// the linearisation the previous unit's k_backsub<LINA> left half done (LmState::lin_b_pending): upper triangle of Jx^T (rho' Omega) Jx
// and -Jx^T (rho' Omega) e per chunk of a pose's observations, at the committed estimate, into the current set's hpp_part — read by
// k_schur_finalize / k_small_solve, the launch after this one.
template <bool MULTI, class Src, bool CERES = false, bool ROLEB = false>
__global__ __launch_bounds__(256, MULTI ? 2 : 4) void k_schur_partial(const Src src) {
    const DeviceGraph& g = graph_of(src);
    const LmState* st = state_of(src, g);
    const int lane = threadIdx.x & 63;
    // (role B sits BEHIND the gather in dispatch order: in front of it — measured — it delays the chunk workgroups, which are the launch's
    // critical path, and costs C2 4 %; behind it the launch is 1.5 us longer than the plain gather)
    if (ROLEB) {
        const int first_b = (((g.n_sch + 3) / 4) + 7) / 8 * 8;
        if ((int)blockIdx.x >= first_b) {
            __shared__ double redb[4 * 27];
            roleb_chunk<Src>(g, st, (int)blockIdx.x - first_b, redb);
            return;
        }
    }
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, chunks are sorted by block row, so give
    // every XCD one contiguous slice of the chunk list: the tiles of a block row are then served by ONE 4 MiB L2
    // instead of eight (speed only; any placement is correct).  gridDim.x is a multiple of 8.
    const int nwg = (((g.n_sch + 3) / 4) + 7) / 8 * 8;        // this window's share of the launch (== gridDim.x for a single window)
    const int bx = (int)blockIdx.x;
    if (bx >= nwg) return;
    const int per_xcd = nwg >> 3;
    const int wg = (bx & 7) * per_xcd + (bx >> 3);
    const int ch = wg * 4 + (threadIdx.x >> 6);
    if (ch >= g.n_sch) return;
    // the chunk descriptor and the lane's first pair do not depend on the LM state: fetch them BEFORE the gate, so the gate's own
    // load (a cold L2 round trip at the head of every kernel) overlaps two levels of the index chain instead of preceding them
    const int4 dsc = g.sch_desc[ch];
    const int e0 = dsc.x + lane;
    const int4 pr = e0 < dsc.y ? g.blk_pairs[e0] : make_int4(0, 0, 0, 0);
    if (!(st->mode & MODE_TRIAL)) return;
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    schur_chunk<MULTI, CERES>(g, L, ch, lane, st->lambda, g.pose[st->sel], dsc, pr);
    if (!Src::batched && !CERES && g.fin_arrive) fin_arrive_at<ROLEB>(g, L, const_cast<LmState*>(st), g.sch_blk[ch], lane);
}

// ================================================================= K5, round 4: Schur complement by RUNS OF LANDMARKS
// The pair-list gather above rebuilds both 3x3 tile cores and inverts the landmark block for EVERY co-observation pair (at track
// length 10 a tile is rebuilt 11 times, a landmark's D^-1 formed 55 times per damped solve) and pays a 42-value cross-lane reduction per
// 64 pairs: 581 VALU instructions per wavefront for ~162 useful multiply-adds per pair (profiles/r03_v4_sq_C4R_counters.json); it is the
// arithmetic-bound kernel of the 200-key-frame window and of batched launches.  k_schur_runs turns the loop inside out:
//   * a workgroup owns a RUN of consecutive landmarks (run_lr x run_m of them; landmark ids grow with time, so a run's observations lie
//     in a short span of W poses — the host knows every run's span from the structure summary and picks run_lr so that a sub-batch's
//     observations fit the LDS);
//   * phase 1, one lane per observation: the tile core N (tile_core), the landmark's damped inverse D, Q = N D and the camera-frame
//     point go to LDS ONCE — these are the "LDS-staged Jacobian tiles" — and the (landmark, pose) -> tile slot table is filled;
//   * phase 2, one lane per (block of the run's span, part): the lane walks the landmarks of its part and ACCUMULATES
//     W_a D W_b^T = [P, P Xb^T; Xa P, Xa P Xb^T], P = Q_a N_b^T, in registers — 81 multiply-adds per pair on operands read from LDS,
//     no tile rebuild, no inverse, no cross-lane traffic;
//   * the parts of a block are added in part order through LDS and the run writes ONE partial per block of its span (sch_part, at
//     run_desc.z); k_schur_finalize adds the partials of the runs whose span holds the block, in run order.
// Every sum has a fixed order that depends on the window alone: results are reproducible and independent of any batch.  Per pair the
// operands are the values the gather forms (same functions), only the association of the sums differs.
constexpr unsigned short RUN_NONE = 0xffffu;
constexpr int RUN_RED = 14;                // sums per item and pass of the reduction over the parts (42 = 3 x 14)
// Global-load chain of a workgroup: run_desc / run_k0 / the LM state (level 1) -> EVERYTHING else (level 2): the seeds and indices of the
// thread's (up to two) observations, H_ll and b_l of the sub-batch's landmarks (consecutive ids: addresses follow from the run number), the
// poses of the run's span.  D_l and R_i are formed once per landmark / pose into LDS, the tiles from those.
#ifndef VISFS_BA_RUN_WAVES
#define VISFS_BA_RUN_WAVES 3           // waves per SIMD the kernel is compiled for (A/B builds)
#endif
// barrier that waits for this wave's LDS traffic only (__syncthreads() also drains the global stores of the partials)
#define RUN_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <class Src, bool CERES = false, bool ROLEB = false>
__global__ __launch_bounds__(256, VISFS_BA_RUN_WAVES) void k_schur_runs(const Src src) {
    const DeviceGraph& g = graph_of(src);
    const LmState* st = state_of(src, g);
    extern __shared__ __attribute__((aligned(16))) double run_lds[];
    const int tid = threadIdx.x;
    const int nwg = (g.n_runs + 7) / 8 * 8;                  // this window's share of the launch
    if (ROLEB) {
        if ((int)blockIdx.x >= nwg) { roleb_chunk<Src>(g, st, (int)blockIdx.x - nwg, run_lds); return; }
    }
    const int bx = (int)blockIdx.x;
    if (bx >= nwg) return;
    // XCD-aware: consecutive runs (neighbouring poses, the same Hll / seeds lines) on one XCD's L2, as the gather does
    const int per_xcd = nwg >> 3;
    const int r = (bx & 7) * per_xcd + (bx >> 3);
    if (r >= g.n_runs) return;
#ifdef VISFS_BA_STAMPS
#define RUN_STAMP(slot) do { if (tid == 0 && r == g.stamp_wg) g.stamps[96 + (slot)] = wall_clock64(); } while (0)
#else
#define RUN_STAMP(slot) do { } while (0)
#endif
    RUN_STAMP(0);
    const int LR = g.run_lr, M = g.run_m;
    const int4 rd = g.run_desc[r];                            // (lowest pose index, span W, first partial slot, -) — before the gate
    int k_next = g.run_k0[r * M];                             // first observation of the first sub-batch
    const int L0 = r * LR * M, L1 = min(g.Nl, L0 + LR * M);
    if (!(st->mode & MODE_TRIAL)) return;
    const int ipmin = rd.x, W = rd.y;
    const int nb = W * (W + 1) / 2;
    if (nb == 0) return;                                     // a run without observations: no partial slot, nothing reads one
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    const double lambda = st->lambda;
    const double* __restrict__ pose = g.pose[st->sel];
    const Intrinsics K = intr_of(g);
    // ---- LDS: tiles (reused for the reduction over the parts), b_l / D_l of the sub-batch, R of the span's poses, slot table, (i, j) of the local blocks
    const int cap = max(g.run_cap, (256 * RUN_RED + RUN_TILE - 1) / RUN_TILE);
    double* tiles = run_lds;                                                  // [cap][21]
    double* sbl = tiles + (size_t)cap * RUN_TILE;                             // [LR][3]
    double* sD = sbl + 3 * LR;                                                // [LR][6]  (D[0] = NaN: no usable inverse)
    double* sR = sD + 6 * LR;                                                 // [run_wmax][9]
    unsigned short* slot = reinterpret_cast<unsigned short*>(sR + 9 * g.run_wmax);   // [LR][W]
    unsigned short* bij = slot + ((LR * g.run_wmax + 3) & ~3);                // [nb] i | j << 8
    for (int t = tid; t < nb; t += 256) {
        int i = 0, off = 0;
        while (t >= off + (W - i)) { off += W - i; ++i; }
        bij[t] = (unsigned short)(i | ((i + t - off) << 8));
    }
    if (tid >= 64 && tid < 64 + W) {                          // R of pose ipmin + (tid - 64), wave 1
        const Rt T = pose_to_Rt(pose + POSE_STRIDE * (ipmin + tid - 64));
        double* o = sR + 9 * (tid - 64);
        o[0] = T.R.m00; o[1] = T.R.m01; o[2] = T.R.m02; o[3] = T.R.m10; o[4] = T.R.m11; o[5] = T.R.m12; o[6] = T.R.m20; o[7] = T.R.m21; o[8] = T.R.m22;
    }
    // blocks of the span in rounds of 256 items; one round whenever the span has <= 22 poses (then P parts share a block)
    for (int blk0 = 0; blk0 < nb; blk0 += 256) {
        const int nbr = min(256, nb - blk0);
        const int P = nb <= 256 ? max(1, 256 / nb) : 1;
        const int blk = tid / P, part = tid - blk * P;
        const bool item = blk < nbr;
        __syncthreads();                                     // bij, sR complete (first round); the previous round's reduction has left the tiles region
        RUN_STAMP(1);
        int bi = 0, bj = 0;
        bool live = false, diag = false;
        if (item) {
            const unsigned v = bij[blk0 + blk];
            bi = (int)(v & 0xffu); bj = (int)(v >> 8);
            live = g.pose_free[ipmin + bi] >= 0 && g.pose_free[ipmin + bj] >= 0;
            diag = bi == bj;
        }
        double G[36], gb[6];
#pragma unroll
        for (int q = 0; q < 36; ++q) G[q] = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) gb[q] = 0.0;
        if (blk0 > 0) k_next = g.run_k0[r * M];
        for (int m = 0; m < M; ++m) {
            const int Lb0 = L0 + m * LR, Lb1 = min(L1, Lb0 + LR);
            if (Lb0 >= Lb1) break;
            const int nl = Lb1 - Lb0;
            const int k0 = k_next;
            k_next = g.run_k0[r * M + m + 1];
            const int nt = k_next - k0;
            // ---- phase 1a: every global load of the sub-batch, none depending on another
            double2 s0[2], s1[2];
            int ol[2], oi[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tid + 256 * u;
                s0[u] = make_double2(0.0, 0.0); s1[u] = make_double2(0.0, 0.0); ol[u] = Lb0; oi[u] = ipmin;
                if (t < nt) {
                    const double2* seed = reinterpret_cast<const double2*>(L.obs_pcw + 4 * (size_t)(k0 + t));
                    s0[u] = seed[0]; s1[u] = seed[1];
                    ol[u] = g.obs_pt[k0 + t]; oi[u] = g.obs_pose[k0 + t];
                }
            }
            double Hh[6] = { 1.0, 0.0, 0.0, 1.0, 0.0, 1.0 };
            if (tid < nl) {
                const double* H = L.Hll + 6 * (size_t)(Lb0 + tid);
#pragma unroll
                for (int q = 0; q < 6; ++q) Hh[q] = H[q];
            }
            __syncthreads();                                 // the previous sub-batch's readers are done with tiles / slot / sbl / sD
            RUN_STAMP(2);
            for (int t = tid; t < nl * W; t += 256) slot[t] = RUN_NONE;
            for (int t = tid; t < 3 * nl; t += 256) sbl[t] = L.bl[3 * (size_t)Lb0 + t];
            if (tid < nl) {
                const int l = Lb0 + tid;
                double a0 = lambda, a1 = lambda, a2 = lambda;
                if (CERES) { a0 = damp_of(g, lambda, Hh[0], g.s2l, 3 * (size_t)l); a1 = damp_of(g, lambda, Hh[3], g.s2l, 3 * (size_t)l + 1); a2 = damp_of(g, lambda, Hh[5], g.s2l, 3 * (size_t)l + 2); }
                const double h[6] = { Hh[0] + a0, Hh[1], Hh[2], Hh[3] + a1, Hh[4], Hh[5] + a2 };
                double D[6];
                sym3_inverse(h, D);
                // (a singular damped block — Gauss-Newton, lambda = 0, a landmark without active edges — drops out as in the gather)
                const bool okD = (D[0] == D[0]) && (fabs(D[0]) <= DBL_MAX) && (D[3] == D[3]) && (fabs(D[3]) <= DBL_MAX) && (D[5] == D[5]) && (fabs(D[5]) <= DBL_MAX);
                double* o = sD + 6 * tid;
                o[0] = okD ? D[0] : __builtin_nan(""); o[1] = D[1]; o[2] = D[2]; o[3] = D[3]; o[4] = D[4]; o[5] = D[5];
            }
            __syncthreads();
            RUN_STAMP(3);
            // ---- phase 1b: the tiles of this thread's observations, from registers and LDS
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tid + 256 * u;
                if (t >= nt || s1[u].y == 0.0) continue;      // no tile: inactive edge, fixed pose or fixed landmark
                const double* D = sD + 6 * (ol[u] - Lb0);
                const double D0 = D[0];
                if (!(D0 == D0)) continue;
                const double* Rr = sR + 9 * (oi[u] - ipmin);
                const Mat3 R{ Rr[0], Rr[1], Rr[2], Rr[3], Rr[4], Rr[5], Rr[6], Rr[7], Rr[8] };
                const Vec3 pc{ s0[u].x, s0[u].y, s1[u].x };
                double N[9];
                tile_core(R, pc, s1[u].y, K, N);
                double* o = tiles + (size_t)t * RUN_TILE;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {                // Q = N D
                    o[3 * rr + 0] = N[3 * rr] * D0 + N[3 * rr + 1] * D[1] + N[3 * rr + 2] * D[2];
                    o[3 * rr + 1] = N[3 * rr] * D[1] + N[3 * rr + 1] * D[3] + N[3 * rr + 2] * D[4];
                    o[3 * rr + 2] = N[3 * rr] * D[2] + N[3 * rr + 1] * D[4] + N[3 * rr + 2] * D[5];
                }
#pragma unroll
                for (int q = 0; q < 9; ++q) o[9 + q] = N[q];
                o[18] = pc.x; o[19] = pc.y; o[20] = pc.z;
                slot[(ol[u] - Lb0) * W + (oi[u] - ipmin)] = (unsigned short)t;
            }
            __syncthreads();
            RUN_STAMP(4);
            // ---- phase 2: this item's block over the landmarks of its part
            if (live) {
                for (int ll = part; ll < nl; ll += P) {
                    const unsigned sa = slot[ll * W + bi], sb = slot[ll * W + bj];
                    if (sa == RUN_NONE || sb == RUN_NONE) continue;
                    const double* ta = tiles + (size_t)sa * RUN_TILE;
                    const double* tb = tiles + (size_t)sb * RUN_TILE;
                    double Q[9], Nb[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) { Q[q] = ta[q]; Nb[q] = tb[9 + q]; }
                    const Vec3 pa{ ta[18], ta[19], ta[20] }, pb{ tb[18], tb[19], tb[20] };
                    double T[18];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        const double p0 = Q[3 * rr] * Nb[0] + Q[3 * rr + 1] * Nb[1] + Q[3 * rr + 2] * Nb[2];       // P = Q Nb^T
                        const double p1 = Q[3 * rr] * Nb[3] + Q[3 * rr + 1] * Nb[4] + Q[3 * rr + 2] * Nb[5];
                        const double p2 = Q[3 * rr] * Nb[6] + Q[3 * rr + 1] * Nb[7] + Q[3 * rr + 2] * Nb[8];
                        T[6 * rr + 0] = p0; T[6 * rr + 1] = p1; T[6 * rr + 2] = p2;
                        T[6 * rr + 3] = pb.y * p2 - pb.z * p1; T[6 * rr + 4] = pb.z * p0 - pb.x * p2; T[6 * rr + 5] = pb.x * p1 - pb.y * p0;
                    }
#pragma unroll
                    for (int q = 0; q < 18; ++q) G[q] += T[q];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {               // bottom half: Pc_a crossed with the columns of the top half, two fused multiply-adds each
                        const double t0 = T[c], t1 = T[6 + c], t2 = T[12 + c];
                        G[18 + c] = fma(-pa.z, t1, fma(pa.y, t2, G[18 + c]));
                        G[24 + c] = fma(-pa.x, t2, fma(pa.z, t0, G[24 + c]));
                        G[30 + c] = fma(-pa.y, t0, fma(pa.x, t1, G[30 + c]));
                    }
                    if (diag) {
                        const double B0 = sbl[3 * ll], B1 = sbl[3 * ll + 1], B2 = sbl[3 * ll + 2];
                        const double v0 = Q[0] * B0 + Q[1] * B1 + Q[2] * B2, v1 = Q[3] * B0 + Q[4] * B1 + Q[5] * B2, v2 = Q[6] * B0 + Q[7] * B1 + Q[8] * B2;
                        gb[0] += v0; gb[1] += v1; gb[2] += v2;
                        gb[3] += pa.y * v2 - pa.z * v1; gb[4] += pa.z * v0 - pa.x * v2; gb[5] += pa.x * v1 - pa.y * v0;
                    }
                }
            }
        }
        // ---- the parts of a block added in part order: the 42 sums of an item go through the tiles region in three thirds of 14 (entry
        // e of the partial: 0..35 the block row-major, 36..41 b_s); LDS-only barriers — the global stores of a third stay in flight
        double* out = g.sch_part + 42 * ((size_t)rd.z + blk0);
        RUN_STAMP(5);
#pragma unroll
        for (int third = 0; third < 3; ++third) {
            RUN_SYNC();
            RUN_STAMP(6 + 2 * third);
            if (item) {
                double* o = tiles + (size_t)tid * RUN_RED;
#pragma unroll
                for (int q = 0; q < RUN_RED; ++q) { const int e = RUN_RED * third + q; o[q] = e < 36 ? G[e < 36 ? e : 0] : gb[e >= 36 ? e - 36 : 0]; }
            }
            RUN_SYNC();
            for (int t = tid; t < RUN_RED * nbr; t += 256) {
                const int b2 = t / RUN_RED, q = t - RUN_RED * b2;
                const double* src2 = tiles + (size_t)(b2 * P) * RUN_RED + q;
                double v = src2[0];
                for (int pp = 1; pp < P; ++pp) v += src2[RUN_RED * pp];
                out[42 * (size_t)b2 + RUN_RED * third + q] = v;
            }
            RUN_STAMP(7 + 2 * third);
        }
    }
}

// k_schur_finalize: one wavefront per stored block:
//   S_ij = Hpp_ij (+lambda I on the diagonal) - sum of the chunk partials;  diagonal waves also produce
//   b_s_i = b_p_i - ..., Hpp_ii / b_p_i (for computeScale) and Minv_i = S_ii^-1 (block-Jacobi preconditioner).
// Poses without any active edge are outside g2o's active set: their block is pinned to I (dx = 0).
// It also clears the hand-off words of the persistent PCG that follows (one zeroing per damped solve).
// 36-lane Gauss-Jordan inverse of an SPD 6x6 block (no pivoting needed): lane 6 r + c holds entry (r, c) going in and the entry
// of the inverse coming out; lanes 36..63 pass anything.
__device__ __forceinline__ double gauss_jordan_6x6(const double val, const int lane) {
    const int r = lane / 6, c = lane % 6;
    double a = (lane < 36) ? val : 0.0;
    double v = (lane < 36 && r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int rr = lane < 36 ? r : 0, cc = lane < 36 ? c : 0;
        const double p = __shfl(a, k * 6 + k, 64);
        const double rk_a = __shfl(a, k * 6 + cc, 64);
        const double rk_v = __shfl(v, k * 6 + cc, 64);
        const double ck = __shfl(a, rr * 6 + k, 64);
        const double ip = 1.0 / p;
        if (rr == k) { a = rk_a * ip; v = rk_v * ip; }
        else { a -= ck * (rk_a * ip); v -= ck * (rk_v * ip); }
    }
    return v;
}

// One wavefront, one stored block of S: shared by k_schur_finalize and the fused small-window kernel.
// (bd, be = blk_desc[2 b], blk_desc[2 b + 1] come from the caller: they do not depend on the LM state, so a kernel can have them in
// flight while its gate is still being read)
// k_schur_runs' partials of one block: the runs whose pose span holds the block, in run order, dealt over the `nw` waves of the caller
// (wave w takes the hits whose ordinal is w mod nw and adds them in ascending order; the caller adds the waves' sums in wave order).
// bdx = first candidate run | count << 20, bdy = pose index of i | pose index of j << 16 (blk_desc).  Lane = run for the descriptors (one
// coalesced load per 64 runs); a hit's partial slot is broadcast with a readlane; eight loads in flight.
__device__ __forceinline__ double run_partial_sum(const DeviceGraph& g, const int bdx, const int bdy, const int lane, const int wave, const int nw) {
    const int r0 = bdx & 0xfffff, r1 = r0 + (int)((unsigned)bdx >> 20);
    const int pi = bdy & 0xffff, pj = (int)((unsigned)bdy >> 16);
    double part = 0.0;
    int ord = 0, n = 0;
    int s8[8];
    auto flush = [&]() {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (u < n && lane < 42) ? g.sch_part[42 * (size_t)s8[u] + lane] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < n) part += v8[u];
        n = 0;
    };
    for (int base = r0; base < r1; base += 64) {
        const int rr = base + lane;
        const int4 rd = rr < r1 ? g.run_desc[rr] : make_int4(0, 0, 0, 0);
        const int li = pi - rd.x, lj = pj - rd.x;
        const bool hit = rr < r1 && li >= 0 && lj < rd.y;
        const int idx = rd.z + li * rd.y - (li * (li - 1)) / 2 + (lj - li);
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int bit = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const int sidx = __builtin_amdgcn_readlane(idx, bit);
            if ((ord++ % nw) != wave) continue;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u == n) s8[u] = sidx;
            if (++n == 8) flush();
        }
    }
    if (n > 0) flush();
    return part;
}

// PUB (the finalisation as the prologue of the k_pcg1 launch, round 4): what the PCG waves of the SAME launch read — S, b_s, Minv — is
// stored write-through (agent-scope relaxed atomic stores: sc1), the wave drains its stores and lane 0 publishes the unit's tag in
// fin_flag[b] (cdna_hip_programming.md §6 Guideline 16, recipe R1); the granules are NOT zeroed here (the PCG waves may already be
// publishing: k_reset zeroes them once per optimise call and the hand-off tags carry the unit).
__device__ __forceinline__ void st_pub(double* p, const double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool PUB = false, bool ARR = false>
__device__ __forceinline__ void schur_block(const DeviceGraph& g, const LinBuf& L, LmState* st, const int b, const int lane, const int4 bd, const int4 be, const double part_in = 0.0,
                                            const unsigned pub_tag = 0u) {
    if (!PUB) {   // zero the granules: n_blk >= Npf waves x 64 lanes cover 4 * 6 Npf words in one pass.
        // CONTRACT with the persistent PCG (k_pcg / k_pcg1) that follows: (1) EVERY block's wave runs this loop — the words are dealt
        // over all n_blk waves (stride n_blk * 64), so a kernel that calls schur_block for a subset of the blocks clears only a
        // subset of the words; (2) the PCG relies on it: a hand-off tag is just the iteration number, so a word that keeps the
        // previous damped solve's value for iteration e is ACCEPTED by a consumer that polls before the producer's store lands.  The
        // waves' redundant recurrences then differ in the last bits, they leave the loop in different iterations, and the others
        // spin for a granule that is never published (LmState::pcg_timeout -> VISFS_BA_ERR_DEVICE).  That is what the dropped DIP
        // variant of round 2 ran into (profiles/r02_dip_variant.log: C3 and the batched runs, whose waves start further apart;
        // DESIGN.md §7).  Whoever moves the finalisation elsewhere must clear ALL 4 * 6 Npf + Npf words before the PCG launch — and
        // b_s, Minv of ALL rows must be complete at that kernel boundary too: every PCG wave reads all of them in its set-up.
        const int nwords = 4 * 6 * g.Npf + g.Npf;           // q granules of both parities + one placement word per block row
        for (int w = b * 64 + lane; w < nwords; w += g.n_blk * 64) g.granules[w] = 0ull;
    }
    const double lambda = st->lambda;
    const int i = be.x, j = be.y;
    const bool diag = (i == j);
    const int r = lane / 6, c = lane % 6;    // meaningful for lane < 36
    double part = part_in;
    if (g.n_runs > 0) {
        // (k_schur_runs' partials: summed by run_partial_sum on the four waves of the block's workgroup, handed in)
    } else if (lane < 42) {
        // (unroll 8 measured slower than 4: a C2 block has ~5 two-pass chunks, most of them would run in the remainder loop)
#pragma unroll 4
        for (int ch = bd.x; ch < bd.y; ++ch) part += ARR ? ld_coherent(g.sch_part + 42 * (size_t)ch + lane) : g.sch_part[42 * (size_t)ch + lane];
    }
    if (!diag) {
        if (lane < 36) {
            double base = 0.0;
            for (int n = bd.z; n < bd.w; ++n) {
                const int code = g.blk_odo[n];
                base += L.odo_blk[120 * (size_t)(code >> 1) + 72 + ((code & 1) ? (c * 6 + r) : lane)];
            }
            if (PUB) st_pub(g.S + 36 * (size_t)b + lane, base - part); else g.S[36 * (size_t)b + lane] = base - part;
            if (!PUB && g.pcg_cu) {                           // k_pcg_cu reads S by scalar row: entry (r, c) belongs to row 6 i + r and, transposed, to row 6 j + c
                const int sl = g.blk_slot[b];
                g.S_rows[((size_t)(sl & 255) * 6 + c) * g.cu_T + 6 * i + r] = base - part;
                g.S_rows[((size_t)(sl >> 8) * 6 + r) * g.cu_T + 6 * j + c] = base - part;
            }
        }
        if (PUB) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(g.fin_flag + b, pub_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const double hv = (lane < 42) ? hpp_entry_r<ARR>(g, L, lane, be.z, be.w, bd.z, bd.w) : 0.0;
    const bool on_diag = lane < 36 && r == c;
    const unsigned long long nz = __ballot(on_diag && hv != 0.0);
    const bool pin = (nz == 0ull);
    double val = 0.0;
    if (lane < 36) {
        val = pin ? (r == c ? 1.0 : 0.0) : (hv + (r == c ? damp_of(g, lambda, hv, g.s2p, 6 * (size_t)i + r) : 0.0) - part);
        if (PUB) st_pub(g.S + 36 * (size_t)b + lane, val); else g.S[36 * (size_t)b + lane] = val;
        if (!PUB && g.pcg_cu) g.S_rows[((size_t)(g.blk_slot[b] & 255) * 6 + c) * g.cu_T + 6 * i + r] = val;
        g.Hpp[36 * (size_t)i + lane] = hv;
    } else if (lane < 42) {
        g.bp[6 * (size_t)i + (lane - 36)] = hv;
        const double bsv = pin ? 0.0 : (hv - part);
        if (PUB) st_pub(g.bs + 6 * (size_t)i + (lane - 36), bsv); else g.bs[6 * (size_t)i + (lane - 36)] = bsv;
    }
    const double v = gauss_jordan_6x6(val, lane);
    if (lane < 36) { if (PUB) st_pub(g.Minv + 36 * (size_t)i + lane, v); else g.Minv[36 * (size_t)i + lane] = v; }
    if (b == 0 && lane == 0) {
        st->pcg_res_in = st->pcg_residual;
        st->n_active[1] += 1;
        if (st->mode & MODE_LIN) st->n_active[0] += 1;
    }
    if (PUB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(g.fin_flag + b, pub_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void schur_block(const DeviceGraph& g, const LinBuf& L, LmState* st, const int b, const int lane) {
    // (the fused small-window kernel: pair-list gather only)
    schur_block<false>(g, L, st, b, lane, g.blk_desc[2 * b], g.blk_desc[2 * b + 1]);     // (first chunk, last + 1, first odometry entry, last + 1), (i, j, first pose-major chunk of i, last + 1)
}

// fin_arrive: a wavefront of k_schur_partial has just stored (write-through) one of block b's partials — a gather chunk's, or, for a
// diagonal block in a launch that carries the pose-major role, a pose chunk's.  It drains its stores and counts itself in; the one that
// completes the count (gather chunks + the pose chunks that really ran: lin_b_pending) finalises the block right there — the sums in
// index order as k_schur_finalize forms them, every partial read past the caches — and leaves the counter at zero for the next launch.
// Blocks finish at different times, so all but the last finalisations hide behind chunks that are still being gathered; the launch of
// k_schur_finalize (5.6 us + a boundary at C2) is gone.  Contract kept: every block's wave clears its share of the PCG's hand-off words,
// and S, b_s, Minv of all rows are complete at this kernel's end.
template <bool ROLEB>
__device__ __forceinline__ void fin_arrive_at(const DeviceGraph& g, const LinBuf& L, LmState* st, const int b, const int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0u;
    if (lane == 0) old = __hip_atomic_fetch_add(g.fin_cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    const int e = g.fin_exp[b];
    const unsigned expect = (unsigned)(e & 0xffff) + ((ROLEB && st->lin_b_pending) ? (unsigned)(e >> 16) : 0u);
    if (old + 1u != expect) return;
    if (lane == 0) g.fin_cnt[b] = 0u;
    schur_block<false, true>(g, L, st, b, lane, g.blk_desc[2 * b], g.blk_desc[2 * b + 1]);
}

// RUNS (the Schur complement came from k_schur_runs): one WORKGROUP per stored block — a block collects one partial per run whose span
// holds it (tens, where the gather left a handful of chunk partials), so the four waves each add a quarter of them and wave 0 finishes.
template <class Src, bool RUNS = false>
__global__ __launch_bounds__(256) void k_schur_finalize(const Src src) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = RUNS ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (b >= g.n_blk) return;
    // the block descriptors first, the gate after: one cold-L2 round trip instead of two at the head of the kernel
    const int4 bd = g.blk_desc[2 * b], be = g.blk_desc[2 * b + 1];
    if (!(st->mode & MODE_TRIAL)) return;
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    if (RUNS) {
        __shared__ double sp[4 * 42];
        const double mine = run_partial_sum(g, bd.x, bd.y, lane, wave, 4);
        if (lane < 42) sp[42 * wave + lane] = mine;
        __syncthreads();
        if (wave != 0) return;
        const double part = lane < 42 ? ((sp[lane] + sp[42 + lane]) + sp[84 + lane]) + sp[126 + lane] : 0.0;
        schur_block<false>(g, L, st, b, lane, bd, be, part);
    } else schur_block<false>(g, L, st, b, lane, bd, be);
}

// ================================================================= K6: block-Jacobi PCG on S, persistent
// [g2o-upstream] LinearSolverPCG::solve: x0 = 0, tolerance 1e-6 on r^T M^-1 r, maxIter = rows, absolute
// tolerance carried in _residual between the solves of one optimize() call.
//
// One launch per damped solve; one workgroup (4 waves) per block row of S.  Every workgroup keeps the
// full vectors r, d, q, s in LDS and performs the (tiny) vector recurrences redundantly and bitwise identically,
// so all workgroups take the same branch at every convergence test.  Only q = S d is distributed: row i computes
// its six entries and publishes them as twelve 8-byte granules {epoch:32 | half of the double:32} with one
// write-through store each; every workgroup then sweeps all granules of the iteration until their tags match
// (cdna_hip_programming.md §6 Guideline 16, form R2: the data is the flag — no fence, no separate flag word).
// Granules are double-buffered on the iteration parity and zeroed by k_schur_finalize before every solve (schur_block: every block's
// wave clears its share — see the contract there; a tag is only the iteration number, stale words of the previous solve would match).
// Residency: grid = Npf <= 256 workgroups of 4 waves: one per CU always fits, so the grid is co-resident on an otherwise
// idle device; concurrent windows (visfs_ba_solve_batch) are limited so that the sum of their grids stays <= 256.
// The XCD (accelerator complex die) this wave runs on: HW_REG_XCC_ID (hardware register 20), bits 3:0.
__device__ __forceinline__ unsigned xcc_id() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
#else
    return 0u;
#endif
}
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// BPL = 6-blocks owned by each lane of wave 0 (ceil(Npf / 64)); MREG: the lane keeps its Minv block in registers (BPL == 1).
// WIDE (more than 64 free poses; BPL = 1, MREG): the vector recurrences run on ALL four waves, thread a owns 6-block a (r, d and
// its Minv block in registers), and the two dot products of an iteration are workgroup reductions (wave butterfly + four LDS
// partials in fixed order) — with wave 0 alone owning up to four blocks per lane and Minv in LDS the vector step cost 5 us
// of an 8 us iteration at C4 (199 block rows) and the set-up 8.5 us.
// MULTIROW (more than 256 free poses, WIDE with BPL = 2 or 4 blocks per thread, Minv read from HBM): a workgroup owns
// DeviceGraph::pcg_rows_per_wg consecutive block rows, so that the grid never exceeds one workgroup per CU (co-residency).
template <int BPL, bool MREG, class Src, bool WIDE, bool MULTIROW = false>
__global__ __launch_bounds__(256) void k_pcg(const Src src) {
    static_assert(!MREG || BPL == 1, "Minv in registers: one block per owner");
    static_assert(!MULTIROW || WIDE, "several rows per workgroup only in the WIDE form");
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    if (!(st->mode & MODE_TRIAL)) return;
    const int R = MULTIROW ? g.pcg_rows_per_wg : 1;   // block rows of S per workgroup
    if ((int)blockIdx.x * R >= g.Npf) return;         // a batched launch is sized for the largest window
#ifdef VISFS_BA_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == (unsigned)g.stamp_wg) g.stamps[127] = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n6 = 6 * g.Npf;
    // (one block row per workgroup wherever the grid fits the device: more rows per workgroup measured slower)
    const int i0 = blockIdx.x * R, i1 = (i0 + R < g.Npf) ? i0 + R : g.Npf;
    constexpr int OWN_STRIDE = WIDE ? 256 : 64;                       // distance between the 6-blocks one owner thread holds
    double* sd = smem;                                                // d (every workgroup holds the full vector)
    double* sq = smem + n6;                                           // q of the current iteration
    double* sP = smem + 2 * n6;                                       // [32] scalars ([8..15]: partials of the WIDE reductions)
    double* sQ = sP + 32;                                             // [R][4][8] per-wave partial rows of q
    double* ss = sQ + 32 * R;                                         // [n6] s = Minv r, only when BPL > 1 (else registers)
    double* sM = ss + (BPL > 1 ? n6 : 0);                             // [Npf][36] when pcg_lds_minv
    double* sS = sM + (g.pcg_lds_minv ? 36 * g.Npf : 0);              // [R][max_row][36] when pcg_lds_srow
    const double* Minv = g.pcg_lds_minv ? sM : g.Minv;
    int* sCol = reinterpret_cast<int*>(sS + (g.pcg_lds_srow ? 36 * (size_t)R * g.pcg_max_row : 0));   // [R][max_row]
    int* sCode = sCol + R * g.pcg_max_row;
    const int MR = g.pcg_max_row;
    const int blk0 = WIDE ? tid : lane;                               // first 6-block this thread owns in the vector recurrences
    const bool vec = WIDE || wave == 0;                               // this thread takes part in them
    // sum over the owners: one wave (butterfly) or the workgroup (butterfly, four LDS partials added in fixed order); `slot`
    // alternates so that a partial is never overwritten before every wave has read it (one barrier per reduction)
    auto owners_sum = [&](const double v, const int slot) -> double {
        const double ws = wave_sum(v);
        if (!WIDE) return ws;
        double* red4 = sP + 8 + 4 * slot;
        if (lane == 0) red4[wave] = ws;
        __syncthreads();
        return ((red4[0] + red4[1]) + red4[2]) + red4[3];
    };
    for (int li = 0; li < i1 - i0; ++li) {
        const int rb = g.row_ptr[i0 + li], nb = g.row_ptr[i0 + li + 1] - rb;
        for (int n = tid; n < nb; n += 256) { sCol[li * MR + n] = g.row_col[rb + n]; sCode[li * MR + n] = g.row_blk[rb + n]; }
    }
    if (g.pcg_lds_minv && !MREG) {
#pragma unroll 4
        for (int t = tid; t < 36 * g.Npf; t += 256) sM[t] = g.Minv[t];
    }
    if (tid == 0) sP[2] = 0.0;                                         // hand-off timeout flag of the workgroup
    // ---- the owners of the vector recurrences (wave 0: lane = block; WIDE: thread = block).  r = b ; d = M^-1 r ; dn = r.d   (fixed order)
    double rr_[BPL][6], dd_[BPL][6], xx_[BPL][6], mm_[MREG ? 36 : 1];
#pragma unroll
    for (int k = 0; k < BPL; ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) xx_[k][c] = 0.0;
    double dn = 0.0, d0 = 0.0;
    if (vec) {
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const int a = blk0 + OWN_STRIDE * k;
            const bool own = a < g.Npf;
#pragma unroll
            for (int c = 0; c < 6; ++c) rr_[k][c] = own ? g.bs[6 * a + c] : 0.0;
            if (MREG) {
#pragma unroll
                for (int q = 0; q < 36; ++q) mm_[q] = own ? g.Minv[36 * (size_t)a + q] : 0.0;
            }
        }
        if (!MREG) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
        // (for !MREG the Minv copy above is completed by the barrier below before it is read)
    }
    __syncthreads();
    {
        double part = 0.0;
        if (vec) {
#pragma unroll
            for (int k = 0; k < BPL; ++k) {
                const int a = blk0 + OWN_STRIDE * k;
                const bool own = a < g.Npf;
#pragma unroll
                for (int r6 = 0; r6 < 6; ++r6) {
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c) v += (MREG ? mm_[6 * r6 + c] : (own ? Minv[36 * a + 6 * r6 + c] : 0.0)) * rr_[k][c];
                    dd_[k][r6] = v;
                    part += rr_[k][r6] * v;
                    if (own) sd[6 * a + r6] = v;
                }
            }
        }
        if (vec) {                                                     // (WIDE: every thread, so the barrier inside is uniform)
            dn = owners_sum(part, 0);
            d0 = 1e-6 * dn;
            const double res_in = st->pcg_res_in;
            if (res_in > 0.0 && res_in > d0) d0 = res_in;
            if (tid == 0) { sP[0] = dn; sP[1] = d0; }
        }
    }
    if (g.pcg_lds_srow) {
        // own block rows; transposed blocks are stored transposed so that the mat-vec reads every block row-major.
        // The block codes come from LDS, so the S loads do not wait on other global loads.
        for (int li = 0; li < i1 - i0; ++li) {
            const int nb = g.row_ptr[i0 + li + 1] - g.row_ptr[i0 + li];
#pragma unroll 4
            for (int t = tid; t < 36 * nb; t += 256) {
                const int n = t / 36, q = t - 36 * n, code = sCode[li * MR + n];
                const double* Sb = g.S + 36 * (size_t)(code >> 1);
                sS[36 * (size_t)(li * MR) + t] = (code & 1) ? Sb[(q % 6) * 6 + q / 6] : Sb[q];
            }
        }
    }
    __syncthreads();
    dn = sP[0]; d0 = sP[1];
    int iter = 0;
    bool timeout = false;
    const unsigned spin_limit = g.fault_pcg ? (1u << 10) : (1u << 22);
#ifdef VISFS_BA_STAMPS
#define PCG_STAMP(slot) do { if (tid == 0 && blockIdx.x == (unsigned)g.stamp_wg && (slot) < 126) g.stamps[(slot)] = wall_clock64(); } while (0)
    PCG_STAMP(0);
#else
#define PCG_STAMP(slot) do { } while (0)
#endif
    while (true) {
        if (dn <= d0 || iter >= n6 || !(dn == dn)) break;
        // ---- owned block rows of q = S d: 32 slots (8 per wave) x 8 lanes (6 rows used), fixed-order partials
        // lane = 8 * row + slot-in-wave: the eight slots of a wave differ in lane bits 0..2, so their sum is three in-row DPP
        // exchanges (the former layout, slot in the high bits, needed two LDS-crossbar permutes per iteration); same pairing order
        const int slot = (tid >> 6) * 8 + (tid & 7), rr = (tid & 63) >> 3;
        for (int li = 0; li < i1 - i0; ++li) {
            const int nb = g.row_ptr[i0 + li + 1] - g.row_ptr[i0 + li];
            double acc = 0.0;
            if (rr < 6) {
                for (int n = slot; n < nb; n += 32) {
                    const double* dj = sd + 6 * sCol[li * MR + n];
                    if (g.pcg_lds_srow) {
                        const double* Sr = sS + 36 * (size_t)(li * MR + n) + 6 * rr;
                        acc += Sr[0] * dj[0] + Sr[1] * dj[1] + Sr[2] * dj[2] + Sr[3] * dj[3] + Sr[4] * dj[4] + Sr[5] * dj[5];
                    } else {
                        const int code = sCode[li * MR + n];
                        const double* Sb = g.S + 36 * (size_t)(code >> 1);
                        if (code & 1) acc += Sb[rr] * dj[0] + Sb[6 + rr] * dj[1] + Sb[12 + rr] * dj[2] + Sb[18 + rr] * dj[3] + Sb[24 + rr] * dj[4] + Sb[30 + rr] * dj[5];
                        else { const double* Sr = Sb + 6 * rr; acc += Sr[0] * dj[0] + Sr[1] * dj[1] + Sr[2] * dj[2] + Sr[3] * dj[3] + Sr[4] * dj[4] + Sr[5] * dj[5]; }
                    }
                }
            }
            acc += xor_lane<1>(acc);
            acc += xor_lane<2>(acc);
            acc += xor_lane<4>(acc);
            if ((lane & 7) == 0) sQ[li * 32 + wave * 8 + (lane >> 3)] = acc;
        }
        __syncthreads();
        PCG_STAMP(1 + 4 * iter);
        // ---- publish: thread 12 li + 2 r + h carries half h of q[i0 + li][r] (sum of the four waves' partials, fixed order)
        const unsigned epoch = (unsigned)iter + 1u;
        unsigned long long* gr = g.granules + (size_t)(iter & 1) * (2 * n6);
        if (tid < 12 * (i1 - i0) && !(g.fault_pcg && blockIdx.x == 0 && iter == 0)) {
            const int li = tid / 12, r6 = (tid % 12) >> 1;
            const double* qp = sQ + li * 32;
            const double qv = ((qp[r6] + qp[8 + r6]) + qp[16 + r6]) + qp[24 + r6];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(qv);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)(bits & 0xffffffffull);
            st_granule(gr + 2 * (6 * i0) + tid, ((unsigned long long)epoch << 32) | half);
        }
        PCG_STAMP(2 + 4 * iter);
        // ---- gather every row's q: every pass re-reads ALL granules of the chunk (loads in flight together), until
        //      every tag matches (Guideline 16, sweep_granules) — one L2 round trip per pass, not per element
        constexpr int SW = 4;
        for (int base = 0; base < n6 && !timeout; base += 256 * SW) {
            unsigned long long lo[SW], hi[SW];
            unsigned spins = 0;
            while (true) {
#pragma unroll
                for (int jj = 0; jj < SW; ++jj) {
                    const int t = base + jj * 256 + tid;
                    if (t < n6) { lo[jj] = ld_granule(gr + 2 * t); hi[jj] = ld_granule(gr + 2 * t + 1); }
                }
                bool ok = true;
#pragma unroll
                for (int jj = 0; jj < SW; ++jj) {
                    const int t = base + jj * 256 + tid;
                    if (t < n6) ok = ok && ((unsigned)(lo[jj] >> 32) == epoch) && ((unsigned)(hi[jj] >> 32) == epoch);
                }
                if (__all(ok)) break;                                  // per wave; the workgroup meets at the barrier below
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_limit) { timeout = true; break; }
            }
            if (!timeout) {
#pragma unroll
                for (int jj = 0; jj < SW; ++jj) {
                    const int t = base + jj * 256 + tid;
                    if (t < n6) sq[t] = __longlong_as_double((long long)((hi[jj] << 32) | (lo[jj] & 0xffffffffull)));
                }
            }
        }
        if (timeout) sP[2] = 1.0;
        __syncthreads();
        PCG_STAMP(3 + 4 * iter);
        // ---- vector recurrences on wave 0, blocks in registers (identical in every workgroup: same data, same order)
        if (vec) {
            // q of the owned blocks: registers when one block per lane, re-read from LDS otherwise (register budget)
            double qq[BPL == 1 ? 6 : 1], sv1[BPL == 1 ? 6 : 1];
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < BPL; ++k) {
                const int a = blk0 + OWN_STRIDE * k;
                const bool own = a < g.Npf;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double qv = own ? sq[6 * a + c] : 0.0;
                    if (BPL == 1) qq[c] = qv;
                    part += dd_[k][c] * qv;
                }
            }
            const double dq = owners_sum(part, 1);
            const double alpha = dn / dq;
            part = 0.0;
#pragma unroll
            for (int k = 0; k < BPL; ++k) {
                const int a = blk0 + OWN_STRIDE * k;
                const bool own = a < g.Npf;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double qv = (BPL == 1) ? qq[c] : (own ? sq[6 * a + c] : 0.0);
                    xx_[k][c] += alpha * dd_[k][c];
                    rr_[k][c] -= alpha * qv;
                }
#pragma unroll
                for (int r6 = 0; r6 < 6; ++r6) {
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; ++c) v += (MREG ? mm_[6 * r6 + c] : (own ? Minv[36 * a + 6 * r6 + c] : 0.0)) * rr_[k][c];
                    if (BPL == 1) sv1[r6] = v; else if (own) ss[6 * a + r6] = v;
                    part += rr_[k][r6] * v;
                }
            }
            const double dnn = owners_sum(part, 0);
            const double beta = dnn / dn;
#pragma unroll
            for (int k = 0; k < BPL; ++k) {
                const int a = blk0 + OWN_STRIDE * k;
                const bool own = a < g.Npf;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double sv = (BPL == 1) ? sv1[c] : (own ? ss[6 * a + c] : 0.0);     // own writes, own reads: no fence needed
                    dd_[k][c] = sv + beta * dd_[k][c];
                    if (own) sd[6 * a + c] = dd_[k][c];
                }
            }
            if (tid == 0) sP[0] = dnn;
        }
        __syncthreads();
        PCG_STAMP(4 + 4 * iter);
        if (sP[2] != 0.0) { timeout = true; break; }
        dn = sP[0];
        iter += 1;
    }
    if (timeout) { if (tid == 0) st->pcg_timeout = 1; return; }
    // x is final: the owner of each of this workgroup's block rows stores it and does K8 (oplus); workgroup 0 publishes the statistics.
    if (vec) {
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            const int a = blk0 + OWN_STRIDE * k;
            if (a >= i0 && a < i1 && (WIDE || MULTIROW || lane == (i0 & 63))) {
#pragma unroll
                for (int c = 0; c < 6; ++c) g.x[6 * a + c] = xx_[k][c];
                const int ip = g.free_pose[a];
                const int sel = st->sel;
                pose_oplus(g.pose[sel] + POSE_STRIDE * ip, xx_[k], g.pose[sel ^ 1] + POSE_STRIDE * ip);
            }
        }
        if (tid == 0 && blockIdx.x == 0) {
            st->pcg_residual = 0.5 * dn;
            st->pcg_iter = iter;
            st->pcg_total += iter;
            if (iter > st->pcg_max) st->pcg_max = iter;
        }
    }
}

// ---- K6, reduced systems of <= 64 block rows: ONE WAVEFRONT per block row, everything in registers
// Same algorithm, same hand-off (granules, Guideline 16 form R2) as k_pcg, laid out so that an iteration needs no LDS and no
// barrier at all: lane a of EVERY wave owns block column a — its 6-blocks of r, d, x, its Minv block and the block S(i, a) of the
// wave's own row i (zero when the row has no such block) all live in registers.  q_i = S_i d is 36 multiply-adds per lane and
// one 6-value reduce-scatter over the wave (in-row DPP stages first, so only two values cross rows); the owners of the six sums
// store the twelve granules; lane a then polls exactly the twelve granules of q_a (96 contiguous bytes) — what it needs for
// its share of the vector recurrences, which every wave performs redundantly and bitwise identically (same data, same
// instruction sequence), so all waves take the same branch at every convergence test.  The four-wave k_pcg paid three workgroup
// barriers and six LDS round trips per iteration for the same arithmetic (stamped: 1.3 us of a 2.5 us iteration).
// Upward halving reduce-scatter (masks 1, 2, 4, ... 32): like ReduceScatter, cheap in-row exchanges while the array is long.
template <int N, int M>
struct ReduceScatterUp {
    static __device__ __forceinline__ void run(double* a, int lane, int& off, int& len) {
        constexpr int H = (N + 1) / 2;
        const bool up = (lane & M) != 0;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const double lo = a[j];
            const double hi = (j + H < N) ? a[j + H] : 0.0;
            a[j] = halves_exchange_sum<M>(lo, hi, up);
        }
        if (up) { off += H; len = len > H ? len - H : 0; } else { len = len < H ? len : H; }
        ReduceScatterUp<H, M * 2>::run(a, lane, off, len);
    }
};
template <int N>
struct ReduceScatterUp<N, 64> {
    static __device__ __forceinline__ void run(double*, int, int&, int&) {}
};

// GV (gather variant, measured in profiles/r02_pcg1_gather_variants.log): 0 = twelve 8-byte loads per sweep, one sweep in flight;
// 1 = six 16-byte loads (two granules each: every 8-byte half is one store of its producer); 2 = 1 + the next sweep is issued
// before the previous one is checked (two sweeps in flight: the poll period halves without waiting less).
// 3 = 2 + the XCD-LOCAL hand-off: the launch carries 8x the workgroups and only every 8th one works (blockIdx.x % 8 == window % 8),
// which the dispatcher is OBSERVED to place on one XCD.  Nothing relies on that: iteration 0 runs the cross-XCD protocol and every
// row publishes its HW_REG_XCC_ID with its q; only if ALL rows report the same XCD — a fact about this launch, read from the
// hardware — do the later iterations publish with PLAIN stores (kept in that XCD's L2, the coherence point of its CUs) and keep
// reading with sc1 loads (past the L1, served by that L2): an L2 round trip instead of a fabric one per hand-off.  Otherwise they
// stay on the write-through path.  Both paths move the same bits; the choice is identical in every wave (same gathered words).
// FIN (round 4): the launch also carries the finalisation of the reduced system — k_schur_finalize as a prologue: workgroup (= wave) b of
// n_blk first finalises stored block b (schur_block<PUB>: write-through stores, drain, flag), the first Npf of them then go on as the PCG
// rows; lane a of row i0 waits for the flags of the two blocks it reads — S(i0, a) and the diagonal block of pose a (Minv_a, b_s a) — and
// fetches them with sc1 loads (Guideline 16 R1: every load of handed-off bytes is sc1).  One launch and one kernel boundary less per
// damped solve.  Finalise-only waves never wait, row waves wait for finalise waves (dispatched right behind them) and for each other as
// before: the co-residency requirement is still "the Npf row waves of a window".  The q hand-off tags carry the unit (tag =
// unit << 10 | iteration + 1) because nobody zeroes the granules between the units of one optimise call any more (k_reset does, once).
template <class Src, int GV, bool FIN = false>
__global__ __launch_bounds__(64) void k_pcg1(const Src src) {
    static_assert(!FIN || GV == 1, "the fused finalisation rides on the default gather variant");
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    if (GV == 3 && (blockIdx.x & 7) != (blockIdx.y & 7)) return;
    const int lane = threadIdx.x;
    unsigned unit_tag = 0u;
    if (FIN) {
        const int b = (int)blockIdx.x;
        if (b >= g.n_blk) return;                     // a batched launch is sized for the largest window
        const int4 bd = g.blk_desc[2 * b], be = g.blk_desc[2 * b + 1];
        if (!(st->mode & MODE_TRIAL)) return;
        unit_tag = (unsigned)(st->trials_run[0] + st->trials_run[1]) + 1u;
        const LinSel<Src> lsel(g, st->lin_sel);
        schur_block<true>(g, lsel.get(), st, b, lane, bd, be, 0.0, unit_tag);
    }
    if (!(st->mode & MODE_TRIAL)) return;
    const int i0 = GV == 3 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (i0 >= g.Npf) return;                          // a batched launch is sized for the largest window
    const int Npf = g.Npf, n6 = 6 * Npf;
    const bool own = lane < Npf;
#ifdef VISFS_BA_STAMPS
#define PCG1_STAMP(slot) do { if (lane == 0 && i0 == g.stamp_wg && (slot) < 100) g.stamps[(slot)] = wall_clock64(); } while (0)
    if (lane == 0 && i0 == g.stamp_wg) g.stamps[127] = wall_clock64();
#else
#define PCG1_STAMP(slot) do { } while (0)
#endif
    // ---- set-up: S(i0, lane), Minv_lane, r = b_s; every load below is independent of the others
    const int code = own ? g.pcg1_code[i0 * Npf + lane] : -1;
    const unsigned spin_limit = g.fault_pcg ? (1u << 10) : (1u << 22);
    bool timeout = false;
    double Sr[36], mm[36], rr[6], dd[6], xx[6];
    if (FIN) {
        // wait for the two blocks this lane reads (finalised by other waves of this launch), then fetch them past the L1 (sc1)
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        const int dcode = own ? g.pcg1_code[lane * Npf + lane] : -1;          // the diagonal block of pose `lane`
        const uint32_t* f1 = g.fin_flag + (code >= 0 ? (code >> 1) : 0);
        const uint32_t* f2 = g.fin_flag + (dcode >= 0 ? (dcode >> 1) : 0);
        for (unsigned spins = 0;; ++spins) {
            const unsigned a1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a2 = __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = (code < 0 || a1 == unit_tag) && (dcode < 0 || a2 == unit_tag);
            if (__all(ok)) break;
            if (spins > spin_limit) { timeout = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (timeout) { if (lane == 0) st->pcg_timeout = 1; return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");              // (no instruction: keeps the loads below behind the poll)
        const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)g.S, 0, (int)(g.n_blk * 288), 0x00020000);
        const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void*)g.Minv, 0, (int)(Npf * 288), 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.bs, 0, (int)(Npf * 48), 0x00020000);
        const int oS = 288 * (code >= 0 ? (code >> 1) : 0), oM = 288 * (own ? lane : 0), oB = 48 * (own ? lane : 0);
        v4u_t vs[18], vm[18], vb[3];
#pragma unroll
        for (int q = 0; q < 18; ++q) { vs[q] = __builtin_amdgcn_raw_buffer_load_b128(rS, oS + 16 * q, 0, 16); vm[q] = __builtin_amdgcn_raw_buffer_load_b128(rM, oM + 16 * q, 0, 16); }
#pragma unroll
        for (int q = 0; q < 3; ++q) vb[q] = __builtin_amdgcn_raw_buffer_load_b128(rB, oB + 16 * q, 0, 16);
        double sv[36];
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            sv[2 * q] = code >= 0 ? __hiloint2double((int)vs[q].y, (int)vs[q].x) : 0.0; sv[2 * q + 1] = code >= 0 ? __hiloint2double((int)vs[q].w, (int)vs[q].z) : 0.0;
            mm[2 * q] = own ? __hiloint2double((int)vm[q].y, (int)vm[q].x) : 0.0; mm[2 * q + 1] = own ? __hiloint2double((int)vm[q].w, (int)vm[q].z) : 0.0;
        }
        const bool tr = (code & 1) != 0;              // the stored block is (lane, i0): use its transpose
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) Sr[6 * r + c] = tr ? sv[6 * c + r] : sv[6 * r + c];
#pragma unroll
        for (int q = 0; q < 3; ++q) { rr[2 * q] = own ? __hiloint2double((int)vb[q].y, (int)vb[q].x) : 0.0; rr[2 * q + 1] = own ? __hiloint2double((int)vb[q].w, (int)vb[q].z) : 0.0; }
#pragma unroll
        for (int c = 0; c < 6; ++c) xx[c] = 0.0;
    } else {
        const double2* Sb = reinterpret_cast<const double2*>(g.S + 36 * (size_t)(code >= 0 ? (code >> 1) : 0));
        const double2* Mb = reinterpret_cast<const double2*>(g.Minv + 36 * (size_t)(own ? lane : 0));
        double sv[36];
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const double2 v = (code >= 0) ? Sb[q] : make_double2(0.0, 0.0);
            sv[2 * q] = v.x; sv[2 * q + 1] = v.y;
            const double2 m = own ? Mb[q] : make_double2(0.0, 0.0);
            mm[2 * q] = m.x; mm[2 * q + 1] = m.y;
        }
        const bool tr = (code & 1) != 0;              // the stored block is (lane, i0): use its transpose
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) Sr[6 * r + c] = tr ? sv[6 * c + r] : sv[6 * r + c];
#pragma unroll
        for (int c = 0; c < 6; ++c) { rr[c] = own ? g.bs[6 * (own ? lane : 0) + c] : 0.0; xx[c] = 0.0; }
    }
    double part = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) v += mm[6 * r + c] * rr[c];
        dd[r] = v;
        part += rr[r] * v;
    }
    double dn = wave_sum(part);
    double d0 = 1e-6 * dn;
    {
        // (FIN: block 0's wave copies pcg_residual to pcg_res_in in this very launch — read the source, which nobody writes before every
        // row has left its loop)
        const double res_in = FIN ? st->pcg_residual : st->pcg_res_in;
        if (res_in > 0.0 && res_in > d0) d0 = res_in;
    }
    int iter = 0;
    bool xcd_local = false;                            // GV == 3: every row of this window runs on one XCD (decided after iteration 0)
    PCG1_STAMP(0);
    while (true) {
        if (dn <= d0 || iter >= n6 || !(dn == dn)) break;
        // ---- q_i0 = sum_a S(i0, a) d_a: six partial sums per lane, reduce-scattered over the wave
        double y[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) v += Sr[6 * r + c] * dd[c];
            y[r] = v;
        }
        int off = 0, len = 6;
        ReduceScatterUp<6, 1>::run(y, lane, off, len);
        const unsigned epoch = (FIN ? (unit_tag << 10) : 0u) + (unsigned)iter + 1u;
        unsigned long long* gr = g.granules + (size_t)(iter & 1) * (2 * n6);
        PCG1_STAMP(1 + 4 * iter);
        if (len >= 1 && !(g.fault_pcg && i0 == 0 && iter == 0)) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(y[0]);
            unsigned long long* o = gr + 2 * (6 * i0 + off);
            const unsigned long long g0 = ((unsigned long long)epoch << 32) | (bits & 0xffffffffull), g1 = ((unsigned long long)epoch << 32) | (bits >> 32);
            if (GV == 3 && xcd_local) { o[0] = g0; o[1] = g1; }            // plain stores: the line stays in this XCD's L2
            else { st_granule(o, g0); st_granule(o + 1, g1); }
        }
        if (GV == 3 && iter == 0 && lane == 0)                             // where this row runs (tag 1 = valid)
            st_granule(g.granules + 4 * n6 + i0, (1ull << 32) | (unsigned long long)xcc_id());
        // ---- q_lane from the granules of block row `lane` (published by the wave of that row): sweep until every tag matches
        PCG1_STAMP(2 + 4 * iter);
        double qq[6];
        {
            const unsigned long long* gl = gr + 12 * (size_t)(own ? lane : 0);
            unsigned long long w[12];
            unsigned spins = 0;
            if (GV == 0) {
                while (true) {
                    bool ok = true;
                    if (own) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) w[k] = ld_granule(gl + k);
#pragma unroll
                        for (int k = 0; k < 12; ++k) ok = ok && ((unsigned)(w[k] >> 32) == epoch);
                    }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit) { timeout = true; break; }
                }
            } else {
                // 16-byte write-through-coherent loads (sc1: served past this CU's L1) through a buffer descriptor over the granule array
                typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.granules, 0, (int)(4 * n6 * 8), 0x00020000);
                const int voff = (int)(((iter & 1) * (2 * n6) + 12 * (own ? lane : 0)) * 8);
                auto sweep = [&](v4u_t (&v)[6]) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * k, 0, 16);
                };
                auto tags_ok = [&](const v4u_t (&v)[6]) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 6; ++k) ok = ok && (v[k].y == epoch) && (v[k].w == epoch);
                    return ok || !own;
                };
                v4u_t a[6], b[6];
                sweep(a);
                if (GV >= 2) {
                    // two sweeps in flight: the next one is ISSUED before the previous one is checked (the sched_barrier pins that
                    // order: the check then waits with vmcnt(6), not for the sweep just issued)
                    while (true) {
                        __builtin_amdgcn_s_sleep(4); sweep(b); __builtin_amdgcn_sched_barrier(0);
                        if (__all(tags_ok(a))) break;
                        __builtin_amdgcn_s_sleep(4); sweep(a); __builtin_amdgcn_sched_barrier(0);
                        if (__all(tags_ok(b))) {
#pragma unroll
                            for (int k = 0; k < 6; ++k) a[k] = b[k];
                            break;
                        }
                        if (++spins > spin_limit) { timeout = true; break; }
                    }
                } else {
                    while (true) {
                        if (__all(tags_ok(a))) break;
                        __builtin_amdgcn_s_sleep(1); sweep(a);
                        if (++spins > spin_limit) { timeout = true; break; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    w[2 * k] = ((unsigned long long)a[k].y << 32) | a[k].x;
                    w[2 * k + 1] = ((unsigned long long)a[k].w << 32) | a[k].z;
                }
            }
            if (timeout) break;
            if (GV == 3 && iter == 0) {
                // the placement word of row `lane`: published before that row's q granules left, so usually there already
                const unsigned long long* xp = g.granules + 4 * n6 + (own ? lane : 0);
                unsigned long long xw = 0ull;
                while (true) {
                    if (own) xw = ld_granule(xp);
                    if (__all(!own || (xw >> 32) == 1ull)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit) { timeout = true; break; }
                }
                if (timeout) break;
                const unsigned x0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xw);      // row 0's XCD
                xcd_local = __all(!own || (unsigned)xw == x0);
            }
#ifdef VISFS_BA_STAMPS
            if (lane == 0 && i0 == g.stamp_wg && iter < 26) g.stamps[100 + iter] = spins + (xcd_local ? 1000u : 0u);
#endif
#pragma unroll
            for (int c = 0; c < 6; ++c)
                qq[c] = own ? __longlong_as_double((long long)((w[2 * c + 1] << 32) | (w[2 * c] & 0xffffffffull))) : 0.0;
        }
        PCG1_STAMP(3 + 4 * iter);
        // ---- vector recurrences, identical in every wave
        part = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) part += dd[c] * qq[c];
        const double dq = wave_sum(part);
        const double alpha = dn / dq;
        part = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) { xx[c] += alpha * dd[c]; rr[c] -= alpha * qq[c]; }
        double sv[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) v += mm[6 * r + c] * rr[c];
            sv[r] = v;
            part += rr[r] * v;
        }
        const double dnn = wave_sum(part);
        const double beta = dnn / dn;
#pragma unroll
        for (int c = 0; c < 6; ++c) dd[c] = sv[c] + beta * dd[c];
        dn = dnn;
        PCG1_STAMP(4 + 4 * iter);
        iter += 1;
    }
    if (timeout) { if (lane == 0) st->pcg_timeout = 1; return; }
    // x is final: the lane owning this wave's block row stores it and does K8 (oplus); row 0 publishes the statistics
    if (lane == i0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) g.x[6 * i0 + c] = xx[c];
        const int ip = g.free_pose[i0];
        const int sel = st->sel;
        pose_oplus(g.pose[sel] + POSE_STRIDE * ip, xx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
    }
    if (lane == 0 && i0 == 0) {
        st->pcg_residual = 0.5 * dn;
        st->pcg_iter = iter;
        st->pcg_total += iter;
        if (iter > st->pcg_max) st->pcg_max = iter;
    }
}

// ---- K6, reduced systems that fit ONE compute unit: no cross-workgroup hand-off at all
// A hand-off between workgroups costs ~1.5 us per PCG iteration on this chip whatever the placement (stamped in k_pcg1: the same on
// one XCD through its L2 as across XCDs, profiles/r02_pcg1_gather_variants.log) — half of a k_pcg1 iteration.  When the system has at
// most 56 free poses and block rows of at most 21 blocks, ONE workgroup solves it, a THREAD PER SCALAR ROW (round 4): the thread keeps
// its row of S — up to CU_KR blocks in registers, the rest in LDS slices it alone reads — and its row of Minv, d and r live in LDS, and
// an iteration is one pass over the row, two workgroup sums and four barriers among five or six wavefronts.
// Why a thread a row: one wavefront issues an instruction every 6-8 cycles whatever it is (profiles/r04_fp64_chain_micro.log), so
// a kernel is as long as the longest instruction list on a SIMD.  Round 2's form (three threads a row, sixteen wavefronts) halved the
// multiply-adds per thread but every wavefront paid the whole of the sums and barriers: 2.9 us an iteration of which 1.8 in the two
// sums (profiles/r04_pcg_cu_stamps.log).  Same recurrences, tolerance and residual carry-over as k_pcg ([g2o-upstream]
// LinearSolverPCG::solve); S, b_s and Minv come from k_schur_finalize.  Every sum has a fixed order.
constexpr int CU_K = 21;               // blocks in a block row of S at most
constexpr int CU_KR = 13;              // ... of which in registers (78 values); the others in LDS, [slice][c][thread]: conflict-free 8-byte reads
constexpr int CU_MAX_N6 = 336;         // 56 free poses = six wavefronts
constexpr int CU_MAX_T = 384;
size_t pcg_cu_lds_bytes(const int npf, const int max_row) {
    const int T = (6 * npf + 63) / 64 * 64;
    return (size_t)std::max(0, max_row - CU_KR) * 6 * T * sizeof(double);
}

// a0 b0 + a1 b1 + ... + a5 b5 as ONE stated chain of fused multiply-adds.  Written out because `a0 * b0 + a1 * b1` under
// -ffp-contract=fast may fuse EITHER product (the other is rounded on its own), and the One / Many instantiations of a kernel did
// choose differently: a batch member differed from its single-window solve in the last bit (test_single_workgroup_pcg_in_a_batch_of_
// unequal_windows).  Every bit-identity claim of this file rests on sums whose association AND fusion are spelled out.
__device__ __forceinline__ double dot6(const double a0, const double a1, const double a2, const double a3, const double a4, const double a5,
                                       const double b0, const double b1, const double b2, const double b3, const double b4, const double b5) {
    double v = a0 * b0;
    v = __builtin_fma(a1, b1, v); v = __builtin_fma(a2, b2, v); v = __builtin_fma(a3, b3, v); v = __builtin_fma(a4, b4, v); v = __builtin_fma(a5, b5, v);
    return v;
}
// ... and as three chains of two, joined in a fixed order (the banded Cholesky's block updates: a shorter dependent chain)
__device__ __forceinline__ double dot6_pairs(const double* __restrict__ a, const double* __restrict__ b) {
    const double t0 = __builtin_fma(a[1], b[1], a[0] * b[0]), t1 = __builtin_fma(a[3], b[3], a[2] * b[2]), t2 = __builtin_fma(a[5], b[5], a[4] * b[4]);
    return (t0 + t1) + t2;
}
template <class Src>
__global__ __launch_bounds__(CU_MAX_T) void k_pcg_cu(const Src src) {
    // No implicit fusion in this body: every multiply-add that is meant to be fused is written as one (dot6, __builtin_fma).  Under
    // -ffp-contract=fast the product of `r * d` was fused into the first add of the wave sum in one instantiation and not in the other:
    // a window solved through visfs_ba_solve_window (One) and as a batch of one (Many) differed in the last bit.
#pragma clang fp contract(off)
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n6 = 6 * g.Npf, T = (n6 + 63) & ~63;            // (a batched launch may be wider than this window: its own width lays the slices out)
    const bool act = tid < n6;
    const int row = act ? tid : 0, bi = row / 6, r6 = row - 6 * bi;
    // (the first loads of the set-up go out with the gate's: a gated-off launch reads a few words for nothing)
    const int rb = g.row_ptr[bi], nb = g.row_ptr[bi + 1] - rb;
    double Mv[6];
    {
        const double2* M = reinterpret_cast<const double2*>(g.Minv + 36 * (size_t)bi + 6 * r6);
        const double2 m0 = M[0], m1 = M[1], m2 = M[2];
        Mv[0] = m0.x; Mv[1] = m0.y; Mv[2] = m1.x; Mv[3] = m1.y; Mv[4] = m2.x; Mv[5] = m2.y;
    }
    double rr = g.bs[row];
    const double res_in = st->pcg_res_in;
    if (!(st->mode & MODE_TRIAL)) return;
#ifdef VISFS_BA_STAMPS
    __shared__ unsigned long long cstamp[64];
#define CU_STAMP(slot) do { if (threadIdx.x == 0 && (slot) < 64) cstamp[(slot)] = wall_clock64(); } while (0)
#else
#define CU_STAMP(slot) do { } while (0)
#endif
    CU_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) double cu_slices[];      // [max_row - CU_KR][T][6]: a thread's slice is 48 contiguous bytes — three conflict-free 16-byte reads
    __shared__ __attribute__((aligned(16))) double sd[CU_MAX_N6 + 8], sr[CU_MAX_N6 + 8];
    __shared__ __attribute__((aligned(16))) double sPa[8], sPb[8];
    const int tslot = min(tid, T - 1);
    if (tid < 8) { sPa[tid] = 0.0; sPb[tid] = 0.0; }          // (wavefronts a launch does not have add nothing)
    // ---- set-up: the thread's row of S from S_rows — entry c of the row's k-th block is 8 bytes a lane, 512 contiguous bytes a
    // wavefront (the 16-byte gathers out of the block-major S touched ~30 cache lines an instruction: 6 of the 7 us of the set-up,
    // profiles/r04_pcg_cu_stamps.log); slots the row does not use were zeroed by the upload.  Nothing depends on the row's list but
    // the column offsets: every load of the set-up is in flight at once.
    const int nsl = max(0, g.pcg_max_row - CU_KR);            // LDS slices of this launch (uniform)
    double Sv[CU_KR][6];
    int coff[CU_K];
    {
        const double* Sr = g.S_rows + min(tid, g.cu_T - 1);
        const size_t NT = (size_t)g.cu_T;
#pragma unroll
        for (int k = 0; k < CU_K; ++k) {
            const int nc = k < nb ? k : nb - 1;               // (a free pose's block row holds at least its diagonal block)
            coff[k] = g.row_col[rb + nc];
        }
#pragma unroll
        for (int k = 0; k < CU_K; ++k) {
            if (k < CU_KR) {
                const bool slot = k < g.pcg_max_row;          // (uniform: S_rows ends with the longest row's last slot)
#pragma unroll
                for (int c = 0; c < 6; ++c) { const double v = Sr[(6 * (size_t)(slot ? k : 0) + c) * NT]; Sv[k < CU_KR ? k : 0][c] = slot ? v : 0.0; }
            } else if (k - CU_KR < nsl) {
                double v[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) v[c] = Sr[(6 * (size_t)k + c) * NT];
                if (tid < T) {
                    double2* sl = reinterpret_cast<double2*>(cu_slices + ((size_t)(k - CU_KR) * T + tid) * 6);       // own writes, own reads: no barrier needed
                    sl[0] = make_double2(v[0], v[1]); sl[1] = make_double2(v[2], v[3]); sl[2] = make_double2(v[4], v[5]);
                }
            }
            coff[k] = (act && k < nb) ? 6 * coff[k] : 0;
        }
    }
    if (!act) {
        rr = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) Mv[c] = 0.0;
    }
    double dd = 0.0, xx = 0.0;
    if (act) sr[row] = rr;
    // sum over the workgroup, every thread gets it: wave butterfly, then the eight per-wave partials (zeros for wavefronts that do not
    // exist) read together and added in a fixed order — a loop over the launch's wave count read them one LDS round trip at a time
    auto wg_sum = [&](const double v, double* part) -> double {
        const double ws = wave_sum(v);
        if (lane == 0) part[wave] = ws;
        __syncthreads();
        const double2* p2 = reinterpret_cast<const double2*>(part);
        const double2 a = p2[0], b = p2[1], c = p2[2], d = p2[3];
        return ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
    };
    auto minv_row = [&]() -> double {                 // (Minv r)[row]: r of the row's block from LDS (behind a barrier)
        const double2* rv = reinterpret_cast<const double2*>(sr + 6 * bi);
        const double2 a0 = rv[0], a1 = rv[1], a2 = rv[2];
        return dot6(Mv[0], Mv[1], Mv[2], Mv[3], Mv[4], Mv[5], a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
    };
    __syncthreads();
    dd = minv_row();
    double dn = wg_sum(rr * dd, sPa);
    double d0 = 1e-6 * dn;
    if (res_in > 0.0 && res_in > d0) d0 = res_in;
    if (act) sd[row] = dd;
    __syncthreads();
    CU_STAMP(1);
    int iter = 0;
    while (true) {
        if (dn <= d0 || iter >= n6 || !(dn == dn)) break;
        // ---- q = S d: the whole row on its thread, four partial sums (fixed association)
        double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int k = 0; k < CU_KR; ++k) {
            const double2* dv = reinterpret_cast<const double2*>(sd + coff[k]);
            const double2 a0 = dv[0], a1 = dv[1], a2 = dv[2];
            acc[k & 3] += dot6(Sv[k][0], Sv[k][1], Sv[k][2], Sv[k][3], Sv[k][4], Sv[k][5], a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
        }
#pragma unroll
        for (int k = CU_KR; k < CU_K; ++k) {
            if (k - CU_KR < nsl) {                            // (uniform)
                const double2* dv = reinterpret_cast<const double2*>(sd + coff[k]);
                const double2 a0 = dv[0], a1 = dv[1], a2 = dv[2];
                const double2* sl = reinterpret_cast<const double2*>(cu_slices + ((size_t)(k - CU_KR) * T + tslot) * 6);
                const double2 s0 = sl[0], s1 = sl[1], s2 = sl[2];
                acc[k & 3] += dot6(s0.x, s0.y, s1.x, s1.y, s2.x, s2.y, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
            }
        }
        const double q = act ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : 0.0;
        if (iter < 8) CU_STAMP(2 + 4 * iter);
        const double dq = wg_sum(dd * q, sPb);
        if (iter < 8) CU_STAMP(3 + 4 * iter);
        const double alpha = dn / dq;
        xx = __builtin_fma(alpha, dd, xx); rr = __builtin_fma(-alpha, q, rr);
        if (act) sr[row] = rr;
        __syncthreads();
        const double z = minv_row();
        if (iter < 8) CU_STAMP(4 + 4 * iter);
        const double dnn = wg_sum(rr * z, sPa);
        const double beta = dnn / dn;
        dd = __builtin_fma(beta, dd, z);
        if (act) sd[row] = dd;
        __syncthreads();
        if (iter < 8) CU_STAMP(5 + 4 * iter);
        dn = dnn;
        iter += 1;
    }
    // x is final: K8 (oplus) by the first row of every block; thread 0 publishes the statistics
    if (act) { g.x[row] = xx; sr[row] = xx; }
    __syncthreads();
    if (act && r6 == 0) {
        double dx[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) dx[c] = sr[row + c];
        const int ip = g.free_pose[bi];
        const int sel = st->sel;
        pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
    }
    if (tid == 0) {
        st->pcg_residual = 0.5 * dn;
        st->pcg_iter = iter;
        st->pcg_total += iter;
        if (iter > st->pcg_max) st->pcg_max = iter;
    }
#ifdef VISFS_BA_STAMPS
    CU_STAMP(40);
    if (tid == 0) cstamp[41] = (unsigned long long)iter;
    __syncthreads();
    if (tid < 64) g.stamps[tid] = cstamp[tid];
#endif
}

// ================================================================= K6 (direct): blocked Cholesky of the reduced camera matrix
// Optimizer/Solver 0, 1, 3 (CSparse / Cholmod / Eigen sparse Cholesky in the reference, Optimizer.cpp:76-91) all factor
// S = L L^T and back-substitute; here S is assembled dense (n = 6 Npf padded to a multiple of 32 with an identity tail)
// and factored right-looking with 32-wide panels:
//   k_chol_diag   (one wavefront)  : wave-synchronous Cholesky of the first 32x32 diagonal block and its inverse;
//   k_chol_update (all CUs)        : L21 = A21 L11^-T formed per 64x64 tile into LDS, then A22 -= L21 L21^T on 32x32 tiles per
//                                    wave with fp64 MFMA (v_mfma_f64_16x16x4_f64) — the dense reduced-camera GEMM, the only
//                                    MFMA-shaped work on this path; L goes to chol_f, the updated matrix stays in dense;
//                                    the wave holding the next diagonal block factors it in place (look-ahead);
//   k_chol_solve  (one workgroup)  : blocked forward / backward substitution, then K8 (pose oplus).
// A non-positive or non-finite pivot sets LmState::solver_failed (g2o: solver returns false → the LM trial is rejected).
constexpr int CH_NB = 32;
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_dense_assemble(const DeviceGraph g) {
    const LmState* st = g.st;
    if (!(st->mode & MODE_TRIAL)) return;
    const int n6 = 6 * g.Npf, NP = g.chol_np;
    const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int t = gid; t < g.n_blk * 36; t += stride) {
        const int b = t / 36, q = t % 36, r = q / 6, c = q % 6;
        const int i = g.blk_i[b], j = g.blk_j[b];
        const double v = g.S[t];
        g.dense[(size_t)(6 * i + r) * NP + 6 * j + c] = v;
        // mirror only off-diagonal blocks: on a diagonal block (r,c) and (c,r) would both write each entry, and the two
        // values may differ in the last bit — a write race that made the direct solver non-repeatable
        if (i != j) g.dense[(size_t)(6 * j + c) * NP + 6 * i + r] = v;
    }
    for (int t = n6 + gid; t < NP; t += stride) g.dense[(size_t)t * NP + t] = 1.0;       // identity tail of the padding
}

// One wavefront: Cholesky of the 32x32 diagonal block and its inverse (lanes 0..31 own one row / one column each).
// Factor the 32x32 diagonal block at kb and invert it: ONE wavefront (lanes 0..31 = rows; the upper half idles).  The block
// arrives in sL (LDS, row-major), L11 goes to chol_f, L11^-1 to linv (the ping-pong half of panel kb / 32).
__device__ __forceinline__ void chol_diag_block(const DeviceGraph& g, const int kb, double (*sL)[CH_NB + 1], double* sInv, const int lane) {
    LmState* st = g.st;
    const int NP = g.chol_np;
    const int r = lane & 31;
    const bool act = lane < 32;
    double* linv = g.chol_linv + ((kb / CH_NB) & 1) * CH_NB * CH_NB;
    double a[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) a[c] = sL[r][c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // left-looking: s_r = A[r][c] - sum_{k<c} L[r][k] L[c][k] with the lane's own row in registers and row c read from LDS as
    // broadcasts, the pivot by a readlane, rsq + two Newton steps instead of sqrt + division (the right-looking form with an IEEE
    // sqrt / division pair cost 0.65 us per column).  Two columns per step: column c + 1 needs column c only through
    // L[c+1][c], a readlane from lane c + 1 — one LDS write -> fence -> barrier -> read round trip per PAIR of columns; every sum
    // in the order of the one-column form, so the factor is bit-identical.
    bool bad = false;
#pragma unroll
    for (int c = 0; c < CH_NB; c += 2) {
        double s0 = a[c], s1 = 0.0, t0 = a[c + 1], t1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < c; k += 2) {
            s0 -= a[k] * sL[c][k]; s1 -= a[k + 1] * sL[c][k + 1];
            t0 -= a[k] * sL[c + 1][k]; t1 -= a[k + 1] * sL[c + 1][k + 1];
        }
        const double sv = s0 + s1;
        const double p = readlane_f64(sv, c);
        if (!(p > 0.0) || !(p <= DBL_MAX)) bad = true;
        const double inv = fast_rsqrt(p);
        a[c] = (r >= c) ? sv * inv : 0.0;                              // r == c: p / sqrt(p) = sqrt(p)
        t0 -= a[c] * readlane_f64(a[c], c + 1);                        // the term the one-column form adds last
        const double tv = t0 + t1;
        const double q = readlane_f64(tv, c + 1);
        if (!(q > 0.0) || !(q <= DBL_MAX)) bad = true;
        const double inv1 = fast_rsqrt(q);
        a[c + 1] = (r >= c + 1) ? tv * inv1 : 0.0;
        if (act) { sL[r][c] = a[c]; sL[r][c + 1] = a[c + 1]; }
        if (act && r == c) sInv[c] = inv;
        if (act && r == c + 1) sInv[c + 1] = inv1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (__any(bad)) { if (lane == 0) st->solver_failed = 1; return; }
    if (act) {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) g.chol_f[(size_t)(kb + r) * NP + kb + c] = a[c];      // the factor lives beside the matrix being updated
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // L11^-1: lane c builds column c by forward substitution, X[r][c] = (d_rc - sum_{k<r} L[r][k] X[k][c]) / L[r][r];
    // a[] is reused as the column (no cross-lane traffic: L comes from LDS broadcasts)
#pragma unroll
    for (int rr = 0; rr < CH_NB; ++rr) {
        double v = (rr == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) v -= sL[rr][k] * a[k];
        a[rr] = (rr >= r) ? v * sInv[rr] : 0.0;
    }
    if (act) {
#pragma unroll
        for (int rr = 0; rr < CH_NB; ++rr) linv[rr * CH_NB + r] = a[rr];      // row-major L11^-1
    }
}

// The first panel's diagonal block (the later ones are factored by k_chol_update's look-ahead).
__global__ __launch_bounds__(64) void k_chol_diag(const DeviceGraph g, const int kb) {
    const LmState* st = g.st;
    if (!(st->mode & MODE_TRIAL) || st->solver_failed) return;
    const int NP = g.chol_np, tid = threadIdx.x;
    __shared__ double sL[CH_NB][CH_NB + 1];        // the block, then L11, row-major (padded)
    __shared__ double sInv[CH_NB];                 // 1 / L11[c][c]
    if (tid < 32) {
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) sL[tid][c] = g.dense[(size_t)(kb + tid) * NP + kb + c];
    }
    chol_diag_block(g, kb, sL, sInv, tid);
}

// One launch per panel: L21 = A21 L11^-T and the trailing update A22 -= L21 L21^T together.  A workgroup owns a 64x64 tile
// (ti >= tj) of the trailing matrix; it first forms the L21 rows of its two block rows itself (x[c] = sum_{k<=c} a[k] Linv[c][k],
// one thread per (row, parity of c), L11^-1 in LDS) — redundantly across the tiles of a block row, which costs less than the
// launch it saves — keeps them in LDS as the MFMA operands, and the diagonal tiles store theirs into the factor matrix
// chol_f (NOT in place: other tiles of the same launch still read the unconverted A21 from `dense`).
// One wave = one 32x32 sub-tile (2x2 MFMA tiles of 16x16, K = 32 in 8 steps of 4).
// fp64 MFMA operand maps (cdna_hip_programming.md §3): A[l&15][k = l>>4], B[k = l>>4][l&15], C/D col = l&15, row = (l>>4) + 4*reg.
__global__ __launch_bounds__(256) void k_chol_update(const DeviceGraph g, const int kb) {
    const LmState* st = g.st;
    if (!(st->mode & MODE_TRIAL) || st->solver_failed) return;
    const int NP = g.chol_np, tid = threadIdx.x;
    const int t0 = kb + CH_NB;                          // first row / column of the trailing matrix
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;                                // lower triangle of 64x64 tiles
    __shared__ double sLinv[CH_NB][CH_NB + 1];
    __shared__ double sL[2][64][CH_NB + 1];             // L21 rows of block row ti ([0]) and tj ([1])
    __shared__ double sD[CH_NB][CH_NB + 1];             // look-ahead: the next diagonal block (tile (0,0), wave 0)
    __shared__ double sDinv[CH_NB];
    const double* linv = g.chol_linv + ((kb / CH_NB) & 1) * CH_NB * CH_NB;
    for (int t = tid; t < CH_NB * CH_NB; t += 256) sLinv[t / CH_NB][t % CH_NB] = linv[t];
    __syncthreads();
    {
        const int which = tid >> 7, row_l = (tid >> 1) & 63, par = tid & 1;      // 2 block rows x 64 rows x 2 parities of c
        const int row = t0 + 64 * (which ? tj : ti) + row_l;
        const bool need = (which == 0 || tj != ti) && row < NP;
        double a[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) a[c] = need ? g.dense[(size_t)row * NP + kb + c] : 0.0;
#pragma unroll
        for (int c2 = 0; c2 < CH_NB / 2; ++c2) {
            const int c = 2 * c2 + par;                 // static register indices throughout: k <= 2 c2 for both parities,
            double v = 0.0;                             // the odd column adds its last term
#pragma unroll
            for (int k = 0; k <= 2 * c2; ++k) v += a[k] * sLinv[c][k];
            if (par) v += a[2 * c2 + 1] * sLinv[c][2 * c2 + 1];
            sL[which][row_l][c] = v;
            if (need && which == 0 && tj == ti) g.chol_f[(size_t)row * NP + kb + c] = v;
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int wi = 32 * (wave >> 1), wj = 32 * (wave & 1);
    const int ri = t0 + 64 * ti + wi, rj = t0 + 64 * tj + wj;
    if (ri >= NP || rj >= NP || rj > ri + 31) return;   // outside the matrix / strictly above the diagonal
    const int jb = (tj != ti) ? 1 : 0;
    double* A = g.dense;
    const int lr = lane & 15, lk = lane >> 4;
    v4f64 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[u][v][q] = A[(size_t)(ri + 16 * u + lk + 4 * q) * NP + rj + 16 * v + lr];
#pragma unroll
    for (int s2 = 0; s2 < CH_NB / 4; ++s2) {
        const int k = 4 * s2 + lk;
        const double a0 = -sL[0][wi + lr][k], a1 = -sL[0][wi + 16 + lr][k];
        const double b0 = sL[jb][wj + lr][k], b1 = sL[jb][wj + 16 + lr][k];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                A[(size_t)(ri + 16 * u + lk + 4 * q) * NP + rj + 16 * v + lr] = acc[u][v][q];
    // look-ahead: the wave that owns the next diagonal block (tile (0,0), sub-tile (0,0)) factors and inverts it straight from
    // its accumulators, so the next panel needs no launch of its own.  Its L11^-1 goes to the other ping-pong half of chol_linv
    // (this launch's tiles still read the current one).
    if (ti == 0 && tj == 0 && wave == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int q = 0; q < 4; ++q) sD[16 * u + lk + 4 * q][16 * v + lr] = acc[u][v][q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        chol_diag_block(g, t0, sD, sDinv, lane);
    }
}

// Blocked forward (L y = b) and backward (L^T x = y) substitution, then K8.  One workgroup.
__global__ __launch_bounds__(1024) void k_chol_solve(const DeviceGraph g) {
    LmState* st = g.st;
    if (!(st->mode & MODE_TRIAL) || st->solver_failed) return;
    const int NP = g.chol_np, n6 = 6 * g.Npf, tid = threadIdx.x;
    const double* A = g.chol_f;                          // L: diagonal blocks from k_chol_diag, panels from k_chol_update
    double* y = g.chol_y;                               // [NP] work vector
    __shared__ double sy[CH_NB];
    for (int t = tid; t < NP; t += 1024) y[t] = (t < n6) ? g.bs[t] : 0.0;
    __syncthreads();
    // ---- forward
    for (int kb = 0; kb < NP; kb += CH_NB) {
        if (tid < 64) {
            const int r = tid & 31;
            double acc = y[kb + r];
            const double* Lr = A + (size_t)(kb + r) * NP + kb;
            const double inv = 1.0 / Lr[r];
            double lrow[CH_NB];                                           // the row of L11 ahead of the dependent chain
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) lrow[c] = Lr[c];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                const double yc = readlane_f64(acc * inv, c);             // lane c holds the finished y_c (uniform index: no LDS crossbar)
                acc = (r == c) ? yc : ((r > c) ? acc - lrow[c] * yc : acc);
            }
            if (tid < 32) { sy[r] = acc; y[kb + r] = acc; }
        }
        __syncthreads();
        for (int row = kb + CH_NB + tid; row < NP; row += 1024) {
            const double* Lr = A + (size_t)row * NP + kb;
            double v = y[row];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) v -= Lr[c] * sy[c];
            y[row] = v;
        }
        __syncthreads();
    }
    // ---- backward
    for (int kb = NP - CH_NB; kb >= 0; kb -= CH_NB) {
        if (tid < 64) {
            const int r = tid & 31;
            double acc = y[kb + r];
            const double inv = 1.0 / A[(size_t)(kb + r) * NP + kb + r];
            double lcol[CH_NB];                                           // L^T[r][c] = L[c][r]
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) lcol[c] = A[(size_t)(kb + c) * NP + kb + r];
#pragma unroll
            for (int c = CH_NB - 1; c >= 0; --c) {
                const double xc = readlane_f64(acc * inv, c);
                acc = (r == c) ? xc : ((r < c) ? acc - lcol[c] * xc : acc);
            }
            if (tid < 32) { sy[r] = acc; y[kb + r] = acc; }
        }
        __syncthreads();
        for (int row = tid; row < kb; row += 1024) {
            double v = y[row];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) v -= A[(size_t)(kb + c) * NP + row] * sy[c];
            y[row] = v;
        }
        __syncthreads();
    }
    for (int t = tid; t < n6; t += 1024) g.x[t] = y[t];
    __syncthreads();
    const int sel = st->sel;
    for (int a = tid; a < g.Npf; a += 1024) {
        const int ip = g.free_pose[a];
        double dx[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dx[q] = y[6 * a + q];
        pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
    }
}

// ================================================================= K6 (direct, banded): block-banded factorisation of S in ONE workgroup
// The reference's default linear solver is a SPARSE Cholesky of the block-sparse reduced camera matrix (CSparse,
// Parameters.h:185, Optimizer.cpp:76-91).  In a sliding window a landmark is seen from a run of consecutive key-frames, so S is
// block-BANDED: block (i, j) exists only for |i - j| <= B, B = longest track - 1 (C2 / C4: B = 9 of 49 / 199 block rows), and a
// triangular factor keeps that band.  This kernel factors the band block column by block column inside one workgroup — no dense
// matrix, no launch per panel — and solves in the same launch.  The factorisation is the plain BLOCK CHOLESKY S = W W^T (6x6 blocks; the
// diagonal block of W is the Cholesky factor C_k of the pivot block, the blocks below are W_ik = G_ik C_k^-T by a triangular solve per
// row: nothing is ever inverted, so the factor is as accurate as a scalar Cholesky — see band_chol6 for the history).
//   ring  [rows][B + 1][36]  LDS: block (I, I - d) of the lower band at [(I mod rows)][d]; rows == Npf when the whole band fits
//                            (C2: 136 KB), else a sliding window of rows >= B + 3 block rows and the factor streams to band_L (HBM);
//   step k (pivot factor C_k published, column k final and unscaled):
//     half 1  W_ik = G_ik C_k^-T in place, a lane per row (wave 0: block rows k + 1, k + 2, which it then takes out of the three blocks
//             the next pivot and the next step's first rows depend on; helper waves: the rows below);  wave 3: forward substitution of
//             column k - 1;
//     half 2  wave 0: D_{k+1} = C C^T (every lane redundantly, operands by LDS broadcast: no cross-lane traffic; positive definiteness
//             = positive pivots) WHILE the helpers apply A_ij -= W_ik W_jk^T for k + 3 <= i, k < j <= i (twelve lanes a block);
//   then the backward substitution on one wavefront (right-looking: x_k = C_k^-T z_k, z_j -= W_kj^T x_k) and K8 (pose oplus).
// Every sum has a fixed order: results are bitwise reproducible.  A non-positive or non-finite pivot sets LmState::solver_failed: g2o's
// solver returns false and the LM trial is rejected.
constexpr int BAND_T = 256;
constexpr int BAND_MAX_W = 22;                                  // (B + 1) <= 22: a (B + 2)-row ring of 6x6 blocks fits the LDS budget
size_t band_lds_bytes(const int npf, const int B, const int rows) {
    const size_t W = (size_t)B + 1;
    // ring + right-hand side + flags, then the table of stored block ids and the (m, j) table of the trailing update
    return ((size_t)rows * W * 36 + 6 * (size_t)npf + 8) * sizeof(double) + ((size_t)npf * W * 4 + 15) / 16 * 16 + (((size_t)(B + 1) * (B + 2)) + 15) / 16 * 16;
}
// The plan for a reduced system of npf block rows with block half-bandwidth B: how many block rows stay in LDS (all of them when the
// band fits), or false when even a (B + 2)-row window does not fit (wide bands: the dense blocked Cholesky takes those).
bool band_plan(const int npf, const int B, int* rows, int* lds_bytes) {
    if (npf < 1 || B < 0 || B + 1 > BAND_MAX_W) return false;
    int r = npf;
    if (band_lds_bytes(npf, B, r) > (size_t)BAND_LDS_BUDGET) {
        const size_t fixed = band_lds_bytes(npf, B, 0);
        if (fixed >= (size_t)BAND_LDS_BUDGET) return false;
        r = (int)(((size_t)BAND_LDS_BUDGET - fixed) / ((size_t)(B + 1) * 36 * sizeof(double)));
        while (r > 0 && band_lds_bytes(npf, B, r) > (size_t)BAND_LDS_BUDGET) --r;
        if (r < B + 3) return false;                            // the row that enters replaces block row k - 1 while row k + B is live
    }
    *rows = r; *lds_bytes = (int)band_lds_bytes(npf, B, r);
    return true;
}

// Round 4: Cholesky of the SPD 6x6 pivot block D = C C^T.  The factor comes back PACKED
// (tri6: row i holds C_i0 .. C_ii) with the RECIPROCAL 1 / C_jj on the diagonal: the triangular solves below multiply by it.
// History: round 3 applied D^-1 through a closed-form inverse (two 3x3 adjugates) and lost the whole solution at cond(S) = 8e10 where
// the scalar Cholesky of the checker keeps five digits (profiles/r03_stage_precision.log: 7.3e-1 vs 1.8e-5) — a pose that sees few
// landmarks (or a window without a fixed pose, Estimator.cpp:252) has a nearly singular pivot block; round 4 first applied it as
// X^T (X g), X = C^-1 (backward stable: profiles/r04_band_pivot_study.log), then dropped X altogether: computing it doubled the pivot's
// dependent chain (~250 fp64 instructions on one wavefront per column), the chain IS the kernel's time, and the plain block Cholesky
// needs only C.  Every lane computes the whole factor redundantly (operands arrive as LDS broadcasts: no cross-lane traffic); six
// dependent pivots, each v_rsq_f64 + two Newton steps.  Returns false when a pivot is not positive (what a failed Cholesky is) or not finite.
__device__ __forceinline__ constexpr int tri6(const int i, const int j) { return i * (i + 1) / 2 + j; }       // j <= i
// 1 / sqrt(x), x > 0, on the pivot chain: v_rsq_f64 seed (relative error < 2^-23) and ONE third-order step, y0 (1 + e / 2 + 3 e^2 / 8) with
// e = 1 - x y0^2 (truncation 5 e^3 / 16 < 2^-68) — four dependent operations where fast_rsqrt's two Newton steps are six.
__device__ __forceinline__ double band_rsqrt(const double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y0), y0, 1.0);
    return __builtin_fma(y0 * e, __builtin_fma(0.375, e, 0.5), y0);
}
// `d` = the 21 values of the lower triangle, already in registers (row by row); `done(j)` is called as soon as column j of the factor is
// final — the caller's stores of it then run behind the remaining pivots instead of after the last one (stamped: eleven 16-byte stores
// of one lane after the chain cost 160-200 ns of a 560 ns half step).
template <class Done>
__device__ __forceinline__ bool band_chol6(double a[21], Done&& done) {
    double plast = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double p = a[tri6(j, j)];
        plast = p;
        const double r = band_rsqrt(p);
        a[tri6(j, j)] = r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) a[tri6(i, j)] *= r;                      // column j of C
        done(j);
#pragma unroll
        for (int i = j + 1; i < 6; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) a[tri6(i, k)] -= a[tri6(i, j)] * a[tri6(k, j)];
    }
    // One test instead of six: a pivot that is negative, zero, infinite or NaN makes its reciprocal root NaN or infinite, the whole column
    // below it (0 * inf and 0 * NaN included) NaN or infinite, and through a_5j^2 the LAST pivot NaN or -inf: the last pivot tells.
    return (plast > 0.0) && (plast <= DBL_MAX);
}
// w C^T = g  (a row of W_ik = G_ik C_k^-T; also y = C^-1 c): w_j = (g_j - sum_{m < j} w_m C_jm) / C_jj
__device__ __forceinline__ void band_trsv_fwd(const double c[21], const double g[6], double w[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v = g[j];
#pragma unroll
        for (int m = 0; m < j; ++m) v -= w[m] * c[tri6(j, m)];
        w[j] = v * c[tri6(j, j)];
    }
}
// x = C^-T z: x_j = (z_j - sum_{m > j} C_mj x_m) / C_jj
__device__ __forceinline__ void band_trsv_bwd(const double c[21], const double z[6], double x[6]) {
#pragma unroll
    for (int j = 5; j >= 0; --j) {
        double v = z[j];
#pragma unroll
        for (int m = 5; m > j; --m) v -= c[tri6(m, j)] * x[m];
        x[j] = v * c[tri6(j, j)];
    }
}

template <class Src>
__global__ __launch_bounds__(BAND_T) void k_band_chol(const Src src) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    if (!(st->mode & MODE_TRIAL)) return;
    extern __shared__ __attribute__((aligned(16))) double band_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef VISFS_BA_STAMPS
    // (stamps go to LDS and leave for HBM at the end: a global store per stamp would make every barrier wait for it)
    __shared__ unsigned long long sstamp[128];
#define BAND_STAMP(slot) do { if (tid == 0 && (slot) < 126) sstamp[(slot)] = wall_clock64(); } while (0)
#else
#define BAND_STAMP(slot) do { } while (0)
#endif
    BAND_STAMP(0);
#ifdef VISFS_BA_STAMPS
    if (tid == 0) sstamp[120] = __builtin_readcyclecounter();      // shader clock: with the real-time stamps, the clock the kernel actually ran at
#endif
    // Barrier of the factor loop: LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the wave — the
    // streaming form keeps loads of the entering block row and stores of the factor in flight across steps on purpose.
#define BAND_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    const int Npf = g.Npf, B = g.band_B, W = B + 1, RR = g.band_rows;
    const bool resident = RR >= Npf;
    const int rowsz = W * 36;
    double* ring = band_lds;                                   // [RR][W][36]
    double* cvec = ring + (size_t)RR * rowsz;                  // [6 Npf] right-hand side -> y (W y = b) -> x
    int* sflag = reinterpret_cast<int*>(cvec + 6 * Npf);       // (8 doubles reserved)
    int* scode = reinterpret_cast<int*>(cvec + 6 * Npf + 8);   // [Npf][W] stored block ids (DeviceGraph::band_code)
    unsigned char* pij = reinterpret_cast<unsigned char*>(scode) + ((size_t)Npf * W * 4 + 15) / 16 * 16;   // [B (B + 1) / 2 - 3][2] (m, j), 3 <= m <= B, 1 <= j <= m, by rows
    // ---- the first RR block rows of S: the lower block (I, I - d) is the transpose of the stored upper block (I - d, I).  The block
    // ids go to LDS first, so that the element loads below are independent of each other and many are in flight per thread.
    // (measured and dropped: letting the rows of a RESIDENT band enter progressively through the streaming path — the 7 us up-front
    // load shrinks to 3.6, but waves 1..3 then wait for global loads in 40 of the 49 steps: 87.7 -> 94.7 us per C2 solve)
    const int nrows0 = min(RR, Npf), nslots0 = nrows0 * W;
    for (int t = tid; t < Npf * W; t += BAND_T) scode[t] = g.band_code[t];
    for (int t = tid; t < 6 * Npf; t += BAND_T) cvec[t] = g.bs[t];
    if (tid == 0) sflag[0] = 0;
    for (int p = tid; p < B * (B + 1) / 2 - 3; p += BAND_T) {
        int m = 3; while (m * (m + 1) / 2 - 3 <= p) ++m;       // rows 3 .. m - 1 hold (m - 1) m / 2 - 3 entries
        pij[2 * p] = (unsigned char)m; pij[2 * p + 1] = (unsigned char)(p - ((m - 1) * m / 2 - 3) + 1);
    }
    __syncthreads();
    {
        const int total = nslots0 * 18;                        // double2 elements: (q, q + 1) of a block never straddle two blocks
        // every load unconditional (element and block id clamped, the value masked afterwards): the block ids of a pass are read from LDS
        // together, then all its global loads are in flight at once — a resident C2 band is ONE pass of 35 loads per thread (predicated
        // loads behind their own LDS read of the block id were twelve serial round trips a pass, three passes: 6.8 us of the kernel)
        constexpr int U = 36;
        for (int t0 = tid; t0 < total; t0 += BAND_T * U) {
            double2 v[U];
            int bid[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int t = min(t0 + BAND_T * u, total - 1); bid[u] = scode[t / 18]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(t0 + BAND_T * u, total - 1), h = t - 18 * (t / 18);
                v[u] = reinterpret_cast<const double2*>(g.S + 36 * (size_t)max(bid[u], 0))[h];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + BAND_T * u;
                if (t < total) {
                    const int sl = t / 18, h = t - 18 * sl, q = 2 * h, r0 = q / 6, c0 = q - 6 * r0;     // stored entry (r0, c0) and (r0, c0 + 1)
                    double* dst = ring + 36 * (size_t)sl;
                    const bool has = bid[u] >= 0;
                    dst[6 * c0 + r0] = has ? v[u].x : 0.0; dst[6 * (c0 + 1) + r0] = has ? v[u].y : 0.0;      // transposed
                }
            }
        }
    }
    __syncthreads();
    BAND_STAMP(1);
    // ring row of block row k + i (0 <= i <= B) when block row k sits in ring row kk
    auto ring_row = [&](const int kk, const int i) -> double* { int r = kk + i; r -= (r >= RR) ? RR : 0; return ring + (size_t)r * rowsz; };
    // the packed factor of a pivot block (21 values at the head of its diagonal slot), as LDS broadcasts
    auto load_factor = [&](const double* __restrict__ slot, double c[21]) {
#pragma unroll
        for (int q = 0; q < 20; q += 2) { const double2 v = reinterpret_cast<const double2*>(slot)[q >> 1]; c[q] = v.x; c[q + 1] = v.y; }
        c[20] = slot[20];
    };
    // forward-substitution step s on one wavefront (block row s in ring row sr): y_s = C_s^-1 c_s (every lane), c_i -= W_is y_s for the blocks below
    auto fwd_step = [&](const int s_, const int sr) {
        double cf[21], cs[6], y[6];
        load_factor(ring + (size_t)sr * rowsz, cf);
#pragma unroll
        for (int c = 0; c < 6; ++c) cs[c] = cvec[6 * s_ + c];
        const int nb = min(B, Npf - 1 - s_);
        // the rows of W this lane multiplies (two at most: 6 B <= 126) are loaded before the triangular solve, not after it
        double Lr[2][6], acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;
            if (e < 6 * nb) {
                const int m = e / 6 + 1, rr = e - 6 * (m - 1);
                const double* L = ring_row(sr, m) + 36 * m + 6 * rr;
#pragma unroll
                for (int c = 0; c < 6; c += 2) { const double2 v = *reinterpret_cast<const double2*>(L + c); Lr[u][c] = v.x; Lr[u][c + 1] = v.y; }
                acc[u] = cvec[6 * (s_ + m) + rr];
            }
        }
        band_trsv_fwd(cf, cs, y);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;
            if (e < 6 * nb) {
                double a = acc[u];
#pragma unroll
                for (int c = 0; c < 6; ++c) a -= Lr[u][c] * y[c];
                cvec[6 * s_ + 6 + e] = a;                      // (6 (s + m) + rr = 6 s + 6 + e)
            }
        }
        if (lane < 6) {
            double yl = y[0];
#pragma unroll
            for (int c = 1; c < 6; ++c) yl = lane == c ? y[c] : yl;
            cvec[6 * s_ + lane] = yl;
        }
    };
    // Row rr of W_{k+m,k} = G_{k+m,k} C_k^-T, in place (two lanes a row, each solves the whole row and writes three columns); in the
    // streaming form the row also goes to HBM
    auto trsm_row = [&](const int k, const int kk, const int m, const int rr, const int hf, const double cf[21]) {
        double* row = ring_row(kk, m) + 36 * m + 6 * rr;
        double a[6], w[6];
#pragma unroll
        for (int c = 0; c < 6; c += 2) { const double2 v = *reinterpret_cast<const double2*>(row + c); a[c] = v.x; a[c + 1] = v.y; }
        band_trsv_fwd(cf, a, w);
#pragma unroll
        for (int c = 0; c < 3; ++c) row[3 * hf + c] = hf ? w[3 + c] : w[c];
        if (!resident) {
            double* h = g.band_L + (size_t)(k + m) * rowsz + 36 * m + 6 * rr + 3 * hf;
#pragma unroll
            for (int c = 0; c < 3; ++c) h[c] = hf ? w[3 + c] : w[c];
        }
    };
    // A_{k+m,k+j} -= W_{k+m,k} W_{k+j,k}^T: row rr, columns 3 hf .. of the block (twelve lanes a block)
    struct UpdOperands { double wm[6], wj[18], cv[3]; };
    auto update_load = [&](const int kk, const int m, const int j, const int rr, const int hf, UpdOperands& o) {
        const double* Wm = ring_row(kk, m) + 36 * m + 6 * rr;
        const double* Wj = ring_row(kk, j) + 36 * j + 18 * hf;             // rows 3 hf .. of W_{k+j,k}
        const double* C = ring_row(kk, m) + 36 * (m - j) + 6 * rr + 3 * hf;
#pragma unroll
        for (int c = 0; c < 6; c += 2) { const double2 v = *reinterpret_cast<const double2*>(Wm + c); o.wm[c] = v.x; o.wm[c + 1] = v.y; }
#pragma unroll
        for (int q = 0; q < 18; q += 2) { const double2 v = *reinterpret_cast<const double2*>(Wj + q); o.wj[q] = v.x; o.wj[q + 1] = v.y; }
#pragma unroll
        for (int c = 0; c < 3; ++c) o.cv[c] = C[c];
    };
    auto update_store = [&](const int kk, const int m, const int j, const int rr, const int hf, const UpdOperands& o) {
        double* C = ring_row(kk, m) + 36 * (m - j) + 6 * rr + 3 * hf;
        const double* l = o.wm; const double* gj = o.wj;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C[c] = o.cv[c] - dot6_pairs(l, gj + 6 * c);
    };
    // wave 0: the lower triangle of the pivot block at ring row kr into registers (issued before the barrier that ends half 1: the block's
    // last update was this wave's own), then D_k = C_k C_k^T (every lane redundantly); the packed factor replaces the head of the block,
    // column by column while the later pivots are still being computed
    auto read_pivot = [&](const int kr, double a[21]) {
        const double* slot = ring + (size_t)kr * rowsz;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) a[tri6(i, j)] = slot[6 * i + j];
    };
    auto factor_pivot = [&](const int k, const int kr, double a[21]) {
        double* slot = ring + (size_t)kr * rowsz;
        // (every lane stores the same values to the same addresses: no branch splits the chain into blocks the scheduler cannot mix)
        const bool ok = band_chol6(a, [&](const int j) {
#pragma unroll
            for (int i = j; i < 6; ++i) slot[tri6(i, j)] = a[tri6(i, j)];
        });
        if (lane == 0) {
            if (!ok) sflag[0] = 1;
            if (!resident) {
                double* h = g.band_L + (size_t)k * rowsz;
#pragma unroll
                for (int q = 0; q < 20; q += 2) reinterpret_cast<double2*>(h)[q >> 1] = make_double2(a[q], a[q + 1]);
                h[20] = a[20];
            }
        }
    };
    // ---- the factor loop.  At the top of step k the factor C_k of the pivot is published and every block of column k is final, still
    // unscaled (G).  The chain that bounds the kernel is wave 0's: two row solves, three block updates, one 6x6 Cholesky per column.
    int kk = 0;                                                // ring row of block row k
    constexpr int NPEND = (BAND_MAX_W * 36 + (BAND_T - 64) - 1) / (BAND_T - 64);
    double pend[NPEND];                                        // streaming form: the block row of S on its way into the ring (waves 1..3)
    int pend_row = -1;
    int enter_row = nrows0 % RR;                               // ring row of the next block row of S to enter (streaming: the row of block row k - 1)
    // a helper thread's first tile of the trailing update never changes (the table lists blocks by rows): out of the loop
    int tile_m = 3, tile_j = 1;
    if (wave != 0 && ((tid - 64) >> 2) < B * (B + 1) / 2 - 3) { tile_m = pij[2 * ((tid - 64) >> 2)]; tile_j = pij[2 * ((tid - 64) >> 2) + 1]; }
    double cf[21];                                             // C_k: wave 0 keeps what it has just computed, the helpers read it back
    if (wave == 0) { read_pivot(0, cf); factor_pivot(0, 0, cf); }
    BAND_SYNC();
    // (a failed pivot does not leave the loop: its NaNs touch no address, the flag is read once after the loop — a read of it per step was
    // an LDS round trip on the chain)
    for (int k = 0; k < Npf; ++k) {
        const int nb = min(B, Npf - 1 - k);
        if (wave != 0) load_factor(ring + (size_t)kk * rowsz, cf);
        // ---- half 1
        if (wave == 0) {
            // groups of twelve lanes: rows: 0 -> block row k + 1, 1 -> block row k + 2;  updates: 0 -> (1,1), 1 -> (2,1), 2 -> (2,2)
            const int grp = lane / 12, u = lane - 12 * grp, rr = u >> 1, hf = u & 1;
            if (grp < 2 && grp + 1 <= nb) trsm_row(k, kk, grp + 1, rr, hf, cf);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int m = grp == 0 ? 1 : 2, j = grp == 2 ? 2 : 1;
            if (grp < 3 && m <= nb) { UpdOperands o; update_load(kk, m, j, rr, hf, o); update_store(kk, m, j, rr, hf, o); }
            if (k + 1 < Npf) {
                // D_{k+1} is final with this wave's own stores: its loads go out before the barrier (cf is dead until the factor replaces it)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                int k1 = kk + 1; if (k1 == RR) k1 = 0;
                read_pivot(k1, cf);
            }
#ifdef VISFS_BA_STAMPS
            if (lane == 0 && k < 16) sstamp[6 + 6 * k] = wall_clock64();
#endif
        } else {
            if (k > 0) {
                // streaming form: the block row of S that entered the registers one step ago goes to its ring row, the loads of the next one are issued
                if (pend_row >= 0) {
                    double* dst = ring + (size_t)pend_row * rowsz;
#pragma unroll
                    for (int u = 0; u < NPEND; ++u) { const int t = tid - 64 + (BAND_T - 64) * u; if (t < rowsz) dst[t] = pend[u]; }
                    pend_row = -1;
                }
                const int Inew = k - 1 + nrows0;
                if (Inew < Npf) {
                    const int* code = scode + Inew * W;             // (from LDS: a block id fetched from HBM first would be a second dependent global round trip)
#pragma unroll
                    for (int u = 0; u < NPEND; ++u) {
                        const int t = tid - 64 + (BAND_T - 64) * u;
                        pend[u] = 0.0;
                        if (t < rowsz) { const int d = t / 36, q = t - 36 * d, b = code[d]; if (b >= 0) pend[u] = g.S[36 * (size_t)b + 6 * (q % 6) + q / 6]; }
                    }
                    pend_row = enter_row;
                    if (++enter_row == RR) enter_row = 0;
                }
            }
            for (int t = tid - 64; t < 12 * (nb - 2); t += BAND_T - 64) trsm_row(k, kk, 3 + t / 12, (t % 12) >> 1, t & 1, cf);
            if (wave == 3 && k > 0) { int kp = kk - 1; if (kp < 0) kp = RR - 1; fwd_step(k - 1, kp); }
#ifdef VISFS_BA_STAMPS
            if (tid == 192 && k < 16) sstamp[7 + 6 * k] = wall_clock64();
#endif
        }
        BAND_SYNC();
#ifdef VISFS_BA_STAMPS
        if (tid == 0 && k < 16) sstamp[2 + 6 * k] = wall_clock64();
#endif
        // ---- half 2
        if (wave == 0) {
            if (k + 1 < Npf) { int k1 = kk + 1; if (k1 == RR) k1 = 0; factor_pivot(k + 1, k1, cf); }
#ifdef VISFS_BA_STAMPS
            if (lane == 0 && k < 16) sstamp[3 + 6 * k] = wall_clock64();
#endif
        } else {
            // the blocks (k + m, k + j), 3 <= m <= nb, 1 <= j <= m (the table lists them by rows: a prefix for short columns), a 3x3 tile
            // a thread: 36 operands for 54 multiply-adds — the twelve-lane form of wave 0 reads 27 for 18, and with three helper waves
            // at it the LDS pipe, not the arithmetic, set the length of this half (stamped: 920 ns -> see profiles/r04_band_chain.log)
            const int ntask = nb >= 3 ? nb * (nb + 1) / 2 - 3 : 0;
            for (int t = tid - 64; t < 4 * ntask; t += BAND_T - 64) {
                const int q = t >> 2, ta = (t >> 1) & 1, tb = t & 1;
                const bool first = t < BAND_T - 64;
                const int m = first ? tile_m : (int)pij[2 * q], j = first ? tile_j : (int)pij[2 * q + 1];
                const double* Wm = ring_row(kk, m) + 36 * m + 18 * ta;       // rows 3 ta .. of W_{k+m,k}
                const double* Wj = ring_row(kk, j) + 36 * j + 18 * tb;       // rows 3 tb .. of W_{k+j,k}
                double* C = ring_row(kk, m) + 36 * (m - j) + 18 * ta + 3 * tb;
                double wm[18], wj[18], cv[9];
#pragma unroll
                for (int e = 0; e < 18; e += 2) { const double2 v = *reinterpret_cast<const double2*>(Wm + e); wm[e] = v.x; wm[e + 1] = v.y; }
#pragma unroll
                for (int e = 0; e < 18; e += 2) { const double2 v = *reinterpret_cast<const double2*>(Wj + e); wj[e] = v.x; wj[e + 1] = v.y; }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) cv[3 * r + c] = C[6 * r + c];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double* l = wm + 6 * r; const double* gj = wj + 6 * c;
                        C[6 * r + c] = cv[3 * r + c] - dot6_pairs(l, gj);
                    }
            }
#ifdef VISFS_BA_STAMPS
            if (tid == 64 && k < 16) sstamp[5 + 6 * k] = wall_clock64();
#endif
        }
        BAND_SYNC();
        if (k < 16) BAND_STAMP(4 + 6 * k);
        if (++kk == RR) kk = 0;
    }
    if (sflag[0]) { if (tid == 0) st->solver_failed = 1; return; }
    BAND_STAMP(110);
    // ---- backward substitution, one wavefront (the other waves only help to bring chunks of the factor back from HBM)
    if (wave == 0) fwd_step(Npf - 1, (Npf - 1) % RR);           // the last forward step: no blocks below
    __syncthreads();
    const bool fast_bwd = 6 * W <= 64;                         // a lane per accumulator of the window (B <= 9: every BASELINE window)
    double zreg = 0.0;
    for (int k1 = Npf; k1 > 0;) {
        const int k0 = resident ? 0 : max(0, k1 - RR);
        if (!resident) {
            // rows [k0, k1) of the factor: contiguous in band_L, RR rows at most -> distinct ring rows
            // (16-byte loads, twelve in flight per thread: one workgroup pulls ~50 KB per memory round trip instead of 16)
            const int h2 = rowsz / 2, cnt = (k1 - k0) * h2;                 // rowsz = 36 (B + 1) is even
            const double2* srcp = reinterpret_cast<const double2*>(g.band_L + (size_t)k0 * rowsz);
            constexpr int U = 12;
            for (int t0 = tid; t0 < cnt; t0 += BAND_T * U) {
                double2 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int t = t0 + BAND_T * u; v[u] = t < cnt ? srcp[t] : make_double2(0.0, 0.0); }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + BAND_T * u;
                    if (t < cnt) { const int r = t / h2, c = t - h2 * r; reinterpret_cast<double2*>(ring + (size_t)((k0 + r) % RR) * rowsz)[c] = v[u]; }
                }
            }
            __syncthreads();
        }
        if (fast_bwd) {
            // ---- rows [k0, k1) of the factor, W~ = [.. W_kj .. C_k], become X_k W~ (X_k = C_k^-1: unit diagonal blocks): the backward
            // recurrence u_j = y_j - sum_{k > j} (X_k W_kj)^T u_k then has NO triangular solve in its chain, and x_k = X_k^T u_k falls out in
            // parallel afterwards.  All threads: X_k over the packed C_k (a thread a row), then a thread a column of a block.
            for (int k = k0 + tid; k < k1; k += BAND_T) {
                double* slot = ring + (size_t)(k % RR) * rowsz;
                double c[21], x[21];
#pragma unroll
                for (int q = 0; q < 21; ++q) c[q] = slot[q];
                // X_jj = 1 / C_jj (stored), X_ij = -X_ii sum_{q = j}^{i - 1} C_iq X_qj
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    x[tri6(j, j)] = c[tri6(j, j)];
#pragma unroll
                    for (int i = j + 1; i < 6; ++i) {
                        double acc = 0.0;
#pragma unroll
                        for (int q = j; q < i; ++q) acc += c[tri6(i, q)] * x[tri6(q, j)];
                        x[tri6(i, j)] = -c[tri6(i, i)] * acc;
                    }
                }
#pragma unroll
                for (int q = 0; q < 21; ++q) slot[q] = x[q];
            }
            __syncthreads();
            if (k1 == Npf) BAND_STAMP(113);
            // (a thread a half block: X_k is read once for three columns — a thread a column re-read it 2600 times a C2 solve and the
            // pass was LDS-bound at 3.8 us)
            for (int t = tid; t < (k1 - k0) * B * 2; t += BAND_T) {
                const int r = t / (2 * B), e = t - 2 * B * r, m = (e >> 1) + 1, c0 = 3 * (e & 1), k = k0 + r;
                if (m <= k) {
                    double* row = ring + (size_t)(k % RR) * rowsz;
                    double x[21], w[3][6];
                    load_factor(row, x);
                    double* blk = row + 36 * m + c0;
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) w[c][i] = blk[6 * i + c];
#pragma unroll
                    for (int i = 5; i >= 0; --i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            double acc = x[tri6(i, 0)] * w[c][0];
#pragma unroll
                            for (int q = 1; q <= i; ++q) acc += x[tri6(i, q)] * w[c][q];
                            blk[6 * i + c] = acc;
                        }
                }
            }
            __syncthreads();
            if (k1 == Npf) BAND_STAMP(114);
        }
        if (wave == 0 && fast_bwd && resident && W == 10) {
            // The chain below for the band every BASELINE window has (ten blocks per row), whole factor resident: one PERIOD of the slot
            // rotation unrolled, so that the lane a step reads, the lanes that retire and the pointer steps are constants — of the ~55
            // instructions of a generic step 33 were such bookkeeping (215 ns a step; one wavefront: an instruction per 6-8 cycles).
            // Block j sits in slot (j + shift) mod 10 with the last block row in slot 9.
            const int cc = lane % 6, slot = (lane / 6) % 10;      // (lanes 60 .. 63 shadow slot 0: same loads, same sums, the same stores)
            {
                const int jm = Npf - 1 - (9 - slot);                // slot s holds block Npf - 10 + s at the start
                zreg = jm >= 0 ? cvec[6 * jm + cc] : 0.0;
            }
            int k = Npf - 1;
            // this lane's column of block (k, k - m), m = (9 - slot) at the first step (m == 0: the row's own block — values unused)
            const double* src = ring + (size_t)k * rowsz + 36 * ((9 - slot + 10) % 10) + cc;
            const double* zsrc = cvec + cc;
            double Lb[2][6], zb[2];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) Lb[0][rr] = src[6 * rr];
            zb[0] = zsrc[6 * max(k - 10, 0)];
            bool more = true;
            while (more) {
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int ks = 9 - i;                           // (a constant once unrolled) the slot of block k
                    const int cur = i & 1, nxt = cur ^ 1;
                    // the next step's operands, off the chain: m steps down with k, the retiring slot restarts at m = 9
                    const double* srcn = src - (rowsz + 36) + (slot == ks ? rowsz : 0);
                    {
                        const double* sp = k > 0 ? srcn : src;      // (row -1 does not exist: re-read this one, unused)
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) Lb[nxt][rr] = sp[6 * rr];
                        zb[nxt] = zsrc[6 * max(k - 11, 0)];
                    }
                    double uk[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) uk[c] = readlane_f64(zreg, 6 * ks + c);
                    double zn = zreg;
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) zn -= Lb[cur][rr] * uk[rr];
                    if (slot == ks) cvec[6 * k + cc] = zreg;       // u_k
                    zreg = slot == ks ? zb[cur] : zn;
                    src = srcn;
                    if (--k < 0) { more = false; break; }
                }
            }
        } else if (wave == 0 && fast_bwd) {
            // Block j's six accumulators live in lanes 6 (j mod (B + 1)) + cc for as long as rows still reach it (rows j + 1 .. j + B): the
            // chain of a step is twelve v_readlane (u_k to every lane) and six multiply-adds — no LDS round trip, no triangular solve.
            // One wavefront issues an instruction every ~6 cycles whatever it is: the step is as long as its instruction list, so
            // nothing is masked that need not be (lanes without a block compute garbage nobody reads; every address is a valid one).
            // (The LDS form below, kept for bands wider than ten blocks, pays a store, a fence and a load per step.)
            int kr = (k1 - 1) % RR;
            const int slot = lane / 6, cc = lane - 6 * slot;
            const bool lane_on = slot < W;
            if (k1 == Npf) {                                        // the window of the last block row and that row itself
                const int jm = (Npf - 1) - ((Npf - 1 - slot) % W + W) % W;     // the block j <= Npf - 1 with j mod W == slot
                zreg = (lane_on && jm >= 0) ? cvec[6 * jm + cc] : 0.0;
            }
            // this lane's block in the window of row k: m = (k - slot) mod W (0: the row's own block), block j = k - m; m and k mod W
            // step down with k (a modulo by a run-time W is some forty instructions).  The operands of the next step are loaded into
            // the OTHER register set before this step's chain starts.
            int mk = (((k1 - 1) - slot) % W + W) % W, ks = (k1 - 1) % W;
            auto load_operands = [&](const int k, const int m, const int krow, double L[6], double& zin) {
                const double* src = ring + (size_t)krow * rowsz + 36 * m + cc;       // (m == 0: the row's own block — values unused)
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) L[rr] = src[6 * rr];
                // the block that takes the slot of block k once u_k is out: k - W, first reached by row k - 1
                zin = cvec[max(0, 6 * (k - W) + cc)];
            };
            auto step = [&](const int k, const double L[6], const double zin) {
                double uk[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) uk[c] = readlane_f64(zreg, 6 * ks + c);
                double zn = zreg;
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) zn -= L[rr] * uk[rr];
                if (mk == 0 && lane_on) cvec[6 * k + cc] = zreg;     // u_k
                zreg = mk == 0 ? zin : zn;
                if (--kr < 0) kr = RR - 1;
                mk = mk == 0 ? W - 1 : mk - 1; ks = ks == 0 ? W - 1 : ks - 1;
            };
            double LA[6], zA, LB[6], zB;
            load_operands(k1 - 1, mk, kr, LA, zA);
            for (int k = k1 - 1; k >= k0; k -= 2) {
                // (the operands of row k - 1 are loaded even when that row belongs to the next chunk or does not exist: the ring row is
                // valid memory, the values are not used)
                { int krn = kr - 1; if (krn < 0) krn = RR - 1; load_operands(k - 1, mk == 0 ? W - 1 : mk - 1, krn, LB, zB); }
                step(k, LA, zA);
                if (k - 1 < k0) break;
                { int krn = kr - 1; if (krn < 0) krn = RR - 1; load_operands(k - 2, mk == 0 ? W - 1 : mk - 1, krn, LA, zA); }
                step(k - 1, LB, zB);
            }
        } else if (wave == 0) {
            int kr = (k1 - 1) % RR;
            double cf[21];
            load_factor(ring + (size_t)kr * rowsz, cf);
            for (int k = k1 - 1; k >= k0; --k) {
                // every row below has been taken out of z_k: x_k = C_k^-T z_k (every lane), then z_j -= W_kj^T x_k for the blocks to the left
                const double* rowk = ring + (size_t)kr * rowsz;
                int krn = kr - 1; if (krn < 0) krn = RR - 1;
                double zk[6], xk[6], cfn[21];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) zk[rr] = cvec[6 * k + rr];
                load_factor(ring + (size_t)(k > k0 ? krn : kr) * rowsz, cfn);      // the next pivot's factor, off the chain
                band_trsv_bwd(cf, zk, xk);
                const int nbk = min(B, k);
                for (int e = lane; e < 6 * nbk; e += 64) {
                    const int m = e / 6 + 1, cc = e - 6 * (m - 1);
                    const double* L = rowk + 36 * m;           // W_kj, j = k - m
                    double acc = cvec[6 * (k - m) + cc];
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) acc -= L[6 * rr + cc] * xk[rr];
                    cvec[6 * (k - m) + cc] = acc;
                }
                if (lane < 6) {
                    double xl = xk[0];
#pragma unroll
                    for (int c = 1; c < 6; ++c) xl = lane == c ? xk[c] : xl;
                    cvec[6 * k + lane] = xl;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 21; ++q) cf[q] = cfn[q];
                kr = krn;
            }
        }
        __syncthreads();
        if (k1 == Npf) BAND_STAMP(115);
        if (fast_bwd) {
            // x_k = X_k^T u_k for the rows of this chunk (their X_k leave the ring with the next one), and K8 (pose oplus) with it
            const int sel = st->sel;
            for (int k = k0 + tid; k < k1; k += BAND_T) {
                double x[21], u[6], dx[6];
                load_factor(ring + (size_t)(k % RR) * rowsz, x);
#pragma unroll
                for (int i = 0; i < 6; ++i) u[i] = cvec[6 * k + i];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double acc = x[tri6(c, c)] * u[c];
#pragma unroll
                    for (int i = c + 1; i < 6; ++i) acc += x[tri6(i, c)] * u[i];
                    dx[c] = acc;
                    g.x[6 * k + c] = acc;
                }
                const int ip = g.free_pose[k];
                pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
            }
            __syncthreads();
        }
        k1 = k0;
    }
    BAND_STAMP(111);
    if (!fast_bwd) {
        for (int t = tid; t < 6 * Npf; t += BAND_T) g.x[t] = cvec[t];
        const int sel = st->sel;
        for (int a = tid; a < Npf; a += BAND_T) {
            const int ip = g.free_pose[a];
            double dx[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) dx[q] = cvec[6 * a + q];
            pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
        }
    }
    BAND_STAMP(112);
#ifdef VISFS_BA_STAMPS
    if (tid == 0) sstamp[121] = __builtin_readcyclecounter();
    __syncthreads();
    if (tid < 126) g.stamps[tid] = sstamp[tid];
#endif
}

// K7 + trial chi2 for one landmark handled by G lanes: dl = (Hll + lambda I)^-1 (b_l - sum_i Hpl_il^T x_i), the trial point,
// and the robust chi2 of its edges at the trial state.  sRt = trial poses, sRt0 = poses of the linearisation point.
// DL (Optimizer/Framework=1 with the DOGLEG strategy): 1 = pass A — the regularised Gauss-Newton landmark step is back-substituted and kept
// (dxl), and instead of the trial evaluation the lane accumulates the inner products of the dogleg construction: chi_acc <- ||J v||^2
// (v = g / m, the direction of the Cauchy point; per robustified residual block), scale_acc <- ||g_s||^2, *step_acc <- ||gn_s||^2,
// *dot_acc <- g_s . gn_s (landmark shares);  2 = pass B — the landmark step is dl_A v + dl_B dn, then the usual trial evaluation.
template <int G, bool STG = true, int DL = 0>
__device__ __forceinline__ void backsub_landmark(const DeviceGraph& g, const LinBuf& L, const int l, const bool lvalid, const int sub, const PoseSrc<STG> Pt, const PoseSrc<STG> P0,
                                                 const double* __restrict__ pt, double* __restrict__ pt_t, const double lambda, const Intrinsics& K,
                                                 const double iv, const double delta, double& chi_acc, double& scale_acc, double* step_acc = nullptr,
                                                 double* dot_acc = nullptr, const double dlA = 0.0, const double dlB = 1.0, Vec3* pn_out = nullptr) {
    int k0 = 0, k1 = 0;
    Vec3 pw{ 0, 0, 0 };
    bool lfree = false;
    if (lvalid) {
        k0 = g.lm_ptr[l]; k1 = g.lm_ptr[l + 1];
        pw = Vec3{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] };
        lfree = !g.pt_fixed[l];
    }
    // c_l = b_l - sum_i Hpl_il^T x_i
    double t0 = 0, t1 = 0, t2 = 0, any = 0.0;
    if (DL != 2)
    for (int k = k0 + sub; k < k1; k += G) {
        const double w = L.obs_w[k];
        if (w == 0.0) continue;
        any = 1.0;
        const int ipk = g.obs_pose[k];
        const int a = g.pose_free[ipk];
        if (a < 0 || !lfree) continue;
        const double2* seed = reinterpret_cast<const double2*>(L.obs_pcw + 4 * (size_t)k);
        const double2 s0 = seed[0], s1 = seed[1];
        // Hpl^T x through the tile structure W = [N ; [Pc]x N]:  W^T x = N^T (x_t - Pc x x_r)
        const Vec3 pcs{ s0.x, s0.y, s1.x };
        double N[9];
        tile_core(P0.get(ipk).R, pcs, s1.y, K, N);
        const double* xp = g.x + 6 * (size_t)a;
        const double u0 = xp[0] - (pcs.y * xp[5] - pcs.z * xp[4]);
        const double u1 = xp[1] - (pcs.z * xp[3] - pcs.x * xp[5]);
        const double u2 = xp[2] - (pcs.x * xp[4] - pcs.y * xp[3]);
        t0 += N[0] * u0 + N[3] * u1 + N[6] * u2;
        t1 += N[1] * u0 + N[4] * u1 + N[7] * u2;
        t2 += N[2] * u0 + N[5] * u1 + N[8] * u2;
    }
    t0 = group_sum<G>(t0); t1 = group_sum<G>(t1); t2 = group_sum<G>(t2);
    any = group_max<G>(any);
    double d0 = 0, d1 = 0, d2 = 0;
    double vl0 = 0.0, vl1 = 0.0, vl2 = 0.0;                    // dogleg: v = g / m = -b_l / m_l of this landmark
    double m0 = 1.0, m1 = 1.0, m2 = 1.0;
    if (DL != 0 && lvalid && lfree) {
        const double* H = L.Hll + 6 * (size_t)l;
        const double* Bq = L.bl + 3 * (size_t)l;
        m0 = damp_of(g, 1.0, H[0], g.s2l, 3 * (size_t)l); m1 = damp_of(g, 1.0, H[3], g.s2l, 3 * (size_t)l + 1); m2 = damp_of(g, 1.0, H[5], g.s2l, 3 * (size_t)l + 2);
        vl0 = -Bq[0] / m0; vl1 = -Bq[1] / m1; vl2 = -Bq[2] / m2;
        if (DL == 2) {
            d0 = dlA * vl0 + dlB * g.dxl[3 * (size_t)l]; d1 = dlA * vl1 + dlB * g.dxl[3 * (size_t)l + 1]; d2 = dlA * vl2 + dlB * g.dxl[3 * (size_t)l + 2];
            if (sub == 0 && step_acc) *step_acc += d0 * d0 + d1 * d1 + d2 * d2;
        }
    }
    if (DL != 2 && lvalid && lfree && any != 0.0) {
        const double* H = L.Hll + 6 * (size_t)l;
        const double* B = L.bl + 3 * (size_t)l;
        double a0 = lambda, a1 = lambda, a2 = lambda;
        if (g.ceres) { a0 = damp_of(g, lambda, H[0], g.s2l, 3 * (size_t)l); a1 = damp_of(g, lambda, H[3], g.s2l, 3 * (size_t)l + 1); a2 = damp_of(g, lambda, H[5], g.s2l, 3 * (size_t)l + 2); }
        const double h[6] = { H[0] + a0, H[1], H[2], H[3] + a1, H[4], H[5] + a2 };
        double D[6];
        sym3_inverse(h, D);
        const double c0 = B[0] - t0, c1 = B[1] - t1, c2 = B[2] - t2;
        d0 = D[0] * c0 + D[1] * c1 + D[2] * c2;
        d1 = D[1] * c0 + D[3] * c1 + D[4] * c2;
        d2 = D[2] * c0 + D[4] * c1 + D[5] * c2;
        if (DL == 0) {
            if (sub == 0) scale_acc += d0 * (a0 * d0 + B[0]) + d1 * (a1 * d1 + B[1]) + d2 * (a2 * d2 + B[2]);
            if (sub == 0 && step_acc) *step_acc += d0 * d0 + d1 * d1 + d2 * d2;      // Optimizer/Framework=1: the landmark's share of ||x - candidate||^2
        } else if (sub == 0) {
            // pass A:  ||g_s||^2 += b_c^2 / m_c, ||gn_s||^2 += m_c dn_c^2, g_s . gn_s += g_c dn_c
            scale_acc += B[0] * B[0] / m0 + B[1] * B[1] / m1 + B[2] * B[2] / m2;
            *step_acc += m0 * d0 * d0 + m1 * d1 * d1 + m2 * d2 * d2;
            *dot_acc -= B[0] * d0 + B[1] * d1 + B[2] * d2;
        }
    }
    const Vec3 pn{ pw.x + d0, pw.y + d1, pw.z + d2 };         // VertexPointXYZ::oplus
    if (pn_out) *pn_out = pn;                                // (every lane of the group holds the same value)
    if (lvalid && sub == 0) {
        pt_t[3 * l] = pn.x; pt_t[3 * l + 1] = pn.y; pt_t[3 * l + 2] = pn.z;
        if (DL != 2) { g.dxl[3 * (size_t)l] = d0; g.dxl[3 * (size_t)l + 1] = d1; g.dxl[3 * (size_t)l + 2] = d2; }
    }
    if (DL == 1) {
        // ||J v||^2 of this landmark's residual blocks at the linearisation point: J_e v = Jp v_l + Jx v_p, weighted rho' / sigma^2
        for (int k = k0 + sub; k < k1; k += G) {
            const double w = L.obs_w[k];
            if (w == 0.0) continue;
            const int ipk = g.obs_pose[k];
            const int a = g.pose_free[ipk];
            const double2* seed = reinterpret_cast<const double2*>(L.obs_pcw + 4 * (size_t)k);
            const double2 s0 = seed[0], s1 = seed[1];
            const Vec3 pcs{ s0.x, s0.y, s1.x };
            double Jp[9], Jx[18];
            stereo_jacobians(P0.get(ipk), pcs, K, Jp, Jx);
            double r0 = 0.0, r1 = 0.0, r2 = 0.0;
            if (lfree) {
                r0 = Jp[0] * vl0 + Jp[1] * vl1 + Jp[2] * vl2; r1 = Jp[3] * vl0 + Jp[4] * vl1 + Jp[5] * vl2; r2 = Jp[6] * vl0 + Jp[7] * vl1 + Jp[8] * vl2;
            }
            if (a >= 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double vp = -g.bp[6 * (size_t)a + c] / damp_of(g, 1.0, g.Hpp[36 * (size_t)a + 7 * c], g.s2p, 6 * (size_t)a + c);
                    r0 += Jx[c] * vp; r1 += Jx[6 + c] * vp; r2 += Jx[12 + c] * vp;
                }
            }
            chi_acc += w * iv * (r0 * r0 + r1 * r1 + r2 * r2);
        }
        return;
    }
    // computeActiveErrors + activeRobustChi2 at the trial state
    for (int k = k0 + sub; k < k1; k += G) {
        if (L.obs_w[k] == 0.0) continue;
        const Rt T = Pt.get(g.obs_pose[k]);
        Vec3 pc;
        const Vec3 e = stereo_error(T, pn, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
        const double c2 = chi2_of(e, iv);
        double rho0 = c2, rho1 = 1.0;
        robustify(g, c2, delta, rho0, rho1);
        chi_acc += rho0;
    }
}

// ================================================================= K7/K8 + chi2 at the trial state
// ODOSPEC (speculative unit of a window with odometry / laser edges): the workgroup that evaluates those edges at the trial poses
// also linearises them there, into the set the speculative k_linearize is about to fill — k_odo_linearize (6.6 us, one
// workgroup) leaves the unit.  The role is register-hungry (224 VGPRs), so this instantiation runs at two waves per SIMD: it is
// used only where k_backsub is a single round of waves anyway (the speculative unit's size limit).
// DEC (the gated unit: batched windows, large windows, Optimizer without the speculative unit): one more workgroup per window, the
// last ones of the launch, takes the LM decision on the trial (decide_gather_role) — k_decide (6.3 us + a launch gap per unit) leaves.
// The batched instantiation drifts from 90 to 114 VGPRs with the role on board (one wave per SIMD less for a bandwidth-bound
// launch): it is held at five waves per SIMD (96 VGPRs, a dozen spill slots outside the loops).
// LINA (single window, two linearisation sets; with DEC): the fused tail of the speculative unit — every landmark workgroup, having
// published its share of the trial's chi2, goes on to linearise ITS landmarks at the trial state into the other set (role A of
// k_linearize: the trial landmark is in registers, the trial poses are staged), the odometry workgroup does the same for its edges
// (ODOSPEC), and the decider, which flips lin_sel on acceptance, marks the pose-major sums as pending (LmState::lin_b_pending): they
// need EVERY trial landmark, so the role-B workgroups of the next k_schur_partial launch form them.  One launch (and its ~5 us of
// dependent-dispatch latency) less per iteration than k_backsub followed by k_linearize<SPEC>.
#ifndef VISFS_BA_BATCH_LINA_WAVES
#define VISFS_BA_BATCH_LINA_WAVES 4      // waves per SIMD the batched fused tail is compiled for (A/B builds)
#endif
template <int G, class Src, bool ODOSPEC, bool STG = true, bool DEC = false, int DL = 0, bool LINA = false>
__global__ __launch_bounds__(256, (DEC && Src::batched) ? (LINA ? (ODOSPEC ? 2 : VISFS_BA_BATCH_LINA_WAVES) : 5) : 1) void k_backsub(const Src src) {
    static_assert(!(DEC && ODOSPEC) || LINA, "the decision rides on the gated unit only");
    static_assert(DL == 0 || (!DEC && !ODOSPEC), "the dogleg passes are plain launches");
    static_assert(!LINA || (DEC && DL == 0), "the fused tail carries the decision");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (DEC) {
        const int dw = decider_window();
        if (dw >= 0) { decider_of(src, dw, smem, LINA ? 2 : 0); return; }
        if (dw == -2) return;
    }
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
#ifdef VISFS_BA_STAMPS
#define BS_STAMP(slot) do { if (LINA && threadIdx.x == 0 && blockIdx.x == (unsigned)g.stamp_wg) g.stamps[32 + (slot)] = wall_clock64(); } while (0)
#else
#define BS_STAMP(slot) do { } while (0)
#endif
    BS_STAMP(0);
    const bool trial = (st->mode & MODE_TRIAL) != 0;
    const bool go = trial && !st->solver_failed && !st->pcg_timeout;
    // a failed solve: nothing to compute, but the decision may only be taken once every workgroup has read the gate
    // (the tag is read where it is used: nothing of the decision's bookkeeping stays live across the landmark role)
    if (DEC && trial && !go && (int)blockIdx.x <= g.n_lin_a && threadIdx.x == 0) publish_trial(g, blockIdx.x, st->decide_epoch + 1u, 0.0, 0.0);
    // snapshot for the speculative linearisation that may follow (its workgroups must not read what the LM decision writes)
    if (!DEC && !LINA && !Src::batched && blockIdx.x == 0 && threadIdx.x == 0) { st->spec_go = go ? 1 : 0; st->spec_src = st->sel ^ 1; st->spec_dst = st->lin_sel ^ 1; }
    if (!go) return;
    double* sRt = smem;
    double* red = smem + (STG ? 12 * g.Np : 0);
    const int sel = st->sel, ls = st->lin_sel;
    const double* __restrict__ pose_t = g.pose[sel ^ 1];      // trial poses (written by the solver epilogue)
    const double* __restrict__ pt = g.pt[sel];
    double* __restrict__ pt_t = g.pt[sel ^ 1];
    const double lambda = st->lambda;
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta;
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (bid > g.n_lin_a) return;
    if (DL == 1 && bid == g.n_lin_a) {
        // dogleg pass A: the laser residual blocks' share of ||J v||^2 (J at the linearisation pose, v_p = -b_p / m_p of the newest pose)
        double jv = 0.0;
        if (g.Nz > 0 && g.pose_free[g.laser_pose] >= 0) {
            const int a = g.pose_free[g.laser_pose];
            double vp[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) vp[c] = -g.bp[6 * (size_t)a + c] / damp_of(g, 1.0, g.Hpp[36 * (size_t)a + 7 * c], g.s2p, 6 * (size_t)a + c);
            for (int z = tid; z < g.Nz; z += 256) {
                double J[6];
                laser_jacobian(g.pose[sel] + POSE_STRIDE * g.laser_pose, g.Tcr, Vec3{ g.laser_xyz[3 * z], g.laser_xyz[3 * z + 1], g.laser_xyz[3 * z + 2] }, g.grid, J, true);
                double r = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) r += J[c] * vp[c];
                jv += g.inv_laser_cov * r * r;
            }
        }
        const double jv_tot = block_sum_256(jv, red);
        if (tid == 0) { g.dl_part[4 * (size_t)bid] = jv_tot; g.dl_part[4 * (size_t)bid + 1] = 0.0; g.dl_part[4 * (size_t)bid + 2] = 0.0; g.dl_part[4 * (size_t)bid + 3] = 0.0; }
        return;
    }
    if (bid == g.n_lin_a) {
        // odometry chi2 at the trial state (the pose part of computeScale is done in k_decide)
        const double ic = g.inv_odo_cov;
        double chi_acc = 0.0;
        for (int e_ = tid; e_ < g.Ne; e_ += 256) {
            const int i = g.odo_i[e_], j = g.odo_j[e_];
            if (g.pose_free[i] < 0 && g.pose_free[j] < 0) continue;
            double e[6];
            odo_error(pose_t + POSE_STRIDE * i, pose_t + POSE_STRIDE * j, g.odo_tq + 7 * e_, e);
#pragma unroll
            for (int d = 0; d < 6; ++d) chi_acc += e[d] * (ic * e[d]);
        }
        for (int z = tid; z < g.Nz; z += 256) {       // laser edges at the trial pose
            const double e = laser_error(pose_t + POSE_STRIDE * g.laser_pose, g.Tcr, Vec3{ g.laser_xyz[3 * z], g.laser_xyz[3 * z + 1], g.laser_xyz[3 * z + 2] }, g.grid);
            chi_acc += e * (g.inv_laser_cov * e);
        }
        const double chi_tot = block_sum_256(chi_acc, red);
        if (tid == 0) { if (DEC) publish_trial(g, bid, st->decide_epoch + 1u, chi_tot, 0.0); else { g.trial_part[2 * bid] = chi_tot; g.trial_part[2 * bid + 1] = 0.0; } }
        if (ODOSPEC) {
            __syncthreads();
            const LinSel<Src> lspec(g, ls ^ 1);                      // == spec_dst of the snapshot above
            (void)odo_role(g, lspec.get(), pose_t, smem, smem + 8);  // this workgroup stages no poses: the LDS is free
        }
        return;
    }
    double* sRt0 = red + 8;                   // poses of the linearisation point (tiles are rebuilt there)
    BS_STAMP(1);
    if (STG) {
        stage_poses(pose_t, g.Np, sRt);
        stage_poses(g.pose[sel], g.Np, sRt0);
        __syncthreads();
    }
    BS_STAMP(2);
    const PoseSrc<STG> Pt{ STG ? sRt : pose_t }, P0{ STG ? sRt0 : (const double*)g.pose[sel] };
    const LinSel<Src> lsel(g, ls); const LinBuf& L = lsel.get();
    constexpr int LPW = 256 / G;
    const int l = bid * LPW + tid / G, sub = tid % G;
    const bool lvalid = l < g.Nl;
    double chi_acc = 0.0, scale_acc = 0.0, step_acc = 0.0, dot_acc = 0.0;
    Vec3 pn{ 0.0, 0.0, 0.0 };
    backsub_landmark<G, STG, DL>(g, L, l, lvalid, sub, Pt, P0, pt, pt_t, lambda, K, iv, delta, chi_acc, scale_acc, &step_acc, &dot_acc, st->dl_A, st->dl_B, LINA ? &pn : nullptr);
    BS_STAMP(3);
    const double chi_tot = block_sum_256(chi_acc, red);
    const double sc_tot = block_sum_256(scale_acc, red);
    BS_STAMP(4);
    if (DL == 1) {
        const double s2_tot = block_sum_256(step_acc, red), s3_tot = block_sum_256(dot_acc, red);
        if (tid == 0) { g.dl_part[4 * (size_t)bid] = chi_tot; g.dl_part[4 * (size_t)bid + 1] = sc_tot; g.dl_part[4 * (size_t)bid + 2] = s2_tot; g.dl_part[4 * (size_t)bid + 3] = s3_tot; }
        return;
    }
    if (g.ceres) { const double st_tot = block_sum_256(step_acc, red); if (tid == 0) g.aux_part[bid] = st_tot; }
    if (tid == 0) { if (DEC) publish_trial(g, bid, st->decide_epoch + 1u, chi_tot, sc_tot); else { g.trial_part[2 * bid] = chi_tot; g.trial_part[2 * bid + 1] = sc_tot; } }
    if (LINA) {
        // role A of the speculative linearisation for this workgroup's landmarks (LmState is not read from here on: the decider may
        // already be rewriting it — sel / ls are the values read at the top)
        const LinSel<Src> lspec(g, ls ^ 1);
        double chi2 = 0.0, md2 = 0.0;
        lin_landmark<G, STG>(g, lspec.get(), l, lvalid, sub, Pt, pt_t, K, iv, delta, chi2, md2, nullptr, &pn);
        // (no lin_part here: the chi2 of an accepted trial is the trial's, max |diag H| is only read at the first iteration of a phase,
        // which linearises with k_linearize)
        BS_STAMP(5);
    }
}

// Between the two dogleg passes (one workgroup): the inner products summed in a fixed order, the step coefficients and the model cost
// change (dogleg_combine), then the trial poses x (+) (dl_A v_p + dl_B dn_p) — the solver's epilogue left x (+) dn_p there.
template <class Src>
__global__ __launch_bounds__(256) void k_dogleg_mid(const Src src) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    __shared__ double red[4];
    if (!(st->mode & MODE_TRIAL) || st->solver_failed || st->pcg_timeout) return;
    const int tid = threadIdx.x;
    double jv = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int w = tid; w < g.n_lin_a + 1; w += 256) { jv += g.dl_part[4 * (size_t)w]; s1 += g.dl_part[4 * (size_t)w + 1]; s2 += g.dl_part[4 * (size_t)w + 2]; s3 += g.dl_part[4 * (size_t)w + 3]; }
    for (int t = tid; t < 6 * g.Npf; t += 256) {
        const double m = damp_of(g, 1.0, g.Hpp[36 * (size_t)(t / 6) + 7 * (t % 6)], g.s2p, t), b = g.bp[t], x = g.x[t];
        s1 += b * b / m; s2 += m * x * x; s3 -= b * x;
    }
    jv = block_sum_256(jv, red); s1 = block_sum_256(s1, red); s2 = block_sum_256(s2, red); s3 = block_sum_256(s3, red);
    double A, B, norm, mcc;
    dogleg_combine(s1, s2, s3, jv, st->tr_radius, st->dl_mu, A, B, norm, mcc);     // (every thread: the same inputs, the same result)
    __syncthreads();
    if (tid == 0) { st->dl_A = A; st->dl_B = B; st->dl_step_norm = norm; st->dl_mcc = mcc; }
    const int sel = st->sel;
    for (int a = tid; a < g.Npf; a += 256) {
        const int ip = g.free_pose[a];
        double dx[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int t = 6 * a + c;
            const double m = damp_of(g, 1.0, g.Hpp[36 * (size_t)a + 7 * c], g.s2p, t);
            dx[c] = A * (-g.bp[t] / m) + B * g.x[t];
        }
        pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
    }
}

// ================================================================= K9: Levenberg-Marquardt control (lm_decide / decide_role: above k_linearize)
template <class Src>
__global__ __launch_bounds__(256) void k_decide(const Src src) {
    const DeviceGraph& g = graph_of(src);
    __shared__ double red[4];
    decide_role(g, state_of(src, g), red, false);
}

// ================================================================= K10: per-edge chi2, outlier marking
// Optimizer.cpp:270-303: computeActiveErrors; edges with chi2() > kernel->delta() (UNSQUARED) go to level 1.
template <class Src, bool STG = true>
__global__ __launch_bounds__(256) void k_eval(const Src src, const int mark, const int phase_just_done) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    if (st->status != 0) return;
    // the host may enqueue a phase end before it knows that the phase is over (one state read per solve): act only then
    if (phase_just_done >= 0 && (!st->done || st->ended != phase_just_done)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sRt = smem;
    double* red = smem + (STG ? 12 * g.Np : 0);
    const int sel = st->sel;
    const double* __restrict__ pt = g.pt[sel];
    if (STG) { stage_poses(g.pose[sel], g.Np, sRt); __syncthreads(); }
    const PoseSrc<STG> P{ STG ? sRt : (const double*)g.pose[sel] };
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta;
    const int tid = threadIdx.x, bid = blockIdx.x;
    const int nb = (g.No + 255) / 256 + 1;                 // this window's blocks (== gridDim.x for a single window)
    if (bid >= nb) return;
    double chi_acc = 0.0;
    int n_out = 0;
    if (bid == nb - 1) {
        const double ic = g.inv_odo_cov;
        const double* pose = g.pose[sel];
        for (int e_ = tid; e_ < g.Ne; e_ += 256) {
            const int i = g.odo_i[e_], j = g.odo_j[e_];
            if (g.pose_free[i] < 0 && g.pose_free[j] < 0) continue;
            double e[6];
            odo_error(pose + POSE_STRIDE * i, pose + POSE_STRIDE * j, g.odo_tq + 7 * e_, e);
#pragma unroll
            for (int d = 0; d < 6; ++d) chi_acc += e[d] * (ic * e[d]);
        }
        for (int z = tid; z < g.Nz; z += 256) {
            const double e = laser_error(pose + POSE_STRIDE * g.laser_pose, g.Tcr, Vec3{ g.laser_xyz[3 * z], g.laser_xyz[3 * z + 1], g.laser_xyz[3 * z + 2] }, g.grid);
            chi_acc += e * (g.inv_laser_cov * e);
        }
    } else {
        const int k = bid * 256 + tid;
        if (k < g.No) {
            // (Ceres flavour, Optimizer.cpp:529-540: EVERY stereo residual block is tested, those between two constant blocks included,
            // with error . (pixelInfo error) — not the objective's ||info error||^2)
            const bool in_objective = g.ceres ? (g.obs_ok[k] != 0) : ((g.obs_level[k] == 0) && g.obs_ok[k]);   // (no levels in the Ceres branch: one pass)
            const bool active = g.ceres ? true : in_objective;
            double c2 = 0.0;
            if (active) {
                const int l = g.obs_pt[k];
                const Rt T = P.get(g.obs_pose[k]);
                Vec3 pc;
                const Vec3 e = stereo_error(T, Vec3{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] }, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
                c2 = chi2_of(e, iv);
                double rho0 = c2, rho1 = 1.0;
                robustify(g, c2, delta, rho0, rho1);
                chi_acc = in_objective ? rho0 : 0.0;
                if (g.ceres) c2 = chi2_of(e, g.inv_pixel_var_out);
                if (mark && delta > 0.0 && c2 > delta) { g.obs_level[k] = 1; g.obs_outlier[k] = 1; n_out = 1; }
            }
            if (mark) g.obs_chi2_out[k] = c2;
        }
    }
    const double chi_tot = block_sum_256(chi_acc, red);
    const double out_tot = block_sum_256((double)n_out, red);
    if (tid == 0) { g.trial_part[2 * bid] = chi_tot; g.trial_part[2 * bid + 1] = out_tot; }
}

// One workgroup: closes a phase (Optimizer.cpp:271-280 after phase 1, :315-318 after phase 2) and arms the next.
// One thread: closes a phase (Optimizer.cpp:271-280 after phase 1, :315-318 after phase 2) and arms the next.
__device__ __noinline__ void phase_end_update(const DeviceGraph& g, LmState* st, const double chi, const double nout, const int phase_just_done, const int next_max_iter) {
    st->ended = phase_just_done + 1;
    if (phase_just_done == 0) {
        st->chi2_phase1 = chi; st->chi2_final = chi;
        st->pcg_phase1 = st->pcg_total;
        if (st->max_iter <= 0) st->chi2_initial = chi;        // optimize(0): nothing linearised
        if (g.ceres) { }                                                  // the Ceres branch has no chi2 guards (Optimizer.cpp:504-540)
        else if (chi != chi) st->status = 3;                              // VISFS_BA_ERR_NAN_CHI2
        else if (chi > 1000000000000.0 || !(chi <= DBL_MAX)) st->status = 4; // VISFS_BA_ERR_HUGE_CHI2_1
        st->n_outliers = (int)nout;
        // arm phase 2: initializeOptimization(0) + optimize(iterations/2) re-initialise lambda and the PCG residual
        st->phase = 1; st->phase_iter = 0; st->max_iter = next_max_iter; st->trial_q = 0;
        st->solver_failed = 0; st->pcg_residual = -1.0;
        st->done = (st->status != 0 || next_max_iter <= 0 || g.huber_delta <= 0.0) ? 1 : 0;
        st->mode = st->done ? 0 : (MODE_LIN | MODE_TRIAL); st->lin_b_pending = 0;
    } else {
        st->chi2_final = chi;
        if (!g.ceres && chi > 1000000000000.0) st->status = 5;           // VISFS_BA_ERR_HUGE_CHI2_2
    }
}

template <class Src>
__global__ __launch_bounds__(256) void k_phase_end(const Src src, const int phase_just_done, const int next_max_iter) {
    const DeviceGraph& g = graph_of(src);
    const int nparts = (g.No + 255) / 256 + 1;          // partials written by k_eval
    LmState* st = state_of(src, g);
    if (st->status != 0) return;
    if (!st->done || st->ended != phase_just_done) return;      // (same gate as k_eval, which does not modify the state)
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double chi = 0.0, nout = 0.0;
    for (int w = tid; w < nparts; w += 256) { chi += g.trial_part[2 * w]; nout += g.trial_part[2 * w + 1]; }
    chi = block_sum_256(chi, red);
    nout = block_sum_256(nout, red);
    if (tid != 0) return;
    phase_end_update(g, st, chi, nout, phase_just_done, next_max_iter);
}

// Arm phase 1 on a fresh graph (all edges level 0, as the reference builds a new optimizer per call);
// restore != 0 also rewinds the estimates to the uploaded ones.
template <class Src>
__global__ __launch_bounds__(256) void k_reset(const Src src, const int max_iter, const int gauss_newton, const int restore) {
    const DeviceGraph& g = graph_of(src);
    const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (restore) {
        for (int t = gid; t < g.Np * POSE_STRIDE; t += stride) { const double v = g.pose0[t]; g.pose[0][t] = v; g.pose[1][t] = v; }
        for (int t = gid; t < g.Nl * 3; t += stride) { const double v = g.pt0[t]; g.pt[0][t] = v; g.pt[1][t] = v; }
    }
    for (int t = gid; t < g.No; t += stride) { g.obs_level[t] = 0; g.obs_outlier[t] = 0; g.obs_chi2_out[t] = 0.0; }
    if (g.fin_arrive) for (int t = gid; t < g.n_blk; t += stride) g.fin_cnt[t] = 0u;
    if (g.fin_pcg) {
        // the fused finalisation + PCG launch: its flags and hand-off words start every optimise call at zero (tags count the call's units)
        for (int t = gid; t < g.n_blk; t += stride) g.fin_flag[t] = 0u;
        for (int t = gid; t < 4 * 6 * g.Npf + g.Npf; t += stride) g.granules[t] = 0ull;
    }
    if (gid == 0) {
        LmState* st = state_of(src, g);
        st->lambda = 0.0; st->ni = 2.0; st->current_chi = 0.0; st->temp_chi = 0.0; st->rho = 0.0; st->scale = 0.0; st->max_diag = 0.0;
        st->pcg_res_in = -1.0; st->pcg_residual = -1.0;
        st->chi2_initial = 0.0; st->chi2_phase1 = 0.0; st->chi2_final = 0.0;
        if (restore) st->sel = 0;
        st->pcg_max = 0; st->pcg_timeout = 0;
        st->lin_sel = 0; st->spec_go = 0; st->spec_src = 0; st->spec_dst = 1; st->lin_b_pending = 0; st->ended = 0; st->pcg_phase1 = 0; st->n_edges_ok = g.n_edges_ok;
        st->n_active[0] = st->n_active[1] = st->n_active[2] = st->n_active[3] = 0;
        st->phase = 0; st->max_iter = max_iter; st->phase_iter = 0; st->trial_q = 0;
        st->done = (max_iter <= 0) ? 1 : 0; st->mode = st->done ? 0 : (MODE_LIN | MODE_TRIAL); st->solver_failed = 0;
        st->pcg_iter = 0; st->pcg_total = 0; st->gauss_newton = gauss_newton; st->status = 0;
        st->n_outliers = 0; st->n_trace = 0;
        st->tr_radius = 1e4; st->tr_x_norm = 0.0; st->tr_invalid = 0; st->tr_reason = 0;
        st->dogleg = g.dogleg; st->dl_mu = 1e-8; st->dl_A = 0.0; st->dl_B = 1.0; st->dl_step_norm = 0.0; st->dl_mcc = 0.0;
        st->iterations_run[0] = st->iterations_run[1] = 0; st->trials_run[0] = st->trials_run[1] = 0;
    }
}

// Test hook: force the gates for a single stage call (visfs_ba_stage_*).
__global__ void k_stage_arm(const DeviceGraph g, const double lambda, const int mode) {
    LmState* st = g.st;
    st->done = 0; st->mode = mode; st->solver_failed = 0; st->lambda = lambda; st->phase_iter = 1;
    st->max_iter = 1 << 30; st->trial_q = 0; st->gauss_newton = 0;
}

// ================================================================= fused single-workgroup path (small windows)
// The production window of VISFS (LocalMap/MapSize = 5: six poses, ~300 landmarks, ~1500 observations) is far too small
// for one kernel per stage — a stage is a few microseconds of work behind a launch and a dependent-load chain.
// k_small_optimize runs BOTH optimise phases, the outlier pass and the final evaluation (Optimizer.cpp:261-318) in ONE
// launch of ONE 512-thread workgroup (256 VGPRs per lane: the inlined stage bodies spill at the 128 a 1024-thread block
// gets): stages are separated by __syncthreads() instead of kernel boundaries, the LM
// state machine (lm_decide / phase_end_update — the very functions the multi-kernel path runs) is stepped by thread 0
// and re-read by everyone after the barrier, the reduced camera system (<= 64 x 64) is solved in LDS.  All arithmetic
// goes through the same device functions as the multi-kernel path (lin_landmark<1>, pose_obs_terms, schur_chunk,
// schur_block, backsub_landmark<1>, ...); only the reduction trees differ, so the two paths agree to rounding.
constexpr int SM_T = 512;
constexpr int SM_WAVES = SM_T / 64;
constexpr int SM_LD = SM_MAX_N6 + 1;           // padded leading dimension of the dense S in LDS

// Sum `a` and `b` (separately) over the workgroup; every thread gets both.  Fixed order: wave butterfly, waves 0..15.
__device__ __forceinline__ void sm_block_sum2(double& a, double& b, double* red) {
    a = wave_sum(a); b = wave_sum(b);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < SM_WAVES; ++w) { sa += red[2 * w]; sb += red[2 * w + 1]; }
    a = sa; b = sb;
}
__device__ __forceinline__ double sm_block_max(double v, double* red) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < SM_WAVES; ++w) m = fmax(m, red[w]);
    return m;
}
__device__ __forceinline__ int ld_state(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// [g2o-upstream] LinearSolverPCG on the dense S in LDS: wave 0, lane r owns scalar row r (n6 <= 64).  Same recurrence,
// tolerance and residual carry-over as k_pcg.
// The same PCG with lane r's row of S in registers and the direction vector broadcast lane by lane with readlanes (uniform indices after
// unrolling) instead of through LDS: the mat-vec of the LDS form was n6 dependent round trips (two ds_reads, a wait and an FMA per
// column: 1.2 us per PCG iteration at the production window's order 30).  Same products, same order of the sums: bit-identical.
// Only instantiated in k_small_solve (the fused kernel keeps the LDS-row version: compile time).
template <int NMAX>
__device__ __forceinline__ void sm_pcg_reg(const DeviceGraph& g, LmState* st, const int n6, const double* sA, const double* sb, double* sx) {
    const int r = threadIdx.x & 63;
    const bool act = r < n6;
    const int blk0 = 6 * (r / 6);
    double m[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) m[c] = act ? g.Minv[36 * (size_t)(r / 6) + 6 * (r % 6) + c] : 0.0;
    double Arow[NMAX];
#pragma unroll
    for (int c = 0; c < NMAX; ++c) Arow[c] = (act && c < n6) ? sA[r * SM_LD + c] : 0.0;
    double rr = act ? sb[r] : 0.0;
    auto apply_minv = [&](const double v) {
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) sacc += m[c] * __shfl(v, (blk0 + c) & 63, 64);
        return sacc;
    };
    double d = apply_minv(rr);
    double dn = wave_sum(rr * d);
    double d0 = 1e-6 * dn;
    const double res_in = st->pcg_res_in;
    if (res_in > 0.0 && res_in > d0) d0 = res_in;
    double x = 0.0;
    int iter = 0;
    while (true) {
        if (dn <= d0 || iter >= n6 || !(dn == dn)) break;
        double q = 0.0;
#pragma unroll
        for (int c0 = 0; c0 < NMAX; c0 += 6) {
            if (c0 < n6) {                                   // (n6 is a multiple of 6 and wave-uniform: a scalar branch per block column)
#pragma unroll
                for (int c = c0; c < c0 + 6 && c < NMAX; ++c) q += Arow[c] * readlane_f64(d, c);
            }
        }
        const double dq = wave_sum(d * q);
        const double alpha = dn / dq;
        x += alpha * d;
        rr -= alpha * q;
        const double sv = apply_minv(rr);
        const double dnn = wave_sum(rr * sv);
        const double beta = dnn / dn;
        d = sv + beta * d;
        dn = dnn;
        iter += 1;
    }
    if (act) sx[r] = x;
    if (r == 0) {
        st->pcg_residual = 0.5 * dn;
        st->pcg_iter = iter;
        st->pcg_total += iter;
        if (iter > st->pcg_max) st->pcg_max = iter;
    }
}

__device__ __forceinline__ void sm_pcg(const DeviceGraph& g, LmState* st, const int n6, const double* sA, const double* sb, double* sd, double* sx) {
    const int r = threadIdx.x & 63;
    const bool act = r < n6;
    const int blk0 = 6 * (r / 6);
    double m[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) m[c] = act ? g.Minv[36 * (size_t)(r / 6) + 6 * (r % 6) + c] : 0.0;
    double rr = act ? sb[r] : 0.0;
    auto apply_minv = [&](const double v) {
        double sacc = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) sacc += m[c] * __shfl(v, (blk0 + c) & 63, 64);
        return sacc;
    };
    double d = apply_minv(rr);
    double dn = wave_sum(rr * d);
    double d0 = 1e-6 * dn;
    const double res_in = st->pcg_res_in;
    if (res_in > 0.0 && res_in > d0) d0 = res_in;
    double x = 0.0;
    int iter = 0;
    while (true) {
        if (dn <= d0 || iter >= n6 || !(dn == dn)) break;
        if (act) sd[r] = d;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        double q = 0.0;
        if (act) { const double* Ar = sA + r * SM_LD; for (int c = 0; c < n6; ++c) q += Ar[c] * sd[c]; }
        const double dq = wave_sum(d * q);
        const double alpha = dn / dq;
        x += alpha * d;
        rr -= alpha * q;
        const double sv = apply_minv(rr);
        const double dnn = wave_sum(rr * sv);
        const double beta = dnn / dn;
        d = sv + beta * d;
        dn = dnn;
        iter += 1;
        __builtin_amdgcn_wave_barrier();
    }
    if (act) sx[r] = x;
    if (r == 0) {
        st->pcg_residual = 0.5 * dn;
        st->pcg_iter = iter;
        st->pcg_total += iter;
        if (iter > st->pcg_max) st->pcg_max = iter;
    }
}

// Dense Cholesky of S in LDS by ONE wavefront, LEFT-looking (lane r owns row r of the lower triangle): for column c every
// lane forms s_r = A[r][c] - sum_{k<c} L[r][k] L[c][k] — the same c terms for every lane, two pipelined LDS reads per term,
// no store inside the loop —, the pivot s_c is broadcast with a readlane, L[r][c] = s_r / sqrt(s_c).  One wave barrier per
// column; the right-looking form needed two and a row update whose length differs per lane (33 us at order 30, now ~3x less).
// Then the two triangular solves.  A non-positive or non-finite pivot sets LmState::solver_failed.
__device__ __forceinline__ bool sm_cholesky_factor_lds(const int n, double* __restrict__ sA) {
    const int r = threadIdx.x & 63;
    const bool act = r < n;
    bool failed = false;
    for (int c = 0; c < n; ++c) {
        double sv = 0.0;
        if (act && r >= c) {
            const double* __restrict__ Ar = sA + r * SM_LD;
            const double* __restrict__ Ac = sA + c * SM_LD;
            sv = Ar[c];
            int k = 0;
            for (; k + 4 <= c; k += 4) sv -= Ar[k] * Ac[k] + Ar[k + 1] * Ac[k + 1] + Ar[k + 2] * Ac[k + 2] + Ar[k + 3] * Ac[k + 3];
            for (; k < c; ++k) sv -= Ar[k] * Ac[k];
        }
        const double p = readlane_f64(sv, c);               // lane c holds the pivot
        if (!(p > 0.0) || !(p <= DBL_MAX)) { failed = true; break; }
        const double inv = fast_rsqrt(p);
        if (act && r >= c) sA[r * SM_LD + c] = sv * inv;     // r == c: p / sqrt(p) = sqrt(p)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    return failed;
}
// L y = b, L^T x = y on one wavefront: the finished component is broadcast with a readlane (uniform index) instead of an
// LDS-crossbar shuffle, and the factor entries of four steps are loaded ahead of the dependent chain.
__device__ __forceinline__ void sm_cholesky_substitute(const int n, const double* __restrict__ sA, const double* __restrict__ sb, double* __restrict__ sx) {
    const int r = threadIdx.x & 63;
    const bool act = r < n;
    const double inv = act ? 1.0 / sA[r * SM_LD + r] : 1.0;
    double acc = act ? sb[r] : 0.0;
    for (int c0 = 0; c0 < n; c0 += 4) {                     // L y = b
        double l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) l[u] = (act && c0 + u < n && r > c0 + u) ? sA[r * SM_LD + c0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u;
            if (c < n) {
                const double yc = readlane_f64(acc * inv, c);
                acc = (r == c) ? yc : acc - l[u] * yc;
            }
        }
    }
    for (int c0 = n - 1; c0 >= 0; c0 -= 4) {                // L^T x = y
        double l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) l[u] = (act && c0 - u >= 0 && r < c0 - u) ? sA[(c0 - u) * SM_LD + r] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 - u;
            if (c >= 0) {
                const double xc = readlane_f64(acc * inv, c);
                acc = (r == c) ? xc : acc - l[u] * xc;
            }
        }
    }
    if (act) sx[r] = acc;
}
__device__ __forceinline__ void sm_cholesky_lds(LmState* st, const int n, double* __restrict__ sA, const double* __restrict__ sb,
                                                double* __restrict__ sx, double* __restrict__ /*scratch*/) {
    if (sm_cholesky_factor_lds(n, sA)) { if ((threadIdx.x & 63) == 0) st->solver_failed = 1; return; }
    sm_cholesky_substitute(n, sA, sb, sx);
}

// Order <= 32 (the production window: 5 free poses → 30): lane r keeps its row of L in REGISTERS (static indices, loops fully
// unrolled), row c is read from LDS as broadcasts: per column c FMAs fed by c pipelined LDS reads, the pivot by a readlane.
// Only instantiated in k_small_solve (the fused kernel keeps the LDS-row version: compile time).
template <int N>
__device__ __forceinline__ bool sm_cholesky_factor_reg(const int n, double* __restrict__ sA) {
    const int r = threadIdx.x & 63;
    const bool act = r < n;
    double a[N];
#pragma unroll
    for (int c = 0; c < N; ++c) a[c] = (act && c <= r && c < n) ? sA[r * SM_LD + c] : 0.0;
    bool failed = false;
    // Two columns per step: column c + 1 needs column c only through L[c+1][c], which lane c + 1 holds in a register by then (one
    // readlane) — so the LDS write -> fence -> barrier -> broadcast-read round trip (~200 of a column's ~640 fixed cycles) is
    // paid once per PAIR of columns.  Sums are taken in the order of the one-column form: the factor is bit-identical.
    // (Four columns per step: no faster at order 30, twice as slow at order 54 — eight accumulators beside the 64-entry row.)
#pragma unroll
    for (int c = 0; c < N; c += 2) {
        if (c < n && !failed) {                                   // uniform
            const double* __restrict__ Ac = sA + c * SM_LD;
            const double* __restrict__ Ad = sA + (c + 1) * SM_LD;
            const bool two = (c + 1 < n);                        // uniform
            double s0 = a[c], s1 = 0.0, t0 = a[c + 1], t1 = 0.0;
#pragma unroll
            for (int k = 0; k + 1 < c; k += 2) {
                s0 -= a[k] * Ac[k]; s1 -= a[k + 1] * Ac[k + 1];
                t0 -= a[k] * Ad[k]; t1 -= a[k + 1] * Ad[k + 1];
            }
            const double sv = s0 + s1;
            const double p = readlane_f64(sv, c);
            if (!(p > 0.0) || !(p <= DBL_MAX)) failed = true;
            else {
                const double l = sv * fast_rsqrt(p);
                a[c] = (r >= c) ? l : 0.0;
                if (act && r >= c) sA[r * SM_LD + c] = l;
                if (two) {
                    t0 -= a[c] * readlane_f64(a[c], c + 1);     // L[r][c] L[c+1][c]: the term the one-column form adds last
                    const double tv = t0 + t1;
                    const double q = readlane_f64(tv, c + 1);
                    if (!(q > 0.0) || !(q <= DBL_MAX)) failed = true;
                    else {
                        const double m = tv * fast_rsqrt(q);
                        a[c + 1] = (r >= c + 1) ? m : 0.0;
                        if (act && r >= c + 1) sA[r * SM_LD + c + 1] = m;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    return failed;
}

__device__ __forceinline__ void sm_solve(const DeviceGraph& g, LmState* st, const int solver, const int n6, double* sA, const double* sb, double* sd, double* sx) {
    if (solver == 2) sm_pcg(g, st, n6, sA, sb, sd, sx);
    else sm_cholesky_lds(st, n6, sA, sb, sx, sd);
}

// Small reduced camera systems (6 Npf <= 64): ONE workgroup does what k_schur_finalize + k_pcg (five cross-workgroup
// hand-offs per iteration for nothing) or + the four direct-solver launches do on the general path: S / b_s / Minv per
// stored block, the dense system in LDS, PCG or Cholesky on one wavefront, K8 (pose oplus).
template <class Src>
__global__ __launch_bounds__(512) void k_small_solve(const Src src, const int solver) {
    const DeviceGraph& g = graph_of(src);
    LmState* st = state_of(src, g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!(st->mode & MODE_TRIAL)) return;
    const LinSel<Src> lsel(g, st->lin_sel); const LinBuf& L = lsel.get();
    __shared__ double sA[SM_MAX_N6 * SM_LD];
    __shared__ double sb[SM_MAX_N6], sx[SM_MAX_N6], sd[SM_MAX_N6];
#ifdef VISFS_BA_STAMPS
#define SS_STAMP(slot) do { if (tid == 0) g.stamps[64 + (slot)] = wall_clock64(); } while (0)
#else
#define SS_STAMP(slot) do { } while (0)
#endif
    const int n6 = 6 * g.Npf;
    SS_STAMP(0);
    // S, b_s and the Jacobi preconditioner straight into LDS — the arithmetic of schur_block (same sums, same order) laid out flat:
    // a thread per (block, entry) sums the chunk partials with all its loads in flight, instead of a wave per block walking
    // dependent loads block after block (that loop was 7.4 us of this 15 us kernel on the production window).
    __shared__ double sHv[(SM_MAX_N6 / 6) * 42], sPt[(SM_MAX_N6 / 6) * 42];       // diagonal blocks: Hpp | b_p entries and their partial sums
    __shared__ int sDiag[SM_MAX_N6 / 6];                                          // stored block id of (a, a)
    for (int t = tid; t < n6 * SM_LD; t += 512) sA[t] = 0.0;
    __syncthreads();
    const double lambda = st->lambda;
    for (int t = tid; t < g.n_blk * 42; t += 512) {
        const int b = t / 42, q = t - 42 * b;
        const int4 bd = g.blk_desc[2 * b];           // (first chunk, last + 1, first odometry entry, last + 1)
        const int4 be = g.blk_desc[2 * b + 1];       // (i, j, first pose-major chunk of i, last + 1)
        const int i = be.x, j = be.y;
        const bool diag = (i == j);
        if (!diag && q >= 36) continue;
        double part = 0.0;
#pragma unroll 4
        for (int ch = bd.x; ch < bd.y; ++ch) part += g.sch_part[42 * (size_t)ch + q];
        if (!diag) {
            const int r = q / 6, c = q - 6 * r;
            double base = 0.0;
            for (int n = bd.z; n < bd.w; ++n) {
                const int code = g.blk_odo[n];
                base += L.odo_blk[120 * (size_t)(code >> 1) + 72 + ((code & 1) ? (c * 6 + r) : q)];
            }
            const double v = base - part;
            g.S[36 * (size_t)b + q] = v;
            sA[(6 * i + r) * SM_LD + 6 * j + c] = v;
            sA[(6 * j + c) * SM_LD + 6 * i + r] = v;
        } else {
            sHv[42 * i + q] = hpp_entry_r(g, L, q, be.z, be.w, bd.z, bd.w);
            sPt[42 * i + q] = part;
            if (q == 0) sDiag[i] = b;
        }
    }
    __syncthreads();
    for (int t = tid; t < g.Npf * 42; t += 512) {
        const int a = t / 42, q = t - 42 * a;
        const double hv = sHv[t], part = sPt[t];
        const double* hd = sHv + 42 * a;
        // poses without any active edge are outside g2o's active set: their block is pinned to I (dx = 0)
        const bool pin = hd[0] == 0.0 && hd[7] == 0.0 && hd[14] == 0.0 && hd[21] == 0.0 && hd[28] == 0.0 && hd[35] == 0.0;
        if (q < 36) {
            const int r = q / 6, c = q - 6 * r;
            const double val = pin ? (r == c ? 1.0 : 0.0) : (hv + (r == c ? damp_of(g, lambda, hv, g.s2p, 6 * (size_t)a + r) : 0.0) - part);
            g.S[36 * (size_t)sDiag[a] + q] = val;
            g.Hpp[36 * (size_t)a + q] = hv;
            sA[(6 * a + r) * SM_LD + 6 * a + c] = val;
        } else {
            const double bsv = pin ? 0.0 : (hv - part);
            g.bp[6 * (size_t)a + (q - 36)] = hv;
            g.bs[6 * (size_t)a + (q - 36)] = bsv;
            sb[6 * a + (q - 36)] = bsv;
        }
    }
    if (tid == 0) {
        st->pcg_res_in = st->pcg_residual;
        st->n_active[1] += 1;
        if (st->mode & MODE_LIN) st->n_active[0] += 1;
    }
    __syncthreads();
    SS_STAMP(1);
    for (int a = wave; a < g.Npf; a += 8) {                       // Minv_a = S_aa^-1 (block-Jacobi preconditioner)
        const double v = gauss_jordan_6x6(lane < 36 ? sA[(6 * a + lane / 6) * SM_LD + 6 * a + lane % 6] : 0.0, lane);
        if (lane < 36) g.Minv[36 * (size_t)a + lane] = v;
    }
    __threadfence_block();
    __syncthreads();
    SS_STAMP(2);
    if (wave == 0) {
        if ((solver & 0xff) == 2) {
            if (solver & 0x100) sm_pcg(g, st, n6, sA, sb, sd, sx);                   // (A/B runs: VISFS_BA_SMALL_PCG_LDS=1, the LDS-row form)
            else if (n6 <= 36) sm_pcg_reg<36>(g, st, n6, sA, sb, sx); else sm_pcg_reg<60>(g, st, n6, sA, sb, sx);
        }
        else {
            const bool failed = (n6 <= 32) ? sm_cholesky_factor_reg<32>(n6, sA) : sm_cholesky_factor_reg<64>(n6, sA);
            if (failed) { if (lane == 0) st->solver_failed = 1; }
            else sm_cholesky_substitute(n6, sA, sb, sx);
        }
    }
    SS_STAMP(3);
    __syncthreads();
    SS_STAMP(4);
    if (ld_state(&st->solver_failed) != 0) return;
    for (int t = tid; t < n6; t += 512) g.x[t] = sx[t];
    const int sel = st->sel;
    for (int a = tid; a < g.Npf; a += 512) {
        const int ip = g.free_pose[a];
        double dx[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dx[q] = sx[6 * a + q];
        pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
    }
    SS_STAMP(5);
}

#ifdef VISFS_BA_STAMPS
#define SM_STAMP(slot) do { if (tid == 0 && units <= 3) g.stamps[(units - 1) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define SM_STAMP(slot) do { } while (0)
#endif
template <class Src>
__global__ __launch_bounds__(SM_T) void k_small_optimize(const Src src, const int solver, const int half) {
    const DeviceGraph& g = graph_of(src);
    const LinBuf L = lin_of(g, 0);            // one workgroup, stages in order: no second linearisation set needed
    LmState* st = state_of(src, g);
    __shared__ double sRt[SM_MAX_POSES * 12];              // R|t of the estimate
    __shared__ double sRtT[SM_MAX_POSES * 12];             // ... of the trial state
    __shared__ double sA[SM_MAX_N6 * SM_LD];               // dense reduced camera matrix / its Cholesky factor
    __shared__ double sb[SM_MAX_N6], sx[SM_MAX_N6], sd[SM_MAX_N6];
    __shared__ double sPart[SM_MAX_WCHUNKS * 27];          // per-wave partials of the pose-major pass, then of the laser sums
    __shared__ double sRed[2 * SM_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta, ic = g.inv_odo_cov, il = g.inv_laser_cov;
    const int n6 = 6 * g.Npf;
    int units = 0;
    const int max_units = 32 * (half > 0 ? half : 1) + 32;     // every unit ends an iteration or grows lambda: bounded anyway

    for (int phase = 0; phase < 2; ++phase) {
        while (true) {
            __syncthreads();
            const int mode = ld_state(&st->mode);
            if (mode == 0) break;
            if (++units > max_units) { if (tid == 0) { st->status = 8; st->done = 1; st->mode = 0; } break; }
            const int sel = ld_state(&st->sel);
            const double* pose = g.pose[sel];
            const double* pt = g.pt[sel];
            if (mode & MODE_LIN) {
                SM_STAMP(0);
                const bool first = ld_state(&st->phase_iter) == 0;
                stage_poses(pose, g.Np, sRt);
                __syncthreads();
                double chi_acc = 0.0, md = 0.0;
                for (int l = tid; l < g.Nl; l += SM_T) lin_landmark<1>(g, L, l, true, 0, PoseSrc<true>{ sRt }, pt, K, iv, delta, chi_acc, md);
                SM_STAMP(1);
                // pose-major pass: one wave per 64 observations of a LIN_CHUNK; the four partials of a chunk are added in
                // role B's order, so hpp_part comes out bit-identical to k_linearize's
                for (int wc = wave; wc < 4 * g.n_chunks; wc += SM_WAVES) {
                    const int c = wc >> 2;
                    const int idx = g.chunk_ptr[c] + 64 * (wc & 3) + lane, end = g.chunk_ptr[c + 1];
                    double acc[27];
#pragma unroll
                    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
                    if (idx < end) pose_obs_terms(g, g.pose_obs[idx], load_Rt(sRt, g.free_pose[g.chunk_pose[c]]), pt, K, iv, delta, acc);
                    int off = 0, len = 27;
                    ReduceScatter<27, 32>::run(acc, lane, off, len);
                    if (len >= 1) sPart[wc * 27 + off] = acc[0];
                }
                for (int e_ = tid; e_ < g.Ne; e_ += SM_T) chi_acc += odo_edge_blocks(g, L, e_, pose, ic);
                __syncthreads();
                SM_STAMP(2);
                for (int t = tid; t < 27 * g.n_chunks; t += SM_T) {
                    const int c = t / 27, q = t - 27 * c;
                    const double* pp = sPart + (4 * c) * 27 + q;
                    L.hpp_part[t] = ((pp[0] + pp[27]) + pp[54]) + pp[81];
                }
                if (g.Nz > 0) {                            // laser edges: all on one pose, reduced into the pseudo-edge slot
                    __syncthreads();                       // sPart is reused
                    const double* tq = pose + POSE_STRIDE * g.laser_pose;
                    double acc[27];
#pragma unroll
                    for (int q = 0; q < 27; ++q) acc[q] = 0.0;
                    for (int z = tid; z < g.Nz; z += SM_T) chi_acc += laser_point_terms(g, tq, z, il, acc);
                    int off = 0, len = 27;
                    ReduceScatter<27, 32>::run(acc, lane, off, len);
                    if (len >= 1) sPart[wave * 27 + off] = acc[0];
                    __syncthreads();
                    if (tid < 27) {
                        double v = 0.0;
#pragma unroll
                        for (int w = 0; w < SM_WAVES; ++w) v += sPart[w * 27 + tid];
                        laser_store_slot(g, L, tid, v);
                    }
                }
                __syncthreads();
                if (first) {                               // computeLambdaInit needs max|diag H| and the chi2 of the linearisation
                    for (int t = tid; t < g.Npf * 42; t += SM_T) {
                        const int a = t / 42, q = t % 42;
                        const double v = hpp_entry(g, L, a, q);
                        if (q < 36) { g.Hpp[36 * (size_t)a + q] = v; if (q % 7 == 0) md = fmax(md, fabs(v)); }
                        else g.bp[6 * (size_t)a + (q - 36)] = v;
                    }
                    double zero = 0.0;
                    sm_block_sum2(chi_acc, zero, sRed);
                    const double md_total = sm_block_max(md, sRed);
                    if (tid == 0) lin_finalize_update(st, chi_acc, md_total);
                }
            }
            __syncthreads();
            const double lambda = st->lambda;
            SM_STAMP(3);
            // ---- one damped solve
            for (int ch = wave; ch < g.n_sch; ch += SM_WAVES) schur_chunk<false>(g, L, ch, lane, lambda, pose);
            __syncthreads();
            SM_STAMP(4);
            for (int b = wave; b < g.n_blk; b += SM_WAVES) schur_block(g, L, st, b, lane);
            for (int t = tid; t < n6 * SM_LD; t += SM_T) sA[t] = 0.0;
            __syncthreads();
            SM_STAMP(5);
            for (int t = tid; t < g.n_blk * 36; t += SM_T) {
                const int b = t / 36, q = t - 36 * b, r = q / 6, c = q - 6 * r;
                const int i = g.blk_i[b], j = g.blk_j[b];
                const double v = g.S[t];
                sA[(6 * i + r) * SM_LD + 6 * j + c] = v;
                if (i != j) sA[(6 * j + c) * SM_LD + 6 * i + r] = v;
            }
            for (int t = tid; t < n6; t += SM_T) sb[t] = g.bs[t];
            __syncthreads();
            SM_STAMP(6);
            if (wave == 0) sm_solve(g, st, solver, n6, sA, sb, sd, sx);
            __syncthreads();
            SM_STAMP(7);
            const bool ok = ld_state(&st->solver_failed) == 0;
            double chi_t = 0.0, sc = 0.0;
            if (ok) {
                double* pose_t = g.pose[sel ^ 1];
                double* pt_t = g.pt[sel ^ 1];
                for (int t = tid; t < n6; t += SM_T) g.x[t] = sx[t];
                for (int a = tid; a < g.Npf; a += SM_T) {
                    const int ip = g.free_pose[a];
                    double dx[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) dx[q] = sx[6 * a + q];
                    pose_oplus(pose + POSE_STRIDE * ip, dx, pose_t + POSE_STRIDE * ip);
                }
                __syncthreads();
                stage_poses(pose_t, g.Np, sRtT);
                __syncthreads();
                SM_STAMP(8);
                for (int l = tid; l < g.Nl; l += SM_T) backsub_landmark<1>(g, L, l, true, 0, PoseSrc<true>{ sRtT }, PoseSrc<true>{ sRt }, pt, pt_t, lambda, K, iv, delta, chi_t, sc);
                SM_STAMP(9);
                for (int e_ = tid; e_ < g.Ne; e_ += SM_T) {
                    const int i = g.odo_i[e_], j = g.odo_j[e_];
                    if (g.pose_free[i] < 0 && g.pose_free[j] < 0) continue;
                    double e[6];
                    odo_error(pose_t + POSE_STRIDE * i, pose_t + POSE_STRIDE * j, g.odo_tq + 7 * e_, e);
#pragma unroll
                    for (int d = 0; d < 6; ++d) chi_t += e[d] * (ic * e[d]);
                }
                for (int z = tid; z < g.Nz; z += SM_T) {
                    const double e = laser_error(pose_t + POSE_STRIDE * g.laser_pose, g.Tcr, Vec3{ g.laser_xyz[3 * z], g.laser_xyz[3 * z + 1], g.laser_xyz[3 * z + 2] }, g.grid);
                    chi_t += e * (il * e);
                }
                for (int t = tid; t < n6; t += SM_T) { const double x = sx[t]; sc += x * (lambda * x + g.bp[t]); }
                sm_block_sum2(chi_t, sc, sRed);
                SM_STAMP(10);
            }
            if (tid == 0) lm_decide_call(st, ok, lambda, chi_t, sc);
            SM_STAMP(11);
        }
        // ---- close the phase: per-edge chi2 at the estimate, outlier marking after phase 1 (Optimizer.cpp:270-303, :315-318)
        __syncthreads();
        if (ld_state(&st->status) == 0) {
            const int sel = ld_state(&st->sel);
            const double* pose = g.pose[sel];
            const double* pt = g.pt[sel];
            const int mark = (phase == 0);
            stage_poses(pose, g.Np, sRt);
            __syncthreads();
            double chi = 0.0, nout = 0.0;
            for (int k = tid; k < g.No; k += SM_T) {
                const bool active = (g.obs_level[k] == 0) && g.obs_ok[k];
                double c2 = 0.0;
                if (active) {
                    const int l = g.obs_pt[k];
                    Vec3 pc;
                    const Vec3 e = stereo_error(load_Rt(sRt, g.obs_pose[k]), Vec3{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] }, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
                    c2 = chi2_of(e, iv);
                    double rho0 = c2, rho1 = 1.0;
                    robustify(g, c2, delta, rho0, rho1);
                    chi += rho0;
                    if (mark && delta > 0.0 && c2 > delta) { g.obs_level[k] = 1; g.obs_outlier[k] = 1; nout += 1.0; }
                }
                if (mark) g.obs_chi2_out[k] = c2;
            }
            for (int e_ = tid; e_ < g.Ne; e_ += SM_T) {
                const int i = g.odo_i[e_], j = g.odo_j[e_];
                if (g.pose_free[i] < 0 && g.pose_free[j] < 0) continue;
                double e[6];
                odo_error(pose + POSE_STRIDE * i, pose + POSE_STRIDE * j, g.odo_tq + 7 * e_, e);
#pragma unroll
                for (int d = 0; d < 6; ++d) chi += e[d] * (ic * e[d]);
            }
            for (int z = tid; z < g.Nz; z += SM_T) {
                const double e = laser_error(pose + POSE_STRIDE * g.laser_pose, g.Tcr, Vec3{ g.laser_xyz[3 * z], g.laser_xyz[3 * z + 1], g.laser_xyz[3 * z + 2] }, g.grid);
                chi += e * (il * e);
            }
            sm_block_sum2(chi, nout, sRed);
            if (tid == 0) phase_end_update(g, st, chi, nout, phase, phase == 0 ? half : 0);
        }
        __syncthreads();
    }
}

// ================================================================= launchers
// Every launcher exists once, templated on the graph source: One{g} for a single window (grid.y = 1), Many{gs} for a batch
// of independent windows (grid.y = number of windows, grid.x sized for the largest one; smaller windows' surplus
// workgroups return at once).
// Measurement: an armed event pair is attached to the NEXT timed launch itself (hipExtLaunchKernelGGL: the events take the
// dispatch's own start / end timestamps, which is what rocprofv3 reports), instead of being recorded around it on the stream.
static thread_local hipEvent_t tl_ev_start = nullptr, tl_ev_stop = nullptr;
void arm_launch_events(hipEvent_t a, hipEvent_t b) { tl_ev_start = a; tl_ev_stop = b; }
bool launch_events_pending() { return tl_ev_start != nullptr; }
#define TIMED_LAUNCH(kern, grid, block, lds, stream, ...)                                                                    \
    do {                                                                                                                     \
        if (tl_ev_start) {                                                                                                   \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, tl_ev_start, tl_ev_stop, 0, __VA_ARGS__);                  \
            tl_ev_start = tl_ev_stop = nullptr;                                                                              \
        } else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                              \
    } while (0)

LaunchDims dims_of(const DeviceGraph& g) {
    LaunchDims d;
    d.group = g.group;
    d.np = g.Np;
    d.lin_blocks = g.n_lin_a + g.n_chunks;
    d.chunks = g.n_chunks;
    d.backsub_blocks = g.n_lin_a + 1;
    d.sch_wgs = g.n_sch > 0 ? (((g.n_sch + 3) / 4) + 7) / 8 * 8 : 0;
    d.run_wgs = g.n_runs > 0 ? (g.n_runs + 7) / 8 * 8 : 0;
    d.run_lds = g.n_runs > 0 ? g.run_lds_bytes : 0;
    d.sch_multi = g.sch_chunk > 64 ? 1 : 0;
    d.fin_wgs = g.n_runs > 0 ? g.n_blk : (g.n_blk + 3) / 4;
    d.pcg_rows = g.Npf;
    d.pcg_lds = g.pcg_lds_bytes;
    d.eval_blocks = (g.No + 255) / 256 + 1;
    d.reset_blocks = std::min(std::max((g.No + 255) / 256, 1), 1024);
    d.has_odo = (g.Ne > 0 || g.Nz > 0) ? 1 : 0;
    d.pcg_one_wave = g.pcg1_code != nullptr ? 1 : 0;
    d.fin_pcg = (g.fin_pcg && g.pcg1_code != nullptr) ? 1 : 0;
    d.fin_pcg_wgs = g.n_blk;
    d.pcg_cu = g.pcg_cu;
    d.band = g.band_B >= 0 ? 1 : 0;                    // direct solver: every window of a launch on the banded factorisation (k_band_chol)
    d.band_lds = g.band_B >= 0 ? g.band_lds_bytes : 0;
    d.ceres = g.ceres;
    d.dogleg = g.dogleg;
    return d;
}
LaunchDims dims_max(const LaunchDims& a, const LaunchDims& b) {
    LaunchDims d = a;
    d.np = std::max(a.np, b.np); d.lin_blocks = std::max(a.lin_blocks, b.lin_blocks); d.chunks = std::max(a.chunks, b.chunks); d.backsub_blocks = std::max(a.backsub_blocks, b.backsub_blocks);
    d.run_wgs = std::max(a.run_wgs, b.run_wgs); d.run_lds = std::max(a.run_lds, b.run_lds);   // (a launch serves run-path windows or gather windows, never both)
    d.sch_wgs = std::max(a.sch_wgs, b.sch_wgs); d.fin_wgs = std::max(a.fin_wgs, b.fin_wgs); d.pcg_rows = std::max(a.pcg_rows, b.pcg_rows);
    d.pcg_lds = std::max(a.pcg_lds, b.pcg_lds); d.eval_blocks = std::max(a.eval_blocks, b.eval_blocks); d.reset_blocks = std::max(a.reset_blocks, b.reset_blocks);
    d.has_odo = a.has_odo | b.has_odo; d.sch_multi = a.sch_multi | b.sch_multi;
    d.pcg_one_wave = a.pcg_one_wave & b.pcg_one_wave;
    d.fin_pcg = a.fin_pcg & b.fin_pcg; d.fin_pcg_wgs = std::max(a.fin_pcg_wgs, b.fin_pcg_wgs);
    d.pcg_cu = a.pcg_cu & b.pcg_cu;
    d.band = a.band & b.band; d.band_lds = std::max(a.band_lds, b.band_lds);
    d.ceres = a.ceres & b.ceres;                       // (a handle has one framework: all windows of a launch agree)
    d.dogleg = a.dogleg & b.dogleg;                    // ... and one trust-region strategy
    return d;
}
// Launch geometry rounded up to SIZE CLASSES (steps of at most 12.5 %): a launch sequence captured for one window of a sliding map then
// serves the next frames too — their windows differ by a few observations, and every kernel that reads its graph from HBM (Many) lets the
// surplus workgroups of a grid sized for a larger window return at once (the batched launches rely on the same property).
static inline int size_class(const int x) {
    if (x <= 8) return x;
    const int e = 31 - __builtin_clz((unsigned)x), step = 1 << (e - 3);
    return (x + step - 1) / step * step;
}
LaunchDims dims_class(const LaunchDims& a) {
    LaunchDims d = a;
    d.np = a.np <= MAX_STAGED_POSES ? std::min(size_class(a.np), MAX_STAGED_POSES) : size_class(a.np);   // (a class never changes the staged / unstaged choice)
    d.chunks = size_class(a.chunks); d.lin_blocks = std::max(size_class(a.lin_blocks), size_class(a.backsub_blocks - 1) + d.chunks);
    d.backsub_blocks = size_class(a.backsub_blocks);
    d.sch_wgs = (size_class(a.sch_wgs) + 7) / 8 * 8; d.run_wgs = (size_class(a.run_wgs) + 7) / 8 * 8; d.run_lds = a.run_lds;
    d.fin_wgs = size_class(a.fin_wgs); d.pcg_rows = size_class(a.pcg_rows); d.fin_pcg_wgs = size_class(a.fin_pcg_wgs);
    if (a.pcg_rows <= 64) d.pcg_rows = std::min(d.pcg_rows, 64);                   // (the k_pcg / k_pcg1 variant follows the row count)
    else if (a.pcg_rows <= MAX_PCG_ONE_ROW_POSES) d.pcg_rows = std::min(d.pcg_rows, MAX_PCG_ONE_ROW_POSES);
    d.pcg_lds = a.pcg_lds > 0 ? std::min(size_class(a.pcg_lds), 160 * 1024) : 0;
    d.eval_blocks = size_class(a.eval_blocks); d.reset_blocks = size_class(a.reset_blocks);
    d.band_lds = a.band_lds > 0 ? std::min((size_class(a.band_lds) + 15) / 16 * 16, (int)BAND_LDS_BUDGET) : 0;
    return d;
}
bool dims_equal(const LaunchDims& a, const LaunchDims& b) { return std::memcmp(&a, &b, sizeof(LaunchDims)) == 0; }

// dynamic LDS of the kernels that stage every pose of the window as R|t (12 doubles each); windows beyond MAX_STAGED_POSES use the
// instantiations that read the poses from HBM instead (STG = false) and need only the reduction scratch
static inline bool staged(const LaunchDims& d) { return d.np <= MAX_STAGED_POSES; }
static inline size_t lds_poses(const LaunchDims& d, int extra) { return (size_t)((staged(d) ? 12 * d.np : 0) + extra) * sizeof(double); }
// more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel (gfx950 has 160 KiB per CU)
template <class K> static inline void ensure_lds(K kern, size_t lds) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int G, class Src>
static void launch_lin_t(const Src& src, const LaunchDims& d, int B, int spec, hipStream_t s) {
    const size_t lds = lds_poses(d, 4 * 27);
    if (!staged(d)) {                                    // (never speculative, never batched: ba_api.cpp keeps such windows on the gated single-window unit)
        TIMED_LAUNCH((k_linearize<G, One, false, false>), dim3(d.lin_blocks, B), dim3(256), lds, s, One{ graph_of_host(src) });
        return;
    }
    if (!Src::batched && spec) { ensure_lds(k_linearize<G, Src, !Src::batched>, lds); TIMED_LAUNCH((k_linearize<G, Src, !Src::batched>), dim3(d.lin_blocks + 1, B), dim3(256), lds, s, src); }
    else { ensure_lds(k_linearize<G, Src, false>, lds); TIMED_LAUNCH((k_linearize<G, Src, false>), dim3(d.lin_blocks, B), dim3(256), lds, s, src); }
}
// dec: the launch carries the LM decision (one workgroup more; k_decide is then not launched)
template <int G, class Src>
static void launch_backsub_t(const Src& src, const LaunchDims& d, int B, int odospec, int dec, hipStream_t s) {
    if (!staged(d)) {
        if (dec) TIMED_LAUNCH((k_backsub<G, One, false, false, true>), dim3(d.backsub_blocks + B, B), dim3(256), (size_t)8 * sizeof(double), s, One{ graph_of_host(src) });
        else TIMED_LAUNCH((k_backsub<G, One, false, false>), dim3(d.backsub_blocks, B), dim3(256), (size_t)8 * sizeof(double), s, One{ graph_of_host(src) });
        return;
    }
    if (!Src::batched && odospec) {
        const size_t lds = (size_t)std::max(24 * d.np + 8, 128) * sizeof(double);
        ensure_lds(k_backsub<G, Src, !Src::batched>, lds);
        TIMED_LAUNCH((k_backsub<G, Src, !Src::batched>), dim3(d.backsub_blocks, B), dim3(256), lds, s, src);
    } else if (dec) {
        const size_t lds = (size_t)(24 * d.np + 8) * sizeof(double);
        ensure_lds(k_backsub<G, Src, false, true, true>, lds);
        TIMED_LAUNCH((k_backsub<G, Src, false, true, true>), dim3(d.backsub_blocks + B, B), dim3(256), lds, s, src);
    } else {
        const size_t lds = (size_t)(24 * d.np + 8) * sizeof(double);
        ensure_lds(k_backsub<G, Src, false>, lds);
        TIMED_LAUNCH((k_backsub<G, Src, false>), dim3(d.backsub_blocks, B), dim3(256), lds, s, src);
    }
}
template <class Src>
static void launch_linearize_src(const Src& src, const LaunchDims& d, int B, int spec, hipStream_t s) {
    // (speculative unit: the odometry / laser edges were linearised by k_backsub<ODOSPEC> already)
    if (d.has_odo && !spec) hipLaunchKernelGGL((k_odo_linearize<Src>), dim3(1, B), dim3(256), 0, s, src, spec);     // lin_part[n_lin_a] stays 0 otherwise
    switch (d.group) {
        case 4: launch_lin_t<4>(src, d, B, spec, s); break;
        case 8: launch_lin_t<8>(src, d, B, spec, s); break;
        case 16: launch_lin_t<16>(src, d, B, spec, s); break;
        case 32: launch_lin_t<32>(src, d, B, spec, s); break;
        default: launch_lin_t<64>(src, d, B, spec, s); break;
    }
}
template <class Src>
static void launch_lin_finalize_src(const Src& src, int force, int B, hipStream_t s) {
    TIMED_LAUNCH((k_lin_finalize<Src>), dim3(1, B), dim3(1024), 0, s, src, force);
}
template <class Src>
static void launch_ceres_lin_finalize_src(const Src& src, int B, hipStream_t s) {
    TIMED_LAUNCH((k_ceres_lin_finalize<Src>), dim3(1, B), dim3(1024), 0, s, src);
}
template <class Src>
static void launch_schur_partial_src(const Src& src, const LaunchDims& d, int B, hipStream_t s) {
    if (d.run_wgs > 0) {                                                   // Schur complement by runs of landmarks
        const size_t lds = (size_t)d.run_lds;
        if (d.ceres) { ensure_lds(k_schur_runs<Src, true>, lds); TIMED_LAUNCH((k_schur_runs<Src, true>), dim3(d.run_wgs, B), dim3(256), lds, s, src); }
        else { ensure_lds(k_schur_runs<Src, false>, lds); TIMED_LAUNCH((k_schur_runs<Src, false>), dim3(d.run_wgs, B), dim3(256), lds, s, src); }
        return;
    }
    if (d.sch_wgs <= 0) return;
    if (d.ceres) {                                                         // Optimizer/Framework=1: the damping is per variable (damp_of)
        if (d.sch_multi) TIMED_LAUNCH((k_schur_partial<true, Src, true>), dim3(d.sch_wgs, B), dim3(256), 0, s, src);
        else TIMED_LAUNCH((k_schur_partial<false, Src, true>), dim3(d.sch_wgs, B), dim3(256), 0, s, src);
        return;
    }
    if (d.sch_multi) TIMED_LAUNCH((k_schur_partial<true, Src>), dim3(d.sch_wgs, B), dim3(256), 0, s, src);
    else TIMED_LAUNCH((k_schur_partial<false, Src>), dim3(d.sch_wgs, B), dim3(256), 0, s, src);
}
template <class Src>
static void launch_schur_finalize_src(const Src& src, const LaunchDims& d, int B, hipStream_t s) {
    if (d.run_wgs > 0) TIMED_LAUNCH((k_schur_finalize<Src, true>), dim3(d.fin_wgs, B), dim3(256), 0, s, src);     // one workgroup per block
    else TIMED_LAUNCH((k_schur_finalize<Src>), dim3(d.fin_wgs, B), dim3(256), 0, s, src);
}
template <class Src>
static void launch_pcg_src(const Src& src, const LaunchDims& d, int B, hipStream_t s) {
    if (d.pcg_cu) { ensure_lds(k_pcg_cu<Src>, (size_t)d.pcg_lds); TIMED_LAUNCH((k_pcg_cu<Src>), dim3(1, B), dim3((6 * d.pcg_rows + 63) / 64 * 64), (size_t)d.pcg_lds, s, src); return; }
    if (d.pcg_one_wave) {
        // default: variant 1 (six 16-byte loads per sweep, one sweep in flight) — 21.4 us per C2 solve and 65.6 k it/s in 16-window
        // batches against 21.2-25.5 / 59.4 k for variant 0 and 20.9 / 61.6 k for variant 2 (profiles/r02_pcg1_gather_variants.log)
        static const int gv = []() { const char* e = std::getenv("VISFS_BA_PCG_GATHER"); return e ? std::atoi(e) : 1; }();
        // the XCD-local form only for a window on its own: its <= 64 one-wave workgroups are resident even if the dispatcher packs
        // them all onto one XCD (that is the intent); a batch on one XCD would not be (co-residency must not depend on placement)
        if (gv == 3 && !Src::batched) TIMED_LAUNCH((k_pcg1<One, 3>), dim3(8 * d.pcg_rows, B), dim3(64), 0, s, One{ graph_of_host(src) });
        else if (gv >= 2) TIMED_LAUNCH((k_pcg1<Src, 2>), dim3(d.pcg_rows, B), dim3(64), 0, s, src);
        else if (gv == 1 && d.fin_pcg) TIMED_LAUNCH((k_pcg1<Src, 1, true>), dim3(d.fin_pcg_wgs, B), dim3(64), 0, s, src);     // finalisation on board
        else if (gv == 1) TIMED_LAUNCH((k_pcg1<Src, 1>), dim3(d.pcg_rows, B), dim3(64), 0, s, src);
        else TIMED_LAUNCH((k_pcg1<Src, 0>), dim3(d.pcg_rows, B), dim3(64), 0, s, src);
    }
    else if (d.pcg_rows <= 64) TIMED_LAUNCH((k_pcg<1, true, Src, false>), dim3(d.pcg_rows, B), dim3(256), (size_t)d.pcg_lds, s, src);
    else if (d.pcg_rows <= MAX_PCG_ONE_ROW_POSES) TIMED_LAUNCH((k_pcg<1, true, Src, true>), dim3(d.pcg_rows, B), dim3(256), (size_t)d.pcg_lds, s, src);
    else {
        // more than 256 free poses (single windows only): several block rows per workgroup, 2 or 4 blocks per owner thread
        const DeviceGraph& g = graph_of_host(src);
        const int wgs = (d.pcg_rows + g.pcg_rows_per_wg - 1) / g.pcg_rows_per_wg;
        if (d.pcg_rows <= 512) { ensure_lds(k_pcg<2, false, One, true, true>, (size_t)d.pcg_lds); TIMED_LAUNCH((k_pcg<2, false, One, true, true>), dim3(wgs, B), dim3(256), (size_t)d.pcg_lds, s, One{ g }); }
        else { ensure_lds(k_pcg<4, false, One, true, true>, (size_t)d.pcg_lds); TIMED_LAUNCH((k_pcg<4, false, One, true, true>), dim3(wgs, B), dim3(256), (size_t)d.pcg_lds, s, One{ g }); }
    }
}
template <class Src>
static void launch_backsub_src(const Src& src, const LaunchDims& d, int B, int odospec, int dec, hipStream_t s) {
    switch (d.group) {
        case 4: launch_backsub_t<4>(src, d, B, odospec, dec, s); break;
        case 8: launch_backsub_t<8>(src, d, B, odospec, dec, s); break;
        case 16: launch_backsub_t<16>(src, d, B, odospec, dec, s); break;
        case 32: launch_backsub_t<32>(src, d, B, odospec, dec, s); break;
        default: launch_backsub_t<64>(src, d, B, odospec, dec, s); break;
    }
}
template <class Src>
static void launch_phase_end_src(const Src& src, const LaunchDims& d, int B, int phase_just_done, int mark, int next_max_iter, hipStream_t s) {
    if (!staged(d)) hipLaunchKernelGGL((k_eval<One, false>), dim3(d.eval_blocks, B), dim3(256), lds_poses(d, 8), s, One{ graph_of_host(src) }, mark, phase_just_done);
    else { ensure_lds(k_eval<Src>, lds_poses(d, 8)); hipLaunchKernelGGL((k_eval<Src>), dim3(d.eval_blocks, B), dim3(256), lds_poses(d, 8), s, src, mark, phase_just_done); }
    hipLaunchKernelGGL((k_phase_end<Src>), dim3(1, B), dim3(256), 0, s, src, phase_just_done, next_max_iter);
}

// ---- single window
void launch_build_index(const DeviceGraph& g, int32_t* hist, hipStream_t s) {
    const int nblocks = (g.No + IDX_T - 1) / IDX_T;
    const size_t lds = (size_t)4 * std::max(g.Npf, 1) * sizeof(int);          // per-wavefront pose counts (<= 16 KB at 1024 free poses)
    ensure_lds(k_index_count, lds); ensure_lds(k_index_scatter, lds);      // (beyond 64 KB: more than 4096 free poses)
    if (nblocks > 0) hipLaunchKernelGGL(k_index_count, dim3(nblocks), dim3(256), lds, s, g, hist);
    hipLaunchKernelGGL(k_index_scan, dim3(std::max(1, g.Npf)), dim3(256), 0, s, g, hist, nblocks);
    if (nblocks > 0) hipLaunchKernelGGL(k_index_scatter, dim3(nblocks), dim3(256), lds, s, g, hist);
}
int index_blocks(int No) { return (No + IDX_T - 1) / IDX_T; }
void launch_build_pairs(const DeviceGraph& g, hipStream_t s) {
    if (g.n_blk <= 0 || g.n_runs > 0) return;             // (the run-based Schur kernel needs no pair lists)
    hipLaunchKernelGGL(k_build_pairs, dim3((g.n_blk + 3) / 4), dim3(256), 0, s, g);
}
void launch_linearize(const DeviceGraph& g, hipStream_t s) { launch_linearize_src(One{ g }, dims_of(g), 1, 0, s); }
void launch_linearize_decide(const DeviceGraph& g, hipStream_t s) { launch_linearize_src(One{ g }, dims_of(g), 1, 1, s); }
void launch_lin_finalize(const DeviceGraph& g, int force, hipStream_t s) { launch_lin_finalize_src(One{ g }, force, 1, s); }
void launch_ceres_lin_finalize(const DeviceGraph& g, hipStream_t s) { launch_ceres_lin_finalize_src(One{ g }, 1, s); }
void launch_schur_partial(const DeviceGraph& g, hipStream_t s) { launch_schur_partial_src(One{ g }, dims_of(g), 1, s); }
void launch_schur_finalize(const DeviceGraph& g, hipStream_t s) { launch_schur_finalize_src(One{ g }, dims_of(g), 1, s); }
void launch_pcg(const DeviceGraph& g, hipStream_t s) { launch_pcg_src(One{ g }, dims_of(g), 1, s); }
void launch_backsub(const DeviceGraph& g, hipStream_t s) { launch_backsub_src(One{ g }, dims_of(g), 1, 0, 0, s); }
void launch_backsub_odospec(const DeviceGraph& g, hipStream_t s) { launch_backsub_src(One{ g }, dims_of(g), 1, 1, 0, s); }
void launch_backsub_decide(const DeviceGraph& g, hipStream_t s) { launch_backsub_src(One{ g }, dims_of(g), 1, 0, 1, s); }
void launch_decide(const DeviceGraph& g, hipStream_t s) { TIMED_LAUNCH((k_decide<One>), dim3(1), dim3(256), 0, s, One{ g }); }
// The fused speculative unit (single window, staged poses): its last launch — back-substitution, trial chi2, the LM decision and role A
// of the trial's linearisation — and its first — the Schur gather with the pending role-B workgroups behind it.
template <int G, class Src>
static void launch_backsub_lin_t(const Src& src, const LaunchDims& d, int B, hipStream_t s) {
    const size_t lds = (size_t)std::max(24 * d.np + 8, 128) * sizeof(double);
    // grid: the landmark workgroups + the odometry / laser workgroup of the largest window, then one decider per window in the last grid row
    if (d.has_odo) { ensure_lds(k_backsub<G, Src, true, true, true, 0, true>, lds); TIMED_LAUNCH((k_backsub<G, Src, true, true, true, 0, true>), dim3(d.backsub_blocks + B, B), dim3(256), lds, s, src); }
    else { ensure_lds(k_backsub<G, Src, false, true, true, 0, true>, lds); TIMED_LAUNCH((k_backsub<G, Src, false, true, true, 0, true>), dim3(d.backsub_blocks + B, B), dim3(256), lds, s, src); }
}
template <class Src>
static void launch_backsub_lin_src(const Src& src, const LaunchDims& d, int B, hipStream_t s) {
    switch (d.group) {
        case 4: launch_backsub_lin_t<4>(src, d, B, s); break;
        case 8: launch_backsub_lin_t<8>(src, d, B, s); break;
        case 16: launch_backsub_lin_t<16>(src, d, B, s); break;
        case 32: launch_backsub_lin_t<32>(src, d, B, s); break;
        default: launch_backsub_lin_t<64>(src, d, B, s); break;
    }
}
template <class Src>
static void launch_schur_partial_roleb_src(const Src& src, const LaunchDims& d, int B, hipStream_t s) {
    if (d.run_wgs > 0) {
        const size_t lds = (size_t)d.run_lds;
        ensure_lds(k_schur_runs<Src, false, true>, lds);
        TIMED_LAUNCH((k_schur_runs<Src, false, true>), dim3(d.run_wgs + d.chunks, B), dim3(256), lds, s, src);
        return;
    }
    const int grid = d.sch_wgs + d.chunks;                    // (a window's role-B workgroups start right behind ITS share of the chunk list)
    if (grid <= 0) return;
    if (d.sch_multi) TIMED_LAUNCH((k_schur_partial<true, Src, false, true>), dim3(grid, B), dim3(256), 0, s, src);
    else TIMED_LAUNCH((k_schur_partial<false, Src, false, true>), dim3(grid, B), dim3(256), 0, s, src);
}
void launch_backsub_lin_decide(const DeviceGraph& g, hipStream_t s) { launch_backsub_lin_src(One{ g }, dims_of(g), 1, s); }
void launch_schur_partial_roleb(const DeviceGraph& g, hipStream_t s) { launch_schur_partial_roleb_src(One{ g }, dims_of(g), 1, s); }
// Optimizer/Framework=1 with the DOGLEG strategy: pass 1 (Gauss-Newton landmark step + the inner products), the combination, pass 2 (trial state)
template <int G>
static void launch_backsub_dogleg_t(const DeviceGraph& g, const LaunchDims& d, int pass, hipStream_t s) {
    const One src{ g };
    if (staged(d)) {
        const size_t lds = lds_poses(d, 8 + 12 * d.np);
        if (pass == 1) { ensure_lds(k_backsub<G, One, false, true, false, 1>, lds); hipLaunchKernelGGL((k_backsub<G, One, false, true, false, 1>), dim3(d.backsub_blocks), dim3(256), lds, s, src); }
        else { ensure_lds(k_backsub<G, One, false, true, false, 2>, lds); hipLaunchKernelGGL((k_backsub<G, One, false, true, false, 2>), dim3(d.backsub_blocks), dim3(256), lds, s, src); }
    } else {
        if (pass == 1) hipLaunchKernelGGL((k_backsub<G, One, false, false, false, 1>), dim3(d.backsub_blocks), dim3(256), (size_t)8 * sizeof(double), s, src);
        else hipLaunchKernelGGL((k_backsub<G, One, false, false, false, 2>), dim3(d.backsub_blocks), dim3(256), (size_t)8 * sizeof(double), s, src);
    }
}
void launch_backsub_dogleg(const DeviceGraph& g, int pass, hipStream_t s) {
    const LaunchDims d = dims_of(g);
    switch (d.group) {
        case 4: launch_backsub_dogleg_t<4>(g, d, pass, s); break;
        case 8: launch_backsub_dogleg_t<8>(g, d, pass, s); break;
        case 16: launch_backsub_dogleg_t<16>(g, d, pass, s); break;
        case 32: launch_backsub_dogleg_t<32>(g, d, pass, s); break;
        default: launch_backsub_dogleg_t<64>(g, d, pass, s); break;
    }
}
void launch_dogleg_mid(const DeviceGraph& g, hipStream_t s) { hipLaunchKernelGGL((k_dogleg_mid<One>), dim3(1), dim3(256), 0, s, One{ g }); }
void launch_phase_end(const DeviceGraph& g, int phase_just_done, int mark, int next_max_iter, hipStream_t s) {
    launch_phase_end_src(One{ g }, dims_of(g), 1, phase_just_done, mark, next_max_iter, s);
}
// stage hook: the outlier pass of Optimizer.cpp:283-303 on the committed estimate, ungated (k_eval alone: marks edges, no phase change)
void launch_eval_mark(const DeviceGraph& g, hipStream_t s) {
    const LaunchDims d = dims_of(g);
    if (!staged(d)) hipLaunchKernelGGL((k_eval<One, false>), dim3(d.eval_blocks), dim3(256), lds_poses(d, 8), s, One{ g }, 1, -1);
    else { ensure_lds(k_eval<One>, lds_poses(d, 8)); hipLaunchKernelGGL((k_eval<One>), dim3(d.eval_blocks), dim3(256), lds_poses(d, 8), s, One{ g }, 1, -1); }
}
void launch_reset(const DeviceGraph& g, int max_iter, int gauss_newton, int restore, hipStream_t s) {
    hipLaunchKernelGGL((k_reset<One>), dim3(dims_of(g).reset_blocks), dim3(256), 0, s, One{ g }, max_iter, gauss_newton, restore);
}
void launch_small_solve(const DeviceGraph& g, int solver, hipStream_t s) {
    static const int pcg_lds = []() { const char* e = std::getenv("VISFS_BA_SMALL_PCG_LDS"); return (e && e[0] == '1') ? 0x100 : 0; }();
    TIMED_LAUNCH((k_small_solve<One>), dim3(1), dim3(512), 0, s, One{ g }, solver | (solver == 2 ? pcg_lds : 0));
}
void launch_small_optimize(const DeviceGraph& g, int solver, int half, hipStream_t s) {
    hipLaunchKernelGGL((k_small_optimize<One>), dim3(1), dim3(SM_T), 0, s, One{ g }, solver, half);
}

void launch_direct(const DeviceGraph& g, hipStream_t s) {
    if (g.band_B >= 0) {                                                      // block-banded S: one workgroup, one launch
        ensure_lds(k_band_chol<One>, (size_t)g.band_lds_bytes);
        TIMED_LAUNCH((k_band_chol<One>), dim3(1), dim3(BAND_T), (size_t)g.band_lds_bytes, s, One{ g });
        return;
    }
    const int NP = g.chol_np;
    int grid = (g.n_blk * 36 + 255) / 256;
    if (grid > 1024) grid = 1024;
    (void)hipMemsetAsync(g.dense, 0, (size_t)NP * NP * sizeof(double), s);     // the factor fills in: clear the scratch every solve
    hipLaunchKernelGGL(k_dense_assemble, dim3(grid), dim3(256), 0, s, g);
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(64), 0, s, g, 0);
    for (int kb = 0; kb + CH_NB < NP; kb += CH_NB) {                          // one launch per panel: L21, trailing update, next L11
        const int T = (NP - kb - CH_NB + 63) / 64;
        hipLaunchKernelGGL(k_chol_update, dim3(T, T), dim3(256), 0, s, g, kb);
    }
    hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(1024), 0, s, g);
}

void launch_stage_arm(const DeviceGraph& g, double lambda, int mode, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_arm, dim3(1), dim3(1), 0, s, g, lambda, mode);
}

// ---- a batch of independent windows (gs: B DeviceGraphs in HBM; d: element-wise maximum of their launch geometry)
void launch_reset_batch(const DeviceGraph* gs, int B, const LaunchDims& d, int max_iter, int gauss_newton, int restore, hipStream_t s) {
    hipLaunchKernelGGL((k_reset<Many>), dim3(d.reset_blocks, B), dim3(256), 0, s, Many{ gs }, max_iter, gauss_newton, restore);
}
// One unit of the LM state machine for every window of the batch (PCG or, for reduced systems <= 64 x 64, k_small_solve).
template <int G>
static void launch_backsub_dogleg_many_t(const Many& src, const LaunchDims& d, int B, int pass, hipStream_t s) {
    const size_t lds = lds_poses(d, 8 + 12 * d.np);
    if (pass == 1) { ensure_lds(k_backsub<G, Many, false, true, false, 1>, lds); hipLaunchKernelGGL((k_backsub<G, Many, false, true, false, 1>), dim3(d.backsub_blocks, B), dim3(256), lds, s, src); }
    else { ensure_lds(k_backsub<G, Many, false, true, false, 2>, lds); hipLaunchKernelGGL((k_backsub<G, Many, false, true, false, 2>), dim3(d.backsub_blocks, B), dim3(256), lds, s, src); }
}
static void launch_backsub_dogleg_many(const Many& src, const LaunchDims& d, int B, int pass, hipStream_t s) {
    switch (d.group) {
        case 4: launch_backsub_dogleg_many_t<4>(src, d, B, pass, s); break;
        case 8: launch_backsub_dogleg_many_t<8>(src, d, B, pass, s); break;
        case 16: launch_backsub_dogleg_many_t<16>(src, d, B, pass, s); break;
        case 32: launch_backsub_dogleg_many_t<32>(src, d, B, pass, s); break;
        default: launch_backsub_dogleg_many_t<64>(src, d, B, pass, s); break;
    }
}
void launch_unit_batch(const DeviceGraph* gs, int B, const LaunchDims& d, bool first, bool small_solve, int solver, bool fused_decide, bool spec_fused, hipStream_t s, LmState* st1) {
    const Many src{ gs, st1 };
    // spec_fused (round 4): the members run the fused speculative unit of a window on its own — k_linearize only in the first unit of a
    // phase, the Schur gather with the pending pose-major role behind it, the back-substitution with the LM decision and the landmark-major
    // role of the trial's linearisation on board: 4 launches per unit instead of 6 (3 instead of 5 with k_small_solve)
    if (!spec_fused || first) launch_linearize_src(src, d, B, 0, s);
    if (d.ceres) launch_ceres_lin_finalize_src(src, B, s);   // Optimizer/Framework=1: cost, gradient test, Jacobi scaling after EVERY linearisation
    else if (first) launch_lin_finalize_src(src, 0, B, s);
    if (spec_fused) launch_schur_partial_roleb_src(src, d, B, s);
    else launch_schur_partial_src(src, d, B, s);
    if (small_solve) hipLaunchKernelGGL((k_small_solve<Many>), dim3(1, B), dim3(512), 0, s, src, solver);
    else if (solver != 2) {
        // the reference-default linear solver in batched launches: one workgroup per window factors its banded S (callers only group
        // windows whose band qualifies: LaunchDims::band)
        launch_schur_finalize_src(src, d, B, s);
        ensure_lds(k_band_chol<Many>, (size_t)d.band_lds);
        hipLaunchKernelGGL((k_band_chol<Many>), dim3(1, B), dim3(BAND_T), (size_t)d.band_lds, s, src);
    }
    else { if (!(d.fin_pcg && d.pcg_one_wave && !d.pcg_cu)) launch_schur_finalize_src(src, d, B, s); launch_pcg_src(src, d, B, s); }
    if (d.dogleg) {
        // Optimizer/Framework=1 with the DOGLEG strategy: the two back-substitution passes around the point on the dogleg path, then the decision
        launch_backsub_dogleg_many(src, d, B, 1, s);
        hipLaunchKernelGGL((k_dogleg_mid<Many>), dim3(1, B), dim3(256), 0, s, src);
        launch_backsub_dogleg_many(src, d, B, 2, s);
        hipLaunchKernelGGL((k_decide<Many>), dim3(1, B), dim3(256), 0, s, src);
        return;
    }
    if (spec_fused) { launch_backsub_lin_src(src, d, B, s); return; }
    // the LM decision rides on k_backsub (fused_decide = false: one k_decide launch for all windows, as in round 1)
    launch_backsub_src(src, d, B, 0, fused_decide ? 1 : 0, s);
    if (!fused_decide) hipLaunchKernelGGL((k_decide<Many>), dim3(1, B), dim3(256), 0, s, src);
}
void launch_phase_end_batch(const DeviceGraph* gs, int B, const LaunchDims& d, int phase_just_done, int mark, int next_max_iter, hipStream_t s, LmState* st1) {
    launch_phase_end_src(Many{ gs, st1 }, d, B, phase_just_done, mark, next_max_iter, s);
}
void launch_small_optimize_batch(const DeviceGraph* gs, int B, int solver, int half, hipStream_t s) {
    hipLaunchKernelGGL((k_small_optimize<Many>), dim3(1, B), dim3(SM_T), 0, s, Many{ gs }, solver, half);
}
// Every window's whole LmState, contiguous (one D2H copy instead of one per window).
__global__ void k_gather_lm(const DeviceGraph* gs, int B, LmState* out) {
    const int b = blockIdx.x;
    if (b >= B) return;
    const unsigned* src = reinterpret_cast<const unsigned*>(gs[b].st);
    unsigned* dst = reinterpret_cast<unsigned*>(out + b);
    for (int t = threadIdx.x; t < (int)(sizeof(LmState) / 4); t += blockDim.x) dst[t] = src[t];
}
void launch_gather_lm(const DeviceGraph* gs, int B, LmState* out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_lm, dim3(B), dim3(64), 0, s, gs, B, out);
}

// How many workgroups of the persistent PCG kernel this launch geometry would use can be RESIDENT on the device at once
// (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units, capped at the hardware's 8 workgroup slots per CU for these
// small workgroups).  The hand-off of k_pcg / k_pcg1 only terminates when every workgroup of a window's grid is resident, so a
// batched launch must never carry more block rows than this (cdna_hip_programming.md §1: residency comes from the grid size alone).
int pcg_resident_capacity(const LaunchDims& d, bool many, int device) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 0;
    int per_cu = 0;
    hipError_t e;
    if (d.pcg_one_wave) e = many ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg1<Many, 0>, 64, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg1<One, 0>, 64, 0);
    else if (d.pcg_rows <= 64) e = many ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg<1, true, Many, false>, 256, (size_t)d.pcg_lds)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg<1, true, One, false>, 256, (size_t)d.pcg_lds);
    else e = many ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg<1, true, Many, true>, 256, (size_t)d.pcg_lds)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg<1, true, One, true>, 256, (size_t)d.pcg_lds);
    if (e != hipSuccess || per_cu <= 0) return 0;
    // the API can over-report by one workgroup per CU for SGPR-heavy kernels (MI355X_MICROARCH.md, residency): keep one slot
    // of margin wherever more than one is reported, and never count more than the 8 slots a CU has for such workgroups
    per_cu = std::min(per_cu, 8);
    if (per_cu > 1) per_cu -= 1;
    return per_cu * cus;
}

bool small_path_fits(const DeviceGraph& g) {
    return g.Np <= SM_MAX_POSES && 6 * g.Npf <= SM_MAX_N6 && g.Npf >= 1 && 4 * g.n_chunks <= SM_MAX_WCHUNKS && g.No <= SM_MAX_OBS && g.n_sch <= SM_MAX_SCH && g.sch_chunk == SCH_CHUNK;
}

bool pcg_cu_fits(int npf, int max_row) { return npf >= 1 && 6 * npf <= CU_MAX_N6 && max_row <= CU_K && pcg_cu_lds_bytes(npf, max_row) <= 140 * 1024; }

bool small_solve_fits(const DeviceGraph& g) { return g.Npf >= 1 && 6 * g.Npf <= SM_MAX_N6; }

int configure_kernels(const DeviceGraph& g) {
    // dynamic LDS above 64 KiB needs an explicit opt-in (the multi-row PCG kernels opt in at their launch: ensure_lds)
    if (g.pcg_lds_bytes > 64 * 1024 && g.Npf <= MAX_PCG_ONE_ROW_POSES) {
        const void* f = g.Npf <= 64 ? reinterpret_cast<const void*>(k_pcg<1, true, One, false>) : reinterpret_cast<const void*>(k_pcg<1, true, One, true>);
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, g.pcg_lds_bytes) != hipSuccess) return -1;
    }
    return 0;
}

}  // namespace visfs_ba
