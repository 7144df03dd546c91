// ba_kernels.hip — hand-written gfx950 kernels of the sliding-window BA hot path.
//
// One "unit" of the device-side LM state machine is the launch sequence
//   k_linearize → k_lin_finalize → k_schur → k_pcg_init → k_pcg_iter × P → k_backsub → k_decide
// Each kernel reads its gate from LmState in HBM and returns immediately when it has nothing
// to do, so a whole phase is enqueued without any host round trip.
//
// Mapping to the reference / g2o:
//   K1,K2  EdgeStereo::computeError / linearizeOplus          → k_linearize (role A, B), k_backsub, k_eval
//   K3     EdgePoseConstraint                                 → k_linearize (role C), k_backsub (odometry role)
//   K4     constructQuadraticForm + RobustKernelHuber          → k_linearize + k_lin_finalize
//   K5     BlockSolver Schur complement                        → k_schur (gather over per-block pair lists)
//   K6     LinearSolverPCG / direct Cholesky                   → k_pcg_init + k_pcg_iter / k_dense_assemble + k_cholesky
//   K7,K8  back-substitution, oplus                            → k_backsub (+ pose update in the solver epilogue)
//   K9     OptimizationAlgorithmLevenberg control              → k_lin_finalize + k_decide
//   K10    outlier marking (Optimizer.cpp:283-303)             → k_eval + k_phase_end
//
// All reductions are fixed-order (wave butterflies + serial tails): no floating-point atomics,
// bitwise reproducible results.  Wavefront = 64 everywhere.
#include "ba_kernels.hpp"

#include <float.h>

namespace visfs_ba {

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
template <int G>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}

// Sum (or max) one value per thread over a 256-thread workgroup; result valid in thread 0.
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

__device__ __forceinline__ Intrinsics intr_of(const DeviceGraph& g) { return Intrinsics{ g.fx, g.fy, g.cx, g.cy, g.bf }; }

// chi2() = e . (Omega e), Omega = I3 / pixelVariance (Optimizer.cpp:153)
__device__ __forceinline__ double chi2_of(const Vec3& e, double iv) { return e.x * (iv * e.x) + e.y * (iv * e.y) + e.z * (iv * e.z); }

// ================================================================= K1/K2/K3/K4: linearise
template <int G>
__global__ __launch_bounds__(256) void k_linearize(const DeviceGraph g) {
    const LmState* st = g.st;
    if (st->done || !st->need_lin) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sRt = smem;                       // [Np][12]
    double* red = smem + 12 * g.Np;           // [4 * 27]
    const int sel = st->sel;
    const double* __restrict__ pose = g.pose[sel];
    const double* __restrict__ pt = g.pt[sel];
    stage_poses(pose, g.Np, sRt);
    __syncthreads();
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta;
    const int bid = blockIdx.x, tid = threadIdx.x;

    if (bid < g.n_lin_a) {
        // ---- role A: landmark-major, G lanes per landmark: Hpl tiles, Hll, b_l, weights, robust chi2
        constexpr int LPW = 256 / G;
        const int l = bid * LPW + tid / G, sub = tid % G;
        const bool lvalid = l < g.Nl;
        int k0 = 0, k1 = 0;
        Vec3 pw{ 0, 0, 0 };
        bool lfree = false;
        if (lvalid) {
            k0 = g.lm_ptr[l]; k1 = g.lm_ptr[l + 1];
            pw = Vec3{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] };
            lfree = !g.pt_fixed[l];
        }
        double h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, b0 = 0, b1 = 0, b2 = 0, chi_acc = 0;
        for (int k = k0 + sub; k < k1; k += G) {
            const int ip = g.obs_pose[k];
            const Rt T = load_Rt(sRt, ip);
            const double u = g.obs_uvr[3 * k], v = g.obs_uvr[3 * k + 1], ur = g.obs_uvr[3 * k + 2];
            Vec3 pc;
            const Vec3 e = stereo_error(T, pw, u, v, ur, K, pc);
            const double c2 = chi2_of(e, iv);
            const bool active = (g.obs_level[k] == 0) && g.obs_ok[k];
            double rho0 = c2, rho1 = 1.0;
            if (delta > 0.0) huber(c2, delta, rho0, rho1);
            g.obs_w[k] = active ? rho1 : 0.0;
            g.obs_chi2[k] = active ? c2 : 0.0;
            if (g.debug) { g.obs_err[3 * k] = active ? e.x : 0.0; g.obs_err[3 * k + 1] = active ? e.y : 0.0; g.obs_err[3 * k + 2] = active ? e.z : 0.0; }
            double2* Wk = reinterpret_cast<double2*>(g.W + 18 * (size_t)k);
            const bool pfree = g.pose_free[ip] >= 0;
            if (active) {
                chi_acc += rho0;
                const double wo = rho1 * iv;        // weightedOmega = rho' * Omega
                double Jp[9], Jx[18];
                stereo_jacobians(T, pc, K, Jp, Jx);
                if (lfree) {
                    h0 += Jp[0] * wo * Jp[0] + Jp[3] * wo * Jp[3] + Jp[6] * wo * Jp[6];
                    h1 += Jp[0] * wo * Jp[1] + Jp[3] * wo * Jp[4] + Jp[6] * wo * Jp[7];
                    h2 += Jp[0] * wo * Jp[2] + Jp[3] * wo * Jp[5] + Jp[6] * wo * Jp[8];
                    h3 += Jp[1] * wo * Jp[1] + Jp[4] * wo * Jp[4] + Jp[7] * wo * Jp[7];
                    h4 += Jp[1] * wo * Jp[2] + Jp[4] * wo * Jp[5] + Jp[7] * wo * Jp[8];
                    h5 += Jp[2] * wo * Jp[2] + Jp[5] * wo * Jp[5] + Jp[8] * wo * Jp[8];
                    b0 -= Jp[0] * wo * e.x + Jp[3] * wo * e.y + Jp[6] * wo * e.z;
                    b1 -= Jp[1] * wo * e.x + Jp[4] * wo * e.y + Jp[7] * wo * e.z;
                    b2 -= Jp[2] * wo * e.x + Jp[5] * wo * e.y + Jp[8] * wo * e.z;
                }
                if (pfree && lfree) {
                    double Wv[18];
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            Wv[r * 3 + c] = Jx[r] * wo * Jp[c] + Jx[6 + r] * wo * Jp[3 + c] + Jx[12 + r] * wo * Jp[6 + c];
#pragma unroll
                    for (int q = 0; q < 9; ++q) Wk[q] = make_double2(Wv[2 * q], Wv[2 * q + 1]);
                } else {
#pragma unroll
                    for (int q = 0; q < 9; ++q) Wk[q] = make_double2(0.0, 0.0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) Wk[q] = make_double2(0.0, 0.0);
            }
        }
        h0 = group_sum<G>(h0); h1 = group_sum<G>(h1); h2 = group_sum<G>(h2);
        h3 = group_sum<G>(h3); h4 = group_sum<G>(h4); h5 = group_sum<G>(h5);
        b0 = group_sum<G>(b0); b1 = group_sum<G>(b1); b2 = group_sum<G>(b2);
        double md = 0.0;
        if (lvalid && sub == 0) {
            double* H = g.Hll + 6 * (size_t)l;
            H[0] = h0; H[1] = h1; H[2] = h2; H[3] = h3; H[4] = h4; H[5] = h5;
            double* B = g.bl + 3 * (size_t)l;
            B[0] = b0; B[1] = b1; B[2] = b2;
            if (lfree) md = fmax(fabs(h0), fmax(fabs(h3), fabs(h5)));
        }
        const double chi_tot = block_sum_256(chi_acc, red);
        const double md_tot = block_max_256(md, red);
        if (tid == 0) { g.lin_part[2 * bid] = chi_tot; g.lin_part[2 * bid + 1] = md_tot; }
    } else if (bid < g.n_lin_a + g.n_chunks) {
        // ---- role B: pose-major chunk: upper triangle of Jx^T (rho' Omega) Jx and -Jx^T (rho' Omega) e
        const int c = bid - g.n_lin_a;
        const int a = g.chunk_pose[c];
        const int begin = g.chunk_ptr[c], end = g.chunk_ptr[c + 1];
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        if (begin + tid < end) {
            const int k = g.pose_obs[begin + tid];
            const bool active = (g.obs_level[k] == 0) && g.obs_ok[k];
            if (active) {
                const int ip = g.free_pose[a];
                const Rt T = load_Rt(sRt, ip);
                const int l = g.obs_pt[k];
                const Vec3 pw{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] };
                Vec3 pc;
                const Vec3 e = stereo_error(T, pw, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
                const double c2 = chi2_of(e, iv);
                double rho0 = c2, rho1 = 1.0;
                if (delta > 0.0) huber(c2, delta, rho0, rho1);
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
        }
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int q = 0; q < 27; ++q) {
            const double s = wave_sum(acc[q]);
            if (lane == 0) red[wave * 27 + q] = s;
        }
        __syncthreads();
        if (tid < 27) g.hpp_part[27 * (size_t)c + tid] = red[tid] + red[27 + tid] + red[54 + tid] + red[81 + tid];
    } else {
        // ---- role C: wheel-odometry edges (EdgePoseConstraint, Omega = I6 / odometryCovariance, no kernel)
        const double ic = g.inv_odo_cov;
        double chi_acc = 0.0;
        for (int e_ = tid; e_ < g.Ne; e_ += 256) {
            const int i = g.odo_i[e_], j = g.odo_j[e_];
            const bool fi = g.pose_free[i] >= 0, fj = g.pose_free[j] >= 0;
            double* o = g.odo_blk + 120 * (size_t)e_;
            if (!fi && !fj) { for (int q = 0; q < 120; ++q) o[q] = 0.0; continue; }     // allVerticesFixed
            double e[6], Ji[36], Jj[36];
            odo_linearize(pose + POSE_STRIDE * i, pose + POSE_STRIDE * j, g.odo_tq + 7 * e_, e, Ji, Jj);
            double c2 = 0.0;
#pragma unroll
            for (int d = 0; d < 6; ++d) c2 += e[d] * (ic * e[d]);
            chi_acc += c2;
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
        }
        const double chi_tot = block_sum_256(chi_acc, red);
        if (tid == 0) { g.lin_part[2 * g.n_lin_a] = chi_tot; g.lin_part[2 * g.n_lin_a + 1] = 0.0; }
    }
}

// Upper-triangle index of (r,c), r <= c, in the 21-entry packing used by role B.
__device__ __forceinline__ int upper_idx(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

// Single workgroup: finish K4 (sum chunk partials + odometry into Hpp/b_p), reduce chi2 and max|diag|,
// computeLambdaInit on the first iteration of a phase ([g2o-upstream] tau = 1e-5).
__global__ __launch_bounds__(1024) void k_lin_finalize(const DeviceGraph g) {
    LmState* st = g.st;
    if (st->done || !st->need_lin) return;
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    double md = 0.0;
    for (int t = tid; t < g.Npf * 42; t += 1024) {
        const int a = t / 42, q = t % 42;
        double v = 0.0;
        if (q < 36) {
            const int r = q / 6, c = q % 6;
            const int u = r <= c ? upper_idx(r, c) : upper_idx(c, r);
            for (int ch = g.pose_chunk_ptr[a]; ch < g.pose_chunk_ptr[a + 1]; ++ch) v += g.hpp_part[27 * (size_t)ch + u];
            for (int n = g.pose_odo_ptr[a]; n < g.pose_odo_ptr[a + 1]; ++n) {
                const int code = g.pose_odo[n];
                v += g.odo_blk[120 * (size_t)(code >> 1) + ((code & 1) ? 36 : 0) + q];
            }
            g.Hpp[36 * (size_t)a + q] = v;
            if (r == c) md = fmax(md, fabs(v));
        } else {
            const int r = q - 36;
            for (int ch = g.pose_chunk_ptr[a]; ch < g.pose_chunk_ptr[a + 1]; ++ch) v += g.hpp_part[27 * (size_t)ch + 21 + r];
            for (int n = g.pose_odo_ptr[a]; n < g.pose_odo_ptr[a + 1]; ++n) {
                const int code = g.pose_odo[n];
                v += g.odo_blk[120 * (size_t)(code >> 1) + ((code & 1) ? 114 : 108) + r];
            }
            g.bp[6 * (size_t)a + r] = v;
        }
    }
    // chi2 / max-diag partials of the workgroups of k_linearize
    double chi = 0.0;
    const int nparts = g.n_lin_a + 1;
    for (int w = tid; w < nparts; w += 1024) { chi += g.lin_part[2 * w]; md = fmax(md, g.lin_part[2 * w + 1]); }
    red[tid] = chi;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double chi_total = red[0];
    __syncthreads();
    red[tid] = md;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
    const double md_total = red[0];
    // poses without any active edge are outside g2o's active set: pin their block (dx = 0)
    for (int a = tid; a < g.Npf; a += 1024) {
        bool any = false;
        for (int r = 0; r < 6; ++r) any |= (g.Hpp[36 * (size_t)a + 7 * r] != 0.0);
        g.pose_pin[a] = any ? 0 : 1;
    }
    if (tid == 0) {
        st->current_chi = chi_total;
        st->max_diag = md_total;
        if (st->phase_iter == 0) {
            if (st->phase == 0) st->chi2_initial = chi_total;
            st->lambda = st->gauss_newton ? 0.0 : 1e-5 * md_total;
            st->ni = 2.0;
        }
        st->need_lin = 0;
        st->trial_q = 0;
        st->n_active[0] += 1;
    }
}

// ================================================================= K5: Schur complement (gather form)
// One wavefront per stored block (i <= j) of the reduced camera matrix:
//   S_ij = Hpp_ij (+lambda I on the diagonal) - sum_l Hpl_il (Hll_l + lambda I)^-1 Hpl_jl^T
//   b_s_i = b_p_i - sum_l Hpl_il (Hll_l + lambda I)^-1 b_l            (diagonal waves)
//   Minv_i = S_ii^-1  (block-Jacobi preconditioner of LinearSolverPCG) (diagonal waves)
__global__ __launch_bounds__(256) void k_schur(const DeviceGraph g) {
    const LmState* st = g.st;
    if (st->done || st->solve_state != 0) return;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= g.n_blk) return;
    const double lambda = st->lambda;
    const int i = g.blk_i[b], j = g.blk_j[b];
    const bool diag = (i == j);
    double acc[36], accb[6];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) accb[q] = 0.0;
    const int e0 = g.blk_ptr[b], e1 = g.blk_ptr[b + 1];
    for (int e = e0 + lane; e < e1; e += 64) {
        const int2 pr = g.blk_pairs[e];
        const int ka = pr.x, kb = pr.y;
        if (g.obs_w[ka] == 0.0 || g.obs_w[kb] == 0.0) continue;
        const int l = g.obs_pt[ka];
        const double* H = g.Hll + 6 * (size_t)l;
        const double h[6] = { H[0] + lambda, H[1], H[2], H[3] + lambda, H[4], H[5] + lambda };
        double D[6];
        sym3_inverse(h, D);
        double Wa[18], Wb[18];
        const double2* pa = reinterpret_cast<const double2*>(g.W + 18 * (size_t)ka);
        const double2* pb = reinterpret_cast<const double2*>(g.W + 18 * (size_t)kb);
#pragma unroll
        for (int q = 0; q < 9; ++q) { const double2 t = pa[q]; Wa[2 * q] = t.x; Wa[2 * q + 1] = t.y; }
#pragma unroll
        for (int q = 0; q < 9; ++q) { const double2 t = pb[q]; Wb[2 * q] = t.x; Wb[2 * q + 1] = t.y; }
        double Y[18];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            Y[r * 3 + 0] = Wa[r * 3] * D[0] + Wa[r * 3 + 1] * D[1] + Wa[r * 3 + 2] * D[2];
            Y[r * 3 + 1] = Wa[r * 3] * D[1] + Wa[r * 3 + 1] * D[3] + Wa[r * 3 + 2] * D[4];
            Y[r * 3 + 2] = Wa[r * 3] * D[2] + Wa[r * 3 + 1] * D[4] + Wa[r * 3 + 2] * D[5];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                acc[r * 6 + c] += Y[r * 3] * Wb[c * 3] + Y[r * 3 + 1] * Wb[c * 3 + 1] + Y[r * 3 + 2] * Wb[c * 3 + 2];
        if (diag) {
            const double* B = g.bl + 3 * (size_t)l;
#pragma unroll
            for (int r = 0; r < 6; ++r) accb[r] += Y[r * 3] * B[0] + Y[r * 3 + 1] * B[1] + Y[r * 3 + 2] * B[2];
        }
    }
    // fixed-order butterfly; element t of the block lands in lane t
    double mine = 0.0, mineb = 0.0;
#pragma unroll
    for (int q = 0; q < 36; ++q) { const double s = wave_sum(acc[q]); if (lane == q) mine = s; }
    if (diag) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { const double s = wave_sum(accb[q]); if (lane == q) mineb = s; }
    }
    const int r = lane / 6, c = lane % 6;    // meaningful for lane < 36
    double val = 0.0;
    if (lane < 36) {
        double base = 0.0;
        if (diag) base = g.Hpp[36 * (size_t)i + lane] + (r == c ? lambda : 0.0);
        for (int n = g.blk_odo_ptr[b]; n < g.blk_odo_ptr[b + 1]; ++n) {
            const int code = g.blk_odo[n];
            base += g.odo_blk[120 * (size_t)(code >> 1) + 72 + ((code & 1) ? (c * 6 + r) : lane)];
        }
        val = base - mine;
        if (diag && g.pose_pin[i]) val = (r == c) ? 1.0 : 0.0;
        g.S[36 * (size_t)b + lane] = val;
    }
    if (diag) {
        if (lane < 6) g.bs[6 * (size_t)i + lane] = g.pose_pin[i] ? 0.0 : (g.bp[6 * (size_t)i + lane] - mineb);
        // 36-lane Gauss-Jordan inverse of the SPD diagonal block (no pivoting needed)
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
        if (lane < 36) g.Minv[36 * (size_t)i + lane] = v;
    }
}

// ================================================================= K6: block-Jacobi PCG on S
// [g2o-upstream] LinearSolverPCG::solve: x0 = 0, tolerance 1e-6 on r^T M^-1 r, maxIter = rows,
// absolute tolerance carried in _residual between the solves of one optimize() call.
//
// k_pcg_init (one workgroup) starts a solve; k_pcg_iter (one wave per block row of S) performs, per
// launch, the vector half of iteration t-1 — redundantly and bitwise identically in every workgroup,
// from the q slices all rows wrote in the previous launch — then its own row of q = S d for iteration t.
// Control words and vectors are double-buffered on a host-supplied launch parity `hp`, so no launch
// ever reads a word that the same launch writes.
__global__ __launch_bounds__(256) void k_pcg_init(const DeviceGraph g, const int hp_write) {
    LmState* st = g.st;
    if (st->done || st->solve_state != 0) return;
    __shared__ double red[4];
    const int n6 = 6 * g.Npf, tid = threadIdx.x;
    double* r = g.pcg_r[hp_write];
    double* d = g.pcg_d[hp_write];
    double dn = 0.0;
    for (int t = tid; t < n6; t += 256) {
        const int a = t / 6, rr = t % 6;
        const double* M = g.Minv + 36 * (size_t)a + 6 * rr;
        const double* bb = g.bs + 6 * (size_t)a;
        const double dv = M[0] * bb[0] + M[1] * bb[1] + M[2] * bb[2] + M[3] * bb[3] + M[4] * bb[4] + M[5] * bb[5];
        r[t] = bb[rr];
        d[t] = dv;
        g.x[t] = 0.0;
        dn += bb[rr] * dv;
    }
    dn = block_sum_256(dn, red);
    if (tid == 0) {
        double d0 = 1e-6 * dn;
        if (st->pcg_residual > 0.0 && st->pcg_residual > d0) d0 = st->pcg_residual;
        st->pcg_d0 = d0;
        PcgCtl* c = g.pcg_ctl + hp_write;
        c->go = 1; c->has_q = 0; c->iter = 0; c->dn = dn;
        st->solve_state = 1;
    }
}

__global__ __launch_bounds__(64) void k_pcg_iter(const DeviceGraph g, const int hp) {
    const PcgCtl ctl = g.pcg_ctl[hp];
    PcgCtl* nxt = g.pcg_ctl + (hp ^ 1);
    const int lane = threadIdx.x, i = blockIdx.x;
    if (!ctl.go) { if (i == 0 && lane == 0) nxt->go = 0; return; }
    if (i == 0 && lane == 0) g.st->n_active[2] += 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n6 = 6 * g.Npf;
    double* sd = smem;                 // d (new)
    const double* r_old = g.pcg_r[hp];
    const double* d_old = g.pcg_d[hp];
    const double* q_old = g.pcg_q[hp];
    double dn = ctl.dn;
    int iter = ctl.iter;
    LmState* st = g.st;
    double xown = (lane < 6) ? g.x[6 * i + lane] : 0.0;
    if (ctl.has_q) {
        // vector half of the previous iteration (identical in every workgroup)
        double dq = 0.0;
        for (int t = lane; t < n6; t += 64) dq += d_old[t] * q_old[t];
        dq = wave_sum(dq);
        const double alpha = dn / dq;
        double* sr = smem + n6;        // r (new)
        double* ss = smem + 2 * n6;    // s = Minv r
        for (int t = lane; t < n6; t += 64) sr[t] = r_old[t] - alpha * q_old[t];
        __syncthreads();
        double dnn = 0.0;
        for (int t = lane; t < n6; t += 64) {
            const int a = t / 6, rr = t % 6;
            const double* M = g.Minv + 36 * (size_t)a + 6 * rr;
            const double* rb = sr + 6 * a;
            const double sv = M[0] * rb[0] + M[1] * rb[1] + M[2] * rb[2] + M[3] * rb[3] + M[4] * rb[4] + M[5] * rb[5];
            ss[t] = sv;
            dnn += sr[t] * sv;
        }
        dnn = wave_sum(dnn);
        const double beta = dnn / dn;
        for (int t = lane; t < n6; t += 64) sd[t] = ss[t] + beta * d_old[t];
        __syncthreads();
        // own slices: x += alpha d, r, d
        if (lane < 6) {
            const int t = 6 * i + lane;
            xown += alpha * d_old[t];
            g.x[t] = xown;
            g.pcg_r[hp ^ 1][t] = sr[t];
            g.pcg_d[hp ^ 1][t] = sd[t];
        }
        dn = dnn;
        iter += 1;
    } else {
        for (int t = lane; t < n6; t += 64) sd[t] = d_old[t];
        if (lane < 6) { const int t = 6 * i + lane; g.pcg_r[hp ^ 1][t] = r_old[t]; g.pcg_d[hp ^ 1][t] = d_old[t]; }
    }
    __syncthreads();
    if (dn <= st->pcg_d0 || iter >= n6 || !(dn == dn)) {
        // converged (or maxIter / NaN): x is final. K8 for this row's pose; row 0 publishes the verdict.
        double dx[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dx[q] = __shfl(xown, q, 64);
        if (lane == 0) {
            const int ip = g.free_pose[i];
            const int sel = st->sel;
            pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
        }
        if (i == 0 && lane == 0) {
            nxt->go = 0;
            st->pcg_residual = 0.5 * dn;
            st->pcg_iter = iter;
            st->pcg_total += iter;
            if (iter > st->pcg_max) st->pcg_max = iter;
            st->solve_state = 2;
        }
        return;
    }
    // own block row of q = S d: 8 slots x 8 lanes (6 rows used)
    const int slot = lane >> 3, rr = lane & 7;
    double acc = 0.0;
    if (rr < 6) {
        for (int n = g.row_ptr[i] + slot; n < g.row_ptr[i + 1]; n += 8) {
            const int j = g.row_col[n], code = g.row_blk[n];
            const double* Sb = g.S + 36 * (size_t)(code >> 1);
            const double* dj = sd + 6 * j;
            if (code & 1) acc += Sb[rr] * dj[0] + Sb[6 + rr] * dj[1] + Sb[12 + rr] * dj[2] + Sb[18 + rr] * dj[3] + Sb[24 + rr] * dj[4] + Sb[30 + rr] * dj[5];
            else { const double* Sr = Sb + 6 * rr; acc += Sr[0] * dj[0] + Sr[1] * dj[1] + Sr[2] * dj[2] + Sr[3] * dj[3] + Sr[4] * dj[4] + Sr[5] * dj[5]; }
        }
    }
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 6) g.pcg_q[hp ^ 1][6 * i + lane] = acc;
    if (i == 0 && lane == 0) { nxt->go = 1; nxt->has_q = 1; nxt->iter = iter; nxt->dn = dn; }
}

// ---- direct solver (Optimizer/Solver 0,1,3: sparse Cholesky in the reference) ----
__global__ __launch_bounds__(256) void k_dense_assemble(const DeviceGraph g) {
    const LmState* st = g.st;
    if (st->done || st->solve_state != 0) return;
    const int n6 = 6 * g.Npf;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < g.n_blk * 36; t += gridDim.x * 256) {
        const int b = t / 36, q = t % 36, r = q / 6, c = q % 6;
        const int i = g.blk_i[b], j = g.blk_j[b];
        const double v = g.S[t];
        g.dense[(size_t)(6 * i + r) * n6 + 6 * j + c] = v;
        g.dense[(size_t)(6 * j + c) * n6 + 6 * i + r] = v;
    }
}

// One workgroup: in-place right-looking Cholesky of the dense S (scratch cleared by the launcher: the
// factor fills in), forward/back substitution, K8 pose update.
__global__ __launch_bounds__(1024) void k_cholesky(const DeviceGraph g) {
    LmState* st = g.st;
    if (st->done || st->solve_state != 0) return;
    const int n = 6 * g.Npf, tid = threadIdx.x;
    double* A = g.dense;
    __shared__ double s_piv;
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            const double p = A[(size_t)k * n + k];
            if (!(p > 0.0) || !(p <= DBL_MAX)) s_fail = 1;
            s_piv = sqrt(p);
        }
        __syncthreads();
        if (s_fail) break;
        const double piv = s_piv;
        // scale column k (stored in row k of the lower triangle: A[i][k], i > k)
        for (int i2 = k + 1 + tid; i2 < n; i2 += 1024) A[(size_t)i2 * n + k] /= piv;
        if (tid == 0) A[(size_t)k * n + k] = piv;
        __syncthreads();
        // trailing update of the lower triangle
        const int m = n - k - 1;
        for (int t = tid; t < m * m; t += 1024) {
            const int ii = k + 1 + t / m, jj = k + 1 + t % m;
            if (jj <= ii) A[(size_t)ii * n + jj] -= A[(size_t)ii * n + k] * A[(size_t)jj * n + k];
        }
        __syncthreads();
    }
    if (s_fail) { if (tid == 0) st->solve_state = 3; return; }
    // L y = b ; L^T x = y  (serial over rows, parallel dot products)
    __shared__ double red[1024];
    double* x = g.x;
    for (int i2 = 0; i2 < n; ++i2) {
        double acc = 0.0;
        for (int k = tid; k < i2; k += 1024) acc += A[(size_t)i2 * n + k] * x[k];
        red[tid] = acc;
        __syncthreads();
        for (int s = 512; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        if (tid == 0) x[i2] = (g.bs[i2] - red[0]) / A[(size_t)i2 * n + i2];
        __syncthreads();
    }
    for (int i2 = n - 1; i2 >= 0; --i2) {
        double acc = 0.0;
        for (int k = i2 + 1 + tid; k < n; k += 1024) acc += A[(size_t)k * n + i2] * x[k];
        red[tid] = acc;
        __syncthreads();
        for (int s = 512; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        if (tid == 0) x[i2] = (x[i2] - red[0]) / A[(size_t)i2 * n + i2];
        __syncthreads();
    }
    const int sel = st->sel;
    for (int a = tid; a < g.Npf; a += 1024) {
        const int ip = g.free_pose[a];
        double dx[6];
        for (int q = 0; q < 6; ++q) dx[q] = x[6 * a + q];
        pose_oplus(g.pose[sel] + POSE_STRIDE * ip, dx, g.pose[sel ^ 1] + POSE_STRIDE * ip);
    }
    if (tid == 0) st->solve_state = 2;
}

// ================================================================= K7/K8 + chi2 at the trial state
template <int G>
__global__ __launch_bounds__(256) void k_backsub(const DeviceGraph g) {
    const LmState* st = g.st;
    if (st->done || st->solve_state != 2) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sRt = smem;
    double* red = smem + 12 * g.Np;
    const int sel = st->sel;
    const double* __restrict__ pose_t = g.pose[sel ^ 1];      // trial poses (written by the solver epilogue)
    const double* __restrict__ pt = g.pt[sel];
    double* __restrict__ pt_t = g.pt[sel ^ 1];
    const double lambda = st->lambda;
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta;
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (bid == g.n_lin_a) {
        // odometry chi2 at the trial state + the pose part of computeScale is done in k_decide
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
        const double chi_tot = block_sum_256(chi_acc, red);
        if (tid == 0) { g.trial_part[2 * bid] = chi_tot; g.trial_part[2 * bid + 1] = 0.0; }
        return;
    }
    stage_poses(pose_t, g.Np, sRt);
    __syncthreads();
    constexpr int LPW = 256 / G;
    const int l = bid * LPW + tid / G, sub = tid % G;
    const bool lvalid = l < g.Nl;
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
    for (int k = k0 + sub; k < k1; k += G) {
        const double w = g.obs_w[k];
        if (w == 0.0) continue;
        any = 1.0;
        const int a = g.pose_free[g.obs_pose[k]];
        if (a < 0 || !lfree) continue;
        const double2* pw2 = reinterpret_cast<const double2*>(g.W + 18 * (size_t)k);
        double Wv[18];
#pragma unroll
        for (int q = 0; q < 9; ++q) { const double2 t = pw2[q]; Wv[2 * q] = t.x; Wv[2 * q + 1] = t.y; }
        const double* xp = g.x + 6 * (size_t)a;
#pragma unroll
        for (int r = 0; r < 6; ++r) { const double xr = xp[r]; t0 += Wv[r * 3] * xr; t1 += Wv[r * 3 + 1] * xr; t2 += Wv[r * 3 + 2] * xr; }
    }
    t0 = group_sum<G>(t0); t1 = group_sum<G>(t1); t2 = group_sum<G>(t2);
    any = group_max<G>(any);
    double d0 = 0, d1 = 0, d2 = 0, scale_acc = 0.0;
    if (lvalid && lfree && any != 0.0) {
        const double* H = g.Hll + 6 * (size_t)l;
        const double* B = g.bl + 3 * (size_t)l;
        const double h[6] = { H[0] + lambda, H[1], H[2], H[3] + lambda, H[4], H[5] + lambda };
        double D[6];
        sym3_inverse(h, D);
        const double c0 = B[0] - t0, c1 = B[1] - t1, c2 = B[2] - t2;
        d0 = D[0] * c0 + D[1] * c1 + D[2] * c2;
        d1 = D[1] * c0 + D[3] * c1 + D[4] * c2;
        d2 = D[2] * c0 + D[4] * c1 + D[5] * c2;
        if (sub == 0) scale_acc = d0 * (lambda * d0 + B[0]) + d1 * (lambda * d1 + B[1]) + d2 * (lambda * d2 + B[2]);
    }
    const Vec3 pn{ pw.x + d0, pw.y + d1, pw.z + d2 };         // VertexPointXYZ::oplus
    if (lvalid && sub == 0) {
        pt_t[3 * l] = pn.x; pt_t[3 * l + 1] = pn.y; pt_t[3 * l + 2] = pn.z;
        g.dxl[3 * (size_t)l] = d0; g.dxl[3 * (size_t)l + 1] = d1; g.dxl[3 * (size_t)l + 2] = d2;
    }
    // computeActiveErrors + activeRobustChi2 at the trial state
    double chi_acc = 0.0;
    for (int k = k0 + sub; k < k1; k += G) {
        if (g.obs_w[k] == 0.0) continue;
        const Rt T = load_Rt(sRt, g.obs_pose[k]);
        Vec3 pc;
        const Vec3 e = stereo_error(T, pn, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
        const double c2 = chi2_of(e, iv);
        double rho0 = c2, rho1 = 1.0;
        if (delta > 0.0) huber(c2, delta, rho0, rho1);
        chi_acc += rho0;
    }
    const double chi_tot = block_sum_256(chi_acc, red);
    const double sc_tot = block_sum_256(scale_acc, red);
    if (tid == 0) { g.trial_part[2 * bid] = chi_tot; g.trial_part[2 * bid + 1] = sc_tot; }
}

// ================================================================= K9: Levenberg-Marquardt control
// [g2o-upstream] OptimizationAlgorithmLevenberg::solve, second half; one workgroup.
__global__ __launch_bounds__(256) void k_decide(const DeviceGraph g) {
    LmState* st = g.st;
    if (st->done || st->solve_state < 2) return;
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const bool ok = st->solve_state == 2;
    const double lambda = st->lambda;
    double chi = 0.0, sc = 0.0;
    if (ok) {
        for (int w = tid; w < g.n_lin_a + 1; w += 256) { chi += g.trial_part[2 * w]; sc += g.trial_part[2 * w + 1]; }
        for (int t = tid; t < 6 * g.Npf; t += 256) { const double x = g.x[t]; sc += x * (lambda * x + g.bp[t]); }
    }
    chi = block_sum_256(chi, red);
    sc = block_sum_256(sc, red);
    if (tid != 0) return;
    const int ph = st->phase;
    st->trials_run[ph] += 1;
    st->solve_state = 0;
    st->n_active[1] += 1;
    if (ok) st->n_active[3] += 1;
    if (st->gauss_newton) {
        // OptimizationAlgorithmGaussNewton: always take the step; Fail ends the phase
        if (ok) st->sel ^= 1;
        if (st->n_trace < MAX_TRACE) { st->trace_lambda[st->n_trace] = 0.0; st->trace_chi2[st->n_trace] = st->current_chi; st->n_trace++; }
        st->phase_iter += 1; st->iterations_run[ph] = st->phase_iter;
        st->need_lin = 1;
        if (!ok || st->phase_iter >= st->max_iter) st->done = 1;
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
        st->trial_q += 1;
        iteration_over = true;
    } else {
        const double nl = lambda * st->ni;
        st->lambda = nl;
        st->ni *= 2.0;                              // pop: estimate unchanged
        if (!(fabs(nl) <= DBL_MAX)) iteration_over = true;            // !isfinite(lambda): break before qmax++
        else {
            st->trial_q += 1;
            if (!(rho < 0.0) || st->trial_q >= 10) iteration_over = true;   // loop runs while rho < 0 && qmax < 10
        }
    }
    if (iteration_over) {
        if (st->trial_q == 10 || rho == 0.0) terminate = true;
        if (st->n_trace < MAX_TRACE) { st->trace_lambda[st->n_trace] = st->lambda; st->trace_chi2[st->n_trace] = st->current_chi; st->n_trace++; }
        st->phase_iter += 1; st->iterations_run[ph] = st->phase_iter;
        st->need_lin = 1;
        if (terminate || st->phase_iter >= st->max_iter) st->done = 1;
    }
}

// ================================================================= K10: per-edge chi2, outlier marking
// Optimizer.cpp:270-303: computeActiveErrors; edges with chi2() > kernel->delta() (UNSQUARED) go to level 1.
__global__ __launch_bounds__(256) void k_eval(const DeviceGraph g, const int mark) {
    LmState* st = g.st;
    if (st->status != 0) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sRt = smem;
    double* red = smem + 12 * g.Np;
    const int sel = st->sel;
    const double* __restrict__ pt = g.pt[sel];
    stage_poses(g.pose[sel], g.Np, sRt);
    __syncthreads();
    const Intrinsics K = intr_of(g);
    const double iv = g.inv_pixel_var, delta = g.huber_delta;
    const int tid = threadIdx.x, bid = blockIdx.x;
    double chi_acc = 0.0;
    int n_out = 0;
    if (bid == gridDim.x - 1) {
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
    } else {
        const int k = bid * 256 + tid;
        if (k < g.No) {
            const bool active = (g.obs_level[k] == 0) && g.obs_ok[k];
            double c2 = 0.0;
            if (active) {
                const int l = g.obs_pt[k];
                const Rt T = load_Rt(sRt, g.obs_pose[k]);
                Vec3 pc;
                const Vec3 e = stereo_error(T, Vec3{ pt[3 * l], pt[3 * l + 1], pt[3 * l + 2] }, g.obs_uvr[3 * k], g.obs_uvr[3 * k + 1], g.obs_uvr[3 * k + 2], K, pc);
                c2 = chi2_of(e, iv);
                double rho0 = c2, rho1 = 1.0;
                if (delta > 0.0) huber(c2, delta, rho0, rho1);
                chi_acc = rho0;
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
__global__ __launch_bounds__(256) void k_phase_end(const DeviceGraph g, const int nparts, const int phase_just_done, const int next_max_iter) {
    LmState* st = g.st;
    if (st->status != 0) return;
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double chi = 0.0, nout = 0.0;
    for (int w = tid; w < nparts; w += 256) { chi += g.trial_part[2 * w]; nout += g.trial_part[2 * w + 1]; }
    chi = block_sum_256(chi, red);
    nout = block_sum_256(nout, red);
    if (tid != 0) return;
    if (phase_just_done == 0) {
        st->chi2_phase1 = chi; st->chi2_final = chi;
        if (st->max_iter <= 0) st->chi2_initial = chi;        // optimize(0): nothing linearised
        if (chi != chi) st->status = 3;                                   // VISFS_BA_ERR_NAN_CHI2
        else if (chi > 1000000000000.0 || !(chi <= DBL_MAX)) st->status = 4; // VISFS_BA_ERR_HUGE_CHI2_1
        st->n_outliers = (int)nout;
        // arm phase 2: initializeOptimization(0) + optimize(iterations/2) re-initialise lambda and the PCG residual
        st->phase = 1; st->phase_iter = 0; st->max_iter = next_max_iter; st->trial_q = 0;
        st->need_lin = 1; st->solve_state = 0; st->pcg_residual = -1.0;
        st->done = (st->status != 0 || next_max_iter <= 0 || g.huber_delta <= 0.0) ? 1 : 0;
    } else {
        st->chi2_final = chi;
        if (chi > 1000000000000.0) st->status = 5;                       // VISFS_BA_ERR_HUGE_CHI2_2
    }
}

// Arm phase 1 on a fresh graph (all edges level 0, as the reference builds a new optimizer per call);
// restore != 0 also rewinds the estimates to the uploaded ones.
__global__ __launch_bounds__(256) void k_reset(const DeviceGraph g, const int max_iter, const int gauss_newton, const int restore) {
    const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (restore) {
        for (int t = gid; t < g.Np * POSE_STRIDE; t += stride) { const double v = g.pose0[t]; g.pose[0][t] = v; g.pose[1][t] = v; }
        for (int t = gid; t < g.Nl * 3; t += stride) { const double v = g.pt0[t]; g.pt[0][t] = v; g.pt[1][t] = v; }
    }
    for (int t = gid; t < g.No; t += stride) { g.obs_level[t] = 0; g.obs_outlier[t] = 0; g.obs_chi2_out[t] = 0.0; }
    if (gid == 0) {
        LmState* st = g.st;
        st->lambda = 0.0; st->ni = 2.0; st->current_chi = 0.0; st->temp_chi = 0.0; st->rho = 0.0; st->scale = 0.0; st->max_diag = 0.0;
        st->pcg_dn = 0.0; st->pcg_d0 = 0.0; st->pcg_residual = -1.0;
        st->chi2_initial = 0.0; st->chi2_phase1 = 0.0; st->chi2_final = 0.0;
        if (restore) st->sel = 0;
        st->pcg_max = 0; st->pad_ = 0;
        st->n_active[0] = st->n_active[1] = st->n_active[2] = st->n_active[3] = 0;
        st->phase = 0; st->max_iter = max_iter; st->phase_iter = 0; st->trial_q = 0;
        st->need_lin = 1; st->done = (max_iter <= 0) ? 1 : 0; st->solve_state = 0;
        st->pcg_iter = 0; st->pcg_total = 0; st->gauss_newton = gauss_newton; st->status = 0;
        st->n_outliers = 0; st->n_trace = 0;
        st->iterations_run[0] = st->iterations_run[1] = 0; st->trials_run[0] = st->trials_run[1] = 0;
        g.pcg_ctl[0].go = 0; g.pcg_ctl[1].go = 0;
    }
}

// Test hook: force the LM gates for a single stage call (visfs_ba_stage_*).
__global__ void k_stage_arm(const DeviceGraph g, const double lambda, const int need_lin) {
    LmState* st = g.st;
    st->done = 0; st->need_lin = need_lin; st->solve_state = 0; st->lambda = lambda; st->phase_iter = 1;
    st->max_iter = 1 << 30; st->trial_q = 0;
}

// ================================================================= launchers
static inline size_t lds_poses(const DeviceGraph& g, int extra) { return (size_t)(12 * g.Np + extra) * sizeof(double); }

template <int G>
static void launch_lin_t(const DeviceGraph& g, hipStream_t s) {
    const int grid = g.n_lin_a + g.n_chunks + 1;
    hipLaunchKernelGGL(k_linearize<G>, dim3(grid), dim3(256), lds_poses(g, 4 * 27), s, g);
}
template <int G>
static void launch_backsub_t(const DeviceGraph& g, hipStream_t s) {
    hipLaunchKernelGGL(k_backsub<G>, dim3(g.n_lin_a + 1), dim3(256), lds_poses(g, 8), s, g);
}

void launch_linearize(const DeviceGraph& g, hipStream_t s) {
    switch (g.group) {
        case 4: launch_lin_t<4>(g, s); break;
        case 8: launch_lin_t<8>(g, s); break;
        case 16: launch_lin_t<16>(g, s); break;
        case 32: launch_lin_t<32>(g, s); break;
        default: launch_lin_t<64>(g, s); break;
    }
}

void launch_lin_finalize(const DeviceGraph& g, hipStream_t s) {
    hipLaunchKernelGGL(k_lin_finalize, dim3(1), dim3(1024), 0, s, g);
}

void launch_schur(const DeviceGraph& g, hipStream_t s) {
    hipLaunchKernelGGL(k_schur, dim3((g.n_blk + 3) / 4), dim3(256), 0, s, g);
}

void launch_pcg_init(const DeviceGraph& g, int hp_write, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(256), 0, s, g, hp_write);
}

void launch_pcg_iter(const DeviceGraph& g, int hp, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_iter, dim3(g.Npf), dim3(64), (size_t)(3 * 6 * g.Npf) * sizeof(double), s, g, hp);
}

void launch_direct(const DeviceGraph& g, hipStream_t s) {
    const int grid = (g.n_blk * 36 + 255) / 256;
    const size_t n6 = (size_t)6 * g.Npf;
    (void)hipMemsetAsync(g.dense, 0, n6 * n6 * sizeof(double), s);
    hipLaunchKernelGGL(k_dense_assemble, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, g);
    hipLaunchKernelGGL(k_cholesky, dim3(1), dim3(1024), 0, s, g);
}

void launch_backsub(const DeviceGraph& g, hipStream_t s) {
    switch (g.group) {
        case 4: launch_backsub_t<4>(g, s); break;
        case 8: launch_backsub_t<8>(g, s); break;
        case 16: launch_backsub_t<16>(g, s); break;
        case 32: launch_backsub_t<32>(g, s); break;
        default: launch_backsub_t<64>(g, s); break;
    }
}

void launch_decide(const DeviceGraph& g, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, s, g);
}

void launch_phase_end(const DeviceGraph& g, int phase_just_done, int mark, int next_max_iter, hipStream_t s) {
    const int nb = (g.No + 255) / 256 + 1;
    hipLaunchKernelGGL(k_eval, dim3(nb), dim3(256), lds_poses(g, 8), s, g, mark);
    hipLaunchKernelGGL(k_phase_end, dim3(1), dim3(256), 0, s, g, nb, phase_just_done, next_max_iter);
}

void launch_reset(const DeviceGraph& g, int max_iter, int gauss_newton, int restore, hipStream_t s) {
    int grid = (g.No + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_reset, dim3(grid), dim3(256), 0, s, g, max_iter, gauss_newton, restore);
}

void launch_stage_arm(const DeviceGraph& g, double lambda, int need_lin, hipStream_t s) {
    hipLaunchKernelGGL(k_stage_arm, dim3(1), dim3(1), 0, s, g, lambda, need_lin);
}

int configure_kernels(const DeviceGraph& g) {
    // dynamic LDS above 64 KiB needs an explicit opt-in
    const size_t need = (size_t)(3 * 6 * g.Npf) * sizeof(double);
    if (need > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_pcg_iter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess) return -1;
    }
    return 0;
}

}  // namespace visfs_ba
