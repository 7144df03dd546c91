// ba_math.hpp — fp64 pose / edge arithmetic shared by the HIP kernels and the host packer.
//
// Every function restates VISFS's own vertex/edge arithmetic and cites the reference
// (paths relative to the reference repo root).  Written for gfx950: scalar-per-lane fp64,
// everything in registers, no local arrays indexed at run time.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define BA_HD __host__ __device__ __forceinline__

namespace visfs_ba {

struct Quat { double x, y, z, w; };
struct Vec3 { double x, y, z; };
struct Mat3 { double m00, m01, m02, m10, m11, m12, m20, m21, m22; };
// Rigid transform kept as rotation matrix + translation (the form the edges consume).
struct Rt { Mat3 R; Vec3 t; };

// Eigen::Quaternion::toRotationMatrix() as used by CameraPose::map (OptimizeTypeDefine.h:45-47).
BA_HD Mat3 quat_to_R(const Quat& q) {
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    Mat3 R;
    R.m00 = 1.0 - (tyy + tzz); R.m01 = txy - twz;         R.m02 = txz + twy;
    R.m10 = txy + twz;         R.m11 = 1.0 - (txx + tzz); R.m12 = tyz - twx;
    R.m20 = txz - twy;         R.m21 = tyz + twx;         R.m22 = 1.0 - (txx + tyy);
    return R;
}

// Eigen::Quaterniond(Matrix3d) — the conversion behind CameraPose(R,t) (OptimizeTypeDefine.h:30-34).
BA_HD Quat R_to_quat(const Mat3& m) {
    Quat q;
    double t = m.m00 + m.m11 + m.m22;
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m.m21 - m.m12) * t;
        q.y = (m.m02 - m.m20) * t;
        q.z = (m.m10 - m.m01) * t;
    } else if (m.m00 >= m.m11 && m.m00 >= m.m22) {          // i = 0, j = 1, k = 2
        t = sqrt(m.m00 - m.m11 - m.m22 + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (m.m21 - m.m12) * t; q.y = (m.m10 + m.m01) * t; q.z = (m.m20 + m.m02) * t;
    } else if (m.m11 > m.m00 && m.m11 >= m.m22) {           // i = 1, j = 2, k = 0
        t = sqrt(m.m11 - m.m22 - m.m00 + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (m.m02 - m.m20) * t; q.z = (m.m21 + m.m12) * t; q.x = (m.m01 + m.m10) * t;
    } else {                                                 // i = 2, j = 0, k = 1
        t = sqrt(m.m22 - m.m00 - m.m11 + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (m.m10 - m.m01) * t; q.x = (m.m02 + m.m20) * t; q.y = (m.m12 + m.m21) * t;
    }
    return q;
}

BA_HD Quat quat_mul(const Quat& a, const Quat& b) {
    Quat o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    o.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return o;
}

BA_HD Quat quat_normalized(const Quat& q) {
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Quat{ q.x / n, q.y / n, q.z / n, q.w / n };
}

// CameraPose::normalizeRotation (OptimizeTypeDefine.h:36-41) == QuaternionPositify (Math.h:308-317).
BA_HD Quat quat_positify(Quat q) {
    if (q.w < 0.0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    return quat_normalized(q);
}

// Eigen::Quaternion::inverse(): conjugate / squaredNorm.
BA_HD Quat quat_inv(const Quat& q) {
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return Quat{ -q.x / n2, -q.y / n2, -q.z / n2, q.w / n2 };
}

// Eigen::Quaternion * Vector3 (_transformVector).
BA_HD Vec3 quat_rot(const Quat& q, const Vec3& v) {
    double ux = q.y * v.z - q.z * v.y, uy = q.z * v.x - q.x * v.z, uz = q.x * v.y - q.y * v.x;
    ux += ux; uy += uy; uz += uz;
    return Vec3{ v.x + q.w * ux + (q.y * uz - q.z * uy),
                 v.y + q.w * uy + (q.z * ux - q.x * uz),
                 v.z + q.w * uz + (q.x * uy - q.y * ux) };
}

BA_HD Vec3 mat_vec(const Mat3& A, const Vec3& v) {
    return Vec3{ A.m00 * v.x + A.m01 * v.y + A.m02 * v.z,
                 A.m10 * v.x + A.m11 * v.y + A.m12 * v.z,
                 A.m20 * v.x + A.m21 * v.y + A.m22 * v.z };
}

BA_HD Mat3 mat_mul(const Mat3& A, const Mat3& B) {
    Mat3 C;
    C.m00 = A.m00 * B.m00 + A.m01 * B.m10 + A.m02 * B.m20;
    C.m01 = A.m00 * B.m01 + A.m01 * B.m11 + A.m02 * B.m21;
    C.m02 = A.m00 * B.m02 + A.m01 * B.m12 + A.m02 * B.m22;
    C.m10 = A.m10 * B.m00 + A.m11 * B.m10 + A.m12 * B.m20;
    C.m11 = A.m10 * B.m01 + A.m11 * B.m11 + A.m12 * B.m21;
    C.m12 = A.m10 * B.m02 + A.m11 * B.m12 + A.m12 * B.m22;
    C.m20 = A.m20 * B.m00 + A.m21 * B.m10 + A.m22 * B.m20;
    C.m21 = A.m20 * B.m01 + A.m21 * B.m11 + A.m22 * B.m21;
    C.m22 = A.m20 * B.m02 + A.m21 * B.m12 + A.m22 * B.m22;
    return C;
}

BA_HD Mat3 mat_T(const Mat3& A) {
    return Mat3{ A.m00, A.m10, A.m20, A.m01, A.m11, A.m21, A.m02, A.m12, A.m22 };
}

// skewSymmetric (Math.h:294-301)
BA_HD Mat3 skew(const Vec3& v) {
    return Mat3{ 0.0, -v.z, v.y, v.z, 0.0, -v.x, -v.y, v.x, 0.0 };
}

// ---- pose state: [tx ty tz qx qy qz qw] (CameraPose::toVector, OptimizeTypeDefine.h:57-67) ----
BA_HD Rt pose_to_Rt(const double* tq) {
    Rt p;
    p.R = quat_to_R(Quat{ tq[3], tq[4], tq[5], tq[6] });
    p.t = Vec3{ tq[0], tq[1], tq[2] };
    return p;
}

// CameraPose::update (OptimizeTypeDefine.cpp:7-14) with deltaQ (Math.h:277-287):
// t += dt; q = normalize((1, dtheta/2) * q) — first order, left-multiplied, not re-positified.
BA_HD void pose_oplus(const double* tq, const double* d, double* out) {
    out[0] = tq[0] + d[0]; out[1] = tq[1] + d[1]; out[2] = tq[2] + d[2];
    const Quat dq{ d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0 };
    const Quat q = quat_normalized(quat_mul(dq, Quat{ tq[3], tq[4], tq[5], tq[6] }));
    out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

struct Intrinsics { double fx, fy, cx, cy, bf; };

// EdgeStereo::computeError (OptimizeTypeDefine.h:121-126) + project (:180-187).
// Returns Pc = R Pw + t through pc (needed by the Jacobians).
BA_HD Vec3 stereo_error(const Rt& T, const Vec3& pw, double u, double v, double ur, const Intrinsics& K, Vec3& pc) {
    pc = mat_vec(T.R, pw);
    pc.x += T.t.x; pc.y += T.t.y; pc.z += T.t.z;
    const double invZ = 1.0 / pc.z;
    const double r0 = pc.x * invZ * K.fx + K.cx;
    const double r1 = pc.y * invZ * K.fy + K.cy;
    const double r2 = r0 - K.bf * invZ;
    return Vec3{ u - r0, v - r1, ur - r2 };
}

// EdgeStereo::linearizeOplus (OptimizeTypeDefine.h:134-178).
// Jp: d e / d point, 3x3 row-major (":145-155").  Jx: d e / d pose (dt | dtheta), 3x6 row-major (":157-176"),
// including the reference's SE(3)-style -[Pc]x rotation block (SURVEY §8a a6 quirk — reproduced as written).
// The reference divides by z and z^2 term by term (~24 fp64 divisions); on the GPU one reciprocal is taken and
// multiplied through (a division costs ~15 VALU instructions): values agree with the literal form to ~1 ulp.
BA_HD void stereo_jacobians(const Rt& T, const Vec3& pc, const Intrinsics& K, double Jp[9], double Jx[18]) {
    const double x = pc.x, y = pc.y;
    const double iz = 1.0 / pc.z, iz2 = iz * iz;
    const double fxz = K.fx * iz, fyz = K.fy * iz;              // fx/z, fy/z
    const double fxx = K.fx * x * iz2, fyy = K.fy * y * iz2;    // fx x/z^2, fy y/z^2
    const double bz = K.bf * iz2;                               // bf/z^2
    const Mat3& R = T.R;
    Jp[0] = -fxz * R.m00 + fxx * R.m20;
    Jp[1] = -fxz * R.m01 + fxx * R.m21;
    Jp[2] = -fxz * R.m02 + fxx * R.m22;
    Jp[3] = -fyz * R.m10 + fyy * R.m20;
    Jp[4] = -fyz * R.m11 + fyy * R.m21;
    Jp[5] = -fyz * R.m12 + fyy * R.m22;
    Jp[6] = Jp[0] - bz * R.m20;
    Jp[7] = Jp[1] - bz * R.m21;
    Jp[8] = Jp[2] - bz * R.m22;
    Jx[0] = -fxz;
    Jx[1] = 0.;
    Jx[2] = fxx;
    Jx[3] = fxx * y;
    Jx[4] = -(K.fx + fxx * x);
    Jx[5] = fxz * y;
    Jx[6] = 0.;
    Jx[7] = -fyz;
    Jx[8] = fyy;
    Jx[9] = K.fy + fyy * y;
    Jx[10] = -fyy * x;
    Jx[11] = -fyz * x;
    Jx[12] = Jx[0];
    Jx[13] = 0.;
    Jx[14] = Jx[2] - bz;
    Jx[15] = Jx[3] - bz * y;
    Jx[16] = Jx[4] + bz * x;
    Jx[17] = Jx[5];
}

// Only the pose Jacobian (pose-major pass).
BA_HD void stereo_jacobian_pose(const Vec3& pc, const Intrinsics& K, double Jx[18]) {
    const double x = pc.x, y = pc.y;
    const double iz = 1.0 / pc.z, iz2 = iz * iz;
    const double fxz = K.fx * iz, fyz = K.fy * iz;
    const double fxx = K.fx * x * iz2, fyy = K.fy * y * iz2;
    const double bz = K.bf * iz2;
    Jx[0] = -fxz;
    Jx[1] = 0.;
    Jx[2] = fxx;
    Jx[3] = fxx * y;
    Jx[4] = -(K.fx + fxx * x);
    Jx[5] = fxz * y;
    Jx[6] = 0.;
    Jx[7] = -fyz;
    Jx[8] = fyy;
    Jx[9] = K.fy + fyy * y;
    Jx[10] = -fyy * x;
    Jx[11] = -fyz * x;
    Jx[12] = Jx[0];
    Jx[13] = 0.;
    Jx[14] = Jx[2] - bz;
    Jx[15] = Jx[3] - bz * y;
    Jx[16] = Jx[4] + bz * x;
    Jx[17] = Jx[5];
}

// Hpl tile of one stereo edge, J_pose^T (rho' Omega) J_point (6x3 row-major), rebuilt from its 32-byte seed
// (Pc, wo) with the SAME formulas as linearizeOplus — the Schur gather and the back-substitution recompute the
// tile instead of re-reading 144 bytes per use.
BA_HD void hpl_tile(const Rt& T, const Vec3& pc, double wo, const Intrinsics& K, double Wv[18]) {
    if (wo == 0.0) {
#pragma unroll
        for (int q = 0; q < 18; ++q) Wv[q] = 0.0;
        return;
    }
    double Jp[9], Jx[18];
    stereo_jacobians(T, pc, K, Jp, Jx);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Wv[r * 3 + c] = Jx[r] * wo * Jp[c] + Jx[6 + r] * wo * Jp[3 + c] + Jx[12 + r] * wo * Jp[6 + c];
}

// Structure of the H_pl tile.  With A = d(pi)/d(Pc) (3x3, five non-zeros) the Jacobians of stereo_jacobians are
// Jp = -A R and Jx = [-A | A [Pc]x], so the tile W = Jx^T (wo I) Jp = [N ; [Pc]x N] with N = (wo A^T A) R: only the 3x3
// core N has to be formed, the lower half is three cross products.  A^T A = [[2a^2, 0, a(c+e)], [0, b^2, bd],
// [a(c+e), bd, c^2+d^2+e^2]] with a = fx/z, b = fy/z, c = -fx x/z^2, d = -fy y/z^2, e = c + bf/z^2.
// Same values as hpl_tile up to rounding (different association); used where the arithmetic is the bound (Schur gather).
// 1 / x where the arithmetic is the bound (the Schur gather and the back-substitution rebuild two tiles and invert one landmark block
// per co-observation pair): v_rcp_f64 + two Newton steps — 5 dependent operations instead of the ~14 of an IEEE division, correctly
// rounded but for rare last-bit cases.  Only the LINEAR SYSTEM passes through it (tiles, landmark inverses); residuals and chi2, which
// drive the LM decisions, keep IEEE divisions.  The host build (tests of the device functions) divides.
BA_HD double fast_recip(const double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
#else
    return 1.0 / x;
#endif
}
BA_HD void tile_core(const Mat3& R, const Vec3& pc, double wo, const Intrinsics& K, double N[9]) {
    const double iz = fast_recip(pc.z), iz2 = iz * iz;
    const double a = K.fx * iz, b = K.fy * iz;
    const double c = -K.fx * pc.x * iz2, d = -K.fy * pc.y * iz2, e = c + K.bf * iz2;
    const double m00 = wo * (2.0 * a * a), m02 = wo * (a * (c + e)), m11 = wo * (b * b), m12 = wo * (b * d);
    const double m22 = wo * (c * c + d * d + e * e);
    N[0] = m00 * R.m00 + m02 * R.m20; N[1] = m00 * R.m01 + m02 * R.m21; N[2] = m00 * R.m02 + m02 * R.m22;
    N[3] = m11 * R.m10 + m12 * R.m20; N[4] = m11 * R.m11 + m12 * R.m21; N[5] = m11 * R.m12 + m12 * R.m22;
    N[6] = m02 * R.m00 + m12 * R.m10 + m22 * R.m20;
    N[7] = m02 * R.m01 + m12 * R.m11 + m22 * R.m21;
    N[8] = m02 * R.m02 + m12 * R.m12 + m22 * R.m22;
}

// [g2o-upstream] RobustKernelHuber::robustify on chi2 = e^T Omega e (delta compared SQUARED).
BA_HD void huber(double e2, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else { const double s = sqrt(e2); rho0 = 2.0 * s * delta - dsqr; rho1 = delta / s; }
}
// [ceres-upstream] HuberLoss(a)::Evaluate on s = ||residual||^2: the same function for a > 0; the Ceres branch attaches it whatever
// robustKernelDelta is (Optimizer.cpp:370,469), so a <= 0 is defined too (rho' clamped to the smallest positive double).
BA_HD void huber_ceres(double s, double a, double& rho0, double& rho1) {
    const double b = a * a;
    if (s > b) { const double r = sqrt(s); rho0 = 2.0 * a * r - b; rho1 = fmax(2.2250738585072014e-308, a / r); }
    else { rho0 = s; rho1 = 1.0; }
}

// Inverse of a symmetric 3x3 given as (xx xy xz yy yz zz), cofactor form (Eigen's fixed-size 3x3 inverse).
BA_HD void sym3_inverse(const double h[6], double o[6]) {
    const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double id = fast_recip(a * c00 + b * c01 + c * c02);     // (a singular block gives inf or NaN: the callers test for both)
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
}

// [ceres-upstream] DoglegStrategy::ComputeTraditionalDoglegStep (dogleg_strategy.cc, Ceres 2.0 / 2.1) from the four inner products of the
// scaled space: S1 = ||g_s||^2, S2 = ||gn_s||^2, S3 = g_s . gn_s, JV2 = ||J v||^2 (v = the unscaled direction of -Cauchy / alpha).  The scaled
// step is A g_s + B gn_s, i.e. A v + B dn in the unscaled variables; norm = its length (dogleg_step_norm_); mcc = the model cost change
// -(g . step + step^T H step / 2) with v^T H v = JV2 and H dn = -g - mu M dn.
BA_HD void dogleg_combine(const double S1, const double S2, const double S3, const double JV2, const double radius, const double mu,
                          double& A, double& B, double& norm, double& mcc) {
    const double gnorm = sqrt(S1), gn_norm = sqrt(S2), alpha = S1 / JV2;
    if (gn_norm <= radius) { A = 0.0; B = 1.0; norm = gn_norm; }                          // case 1: the Gauss-Newton step lies inside the trust region
    else if (gnorm * alpha >= radius) { A = -radius / gnorm; B = 0.0; norm = radius; }     // case 2: even the Cauchy point lies outside: scaled back
    else {                                                                              // case 3: where the segment Cauchy -> Gauss-Newton leaves the region
        const double b_dot_a = -alpha * S3;
        const double a_sq = (alpha * gnorm) * (alpha * gnorm);
        const double bma_sq = a_sq - 2.0 * b_dot_a + gn_norm * gn_norm;
        const double c = b_dot_a - a_sq;
        const double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
        const double beta = (c <= 0.0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
        A = -alpha * (1.0 - beta); B = beta;
        norm = sqrt(A * A * S1 + 2.0 * A * B * S3 + B * B * S2);
    }
    const double sHs = A * A * JV2 + 2.0 * A * B * (-S1 - mu * S3) + B * B * (-S3 - mu * S2);
    mcc = -(A * S1 + B * S3 + 0.5 * sHs);
}

// ---- wheel-odometry edge: EdgePoseConstraint (OptimizeTypeDefine.cpp:35-88) ----
// Bottom-right 3x3 of QuaternionLeft(a) * QuaternionRight(b) (Math.h:324-345; both positify their argument).
BA_HD Mat3 quat_LR_br(const Quat& a_in, const Quat& b_in) {
    const Quat a = quat_positify(a_in), b = quat_positify(b_in);
    // L = [aw, -av^T; av, aw I + [av]x], R = [bw, -bv^T; bv, bw I - [bv]x]
    // (L R)[1:,1:] = av (-bv^T) + (aw I + [av]x)(bw I - [bv]x)
    const Mat3 A{ a.w, -a.z, a.y, a.z, a.w, -a.x, -a.y, a.x, a.w };
    const Mat3 B{ b.w, b.z, -b.y, -b.z, b.w, b.x, b.y, -b.x, b.w };
    Mat3 M = mat_mul(A, B);
    M.m00 -= a.x * b.x; M.m01 -= a.x * b.y; M.m02 -= a.x * b.z;
    M.m10 -= a.y * b.x; M.m11 -= a.y * b.y; M.m12 -= a.y * b.z;
    M.m20 -= a.z * b.x; M.m21 -= a.z * b.y; M.m22 -= a.z * b.z;
    return M;
}

// Bottom-right 3x3 of QuaternionLeft(a): aw I + [av]x on the positified a.
BA_HD Mat3 quat_L_br(const Quat& a_in) {
    const Quat a = quat_positify(a_in);
    return Mat3{ a.w, -a.z, a.y, a.z, a.w, -a.x, -a.y, a.x, a.w };
}

// e (6) only.
BA_HD void odo_error(const double* tq1, const double* tq2, const double* m, double e[6]) {
    const Quat Q1{ tq1[3], tq1[4], tq1[5], tq1[6] }, Q2{ tq2[3], tq2[4], tq2[5], tq2[6] }, mQ{ m[3], m[4], m[5], m[6] };
    const Quat Q2i = quat_inv(Q2);
    const Quat Q12 = quat_mul(Q1, Q2i);
    const Vec3 r = quat_rot(Q12, Vec3{ -tq2[0], -tq2[1], -tq2[2] });
    e[0] = r.x + tq1[0] - m[0]; e[1] = r.y + tq1[1] - m[1]; e[2] = r.z + tq1[2] - m[2];
    const Quat t2 = quat_mul(quat_mul(quat_inv(mQ), Q1), Q2i);
    e[3] = 2 * t2.x; e[4] = 2 * t2.y; e[5] = 2 * t2.z;
}

// e plus the "Left update" Jacobians (OptimizeTypeDefine.cpp:64-73); Ji, Jj 6x6 row-major.
BA_HD void odo_linearize(const double* tq1, const double* tq2, const double* m, double e[6], double Ji[36], double Jj[36]) {
    const Quat Q1{ tq1[3], tq1[4], tq1[5], tq1[6] }, Q2{ tq2[3], tq2[4], tq2[5], tq2[6] }, mQ{ m[3], m[4], m[5], m[6] };
    const Vec3 nP2{ -tq2[0], -tq2[1], -tq2[2] };
    const Quat Q2i = quat_inv(Q2);
    const Quat Q12 = quat_mul(Q1, Q2i);
    const Vec3 r = quat_rot(Q12, nP2);
    e[0] = r.x + tq1[0] - m[0]; e[1] = r.y + tq1[1] - m[1]; e[2] = r.z + tq1[2] - m[2];
    const Quat t2 = quat_mul(quat_mul(quat_inv(mQ), Q1), Q2i);
    e[3] = 2 * t2.x; e[4] = 2 * t2.y; e[5] = 2 * t2.z;
#pragma unroll
    for (int i = 0; i < 36; ++i) { Ji[i] = 0.0; Jj[i] = 0.0; }
    Ji[0] = Ji[7] = Ji[14] = 1.0;
    const Vec3 b = quat_rot(Q1, quat_rot(Q2i, nP2));
    const Mat3 S = skew(b);
    Ji[3] = -S.m00; Ji[4] = -S.m01; Ji[5] = -S.m02;
    Ji[9] = -S.m10; Ji[10] = -S.m11; Ji[11] = -S.m12;
    Ji[15] = -S.m20; Ji[16] = -S.m21; Ji[17] = -S.m22;
    const Mat3 LR = quat_LR_br(quat_mul(Q2, quat_inv(Q1)), mQ);
    Ji[21] = LR.m00; Ji[22] = LR.m01; Ji[23] = LR.m02;
    Ji[27] = LR.m10; Ji[28] = LR.m11; Ji[29] = LR.m12;
    Ji[33] = LR.m20; Ji[34] = LR.m21; Ji[35] = LR.m22;
    const Mat3 R12 = quat_to_R(Q12);
    Jj[0] = -R12.m00; Jj[1] = -R12.m01; Jj[2] = -R12.m02;
    Jj[6] = -R12.m10; Jj[7] = -R12.m11; Jj[8] = -R12.m12;
    Jj[12] = -R12.m20; Jj[13] = -R12.m21; Jj[14] = -R12.m22;
    const Mat3 U = mat_mul(mat_mul(quat_to_R(Q1), quat_to_R(Q2i)), skew(nP2));
    Jj[3] = U.m00; Jj[4] = U.m01; Jj[5] = U.m02;
    Jj[9] = U.m10; Jj[10] = U.m11; Jj[11] = U.m12;
    Jj[15] = U.m20; Jj[16] = U.m21; Jj[17] = U.m22;
    const Mat3 L = quat_L_br(t2);
    Jj[21] = -L.m00; Jj[22] = -L.m01; Jj[23] = -L.m02;
    Jj[27] = -L.m10; Jj[28] = -L.m11; Jj[29] = -L.m12;
    Jj[33] = -L.m20; Jj[34] = -L.m21; Jj[35] = -L.m22;
}

// ---- 3x4 row-major isometries (host packer: Optimizer.cpp:104-109, 131-140, 324-329) ----
BA_HD void iso_mul(const double* A, const double* B, double* C) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) C[r * 4 + c] = A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c] + A[r * 4 + 2] * B[8 + c];
        C[r * 4 + 3] = A[r * 4] * B[3] + A[r * 4 + 1] * B[7] + A[r * 4 + 2] * B[11] + A[r * 4 + 3];
    }
}
BA_HD void iso_inv(const double* A, double* C) {   // Eigen Transform::inverse(Isometry): R^T, -R^T t
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[r * 4 + c] = A[c * 4 + r];
    for (int r = 0; r < 3; ++r) C[r * 4 + 3] = -(C[r * 4] * A[3] + C[r * 4 + 1] * A[7] + C[r * 4 + 2] * A[11]);
}
BA_HD void iso_to_tq(const double* T, double* tq) {   // CameraPose(R,t) / g2o::SE3Quat(R,t)
    const Quat q = quat_positify(R_to_quat(Mat3{ T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] }));
    tq[0] = T[3]; tq[1] = T[7]; tq[2] = T[11];
    tq[3] = q.x; tq[4] = q.y; tq[5] = q.z; tq[6] = q.w;
}

// ---- laser occupied-space factor: EdgeOccupiedObservation (TypeOccupiedSpace2D.h:75-185) ----
// The probability grid as the edge reads it through GridArrayAdapter (TypeOccupiedSpace2D.h:22-48).
struct GridView {
    const float* cost;          // [ny][nx] Grid2D::getCorrespondenceCost(Array2i(x, y)), flat nx * y + x (Grid2d.h:93-95)
    int nx, ny;
    double resolution, max_x, max_y;
};
constexpr int kGridPadding = 2147483647 / 4;        // kPadding = INT_MAX / 4 (TypeOccupiedSpace2D.h:20)
constexpr double kMaxCorrespondenceCost = 1.0 - 0.1; // Map::kMaxCorrespondenceCost (ProbabilityValues.h:41-44)

// GridArrayAdapter::GetValue: outside the padded window the constant, inside the float cost widened to double.
BA_HD double grid_value(const GridView& g, int row, int col) {
    const int y = row - kGridPadding, x = col - kGridPadding;
    if (y < 0 || x < 0 || y >= g.ny || x >= g.nx) return kMaxCorrespondenceCost;
    return static_cast<double>(g.cost[g.nx * y + x]);
}
// [ceres-upstream] CubicHermiteSpline<1>: Catmull-Rom spline through p1..p2, Horner form.
BA_HD void cubic_hermite(double p0, double p1, double p2, double p3, double x, double& f, double& dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    f = p1 + x * (c + x * (b + x * a));
    dfdx = c + x * (2.0 * b + 3.0 * a * x);
}
// [ceres-upstream] BiCubicInterpolator::Evaluate(r, c, f, dfdr, dfdc): four row splines, then the column spline.
BA_HD void bicubic(const GridView& g, double r, double c, double& f, double& dfdr, double& dfdc) {
    const int row = static_cast<int>(floor(r)), col = static_cast<int>(floor(c));
    double fr[4], dfr[4];
    for (int i = 0; i < 4; ++i)
        cubic_hermite(grid_value(g, row - 1 + i, col - 1), grid_value(g, row - 1 + i, col), grid_value(g, row - 1 + i, col + 1),
                      grid_value(g, row - 1 + i, col + 2), c - col, fr[i], dfr[i]);
    double unused;
    cubic_hermite(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
    cubic_hermite(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, unused);
}
// The functor (TypeOccupiedSpace2D.h:97-124) for quaternion (x, y, z, w) taken WITHOUT normalisation, as Eigen's
// toRotationMatrix does: Twc = [R | t]^-1 * Tcr (Isometry inverse: R^T, -R^T t), Po = Twc * P, grid coordinates
// (max - Po) / resolution - 0.5 + kPadding.  Also returns m = Rcr P + tcr - t and R for the Jacobian.
BA_HD void laser_grid_coords(const double* tq, double w, const double* Tcr, const Vec3& P, const GridView& g,
                             double& r, double& c, Mat3& R, Vec3& m) {
    R = quat_to_R(Quat{ tq[3], tq[4], tq[5], w });
    // inverse translation -(R^T t)
    const Vec3 ti{ -(R.m00 * tq[0] + R.m10 * tq[1] + R.m20 * tq[2]), -(R.m01 * tq[0] + R.m11 * tq[1] + R.m21 * tq[2]),
                   -(R.m02 * tq[0] + R.m12 * tq[1] + R.m22 * tq[2]) };
    // Twc = inverse * Tcr: only rows 0 and 1 are needed
    const double L00 = R.m00 * Tcr[0] + R.m10 * Tcr[4] + R.m20 * Tcr[8], L01 = R.m00 * Tcr[1] + R.m10 * Tcr[5] + R.m20 * Tcr[9],
                 L02 = R.m00 * Tcr[2] + R.m10 * Tcr[6] + R.m20 * Tcr[10], T0 = R.m00 * Tcr[3] + R.m10 * Tcr[7] + R.m20 * Tcr[11] + ti.x;
    const double L10 = R.m01 * Tcr[0] + R.m11 * Tcr[4] + R.m21 * Tcr[8], L11 = R.m01 * Tcr[1] + R.m11 * Tcr[5] + R.m21 * Tcr[9],
                 L12 = R.m01 * Tcr[2] + R.m11 * Tcr[6] + R.m21 * Tcr[10], T1 = R.m01 * Tcr[3] + R.m11 * Tcr[7] + R.m21 * Tcr[11] + ti.y;
    const double Po0 = L00 * P.x + L01 * P.y + L02 * P.z + T0;
    const double Po1 = L10 * P.x + L11 * P.y + L12 * P.z + T1;
    r = (g.max_x - Po0) / g.resolution - 0.5 + static_cast<double>(kGridPadding);
    c = (g.max_y - Po1) / g.resolution - 0.5 + static_cast<double>(kGridPadding);
    m = Vec3{ Tcr[0] * P.x + Tcr[1] * P.y + Tcr[2] * P.z + Tcr[3] - tq[0], Tcr[4] * P.x + Tcr[5] * P.y + Tcr[6] * P.z + Tcr[7] - tq[1],
              Tcr[8] * P.x + Tcr[9] * P.y + Tcr[10] * P.z + Tcr[11] - tq[2] };
}
// computeError (TypeOccupiedSpace2D.h:126-131).
BA_HD double laser_error(const double* tq, const double* Tcr, const Vec3& P, const GridView& g) {
    double r, c, f, dr, dc; Mat3 R; Vec3 m;
    laser_grid_coords(tq, tq[6], Tcr, P, g, r, c, R, m);
    bicubic(g, r, c, f, dr, dc);
    return f;
}
// linearizeOplus (TypeOccupiedSpace2D.h:145-179).  The reference runs ceres autodiff over StaticParameterDims<6, 3>:
// the pose block carries six jets (t, qx, qy, qz), so the functor's pose[6] aliases the first coordinate of the range
// point.  Its Jacobian is therefore that of the functor with q.w := P.x, w.r.t. (t1 t2 t3 qx qy qz), evaluated at the
// grid position that aliased pose maps to.  Reproduced analytically: Po = R^T m, dPo/dt = -R^T, dPo/dq = (dR^T/dq) m.
// true_w: differentiate with the pose's own q.w (the Ceres factor: AutoDiffCostFunction<..., 7> over the full pose, then the
// [I6; 0] Jacobian of PoseLocalParameterization drops the q.w column, OccupiedSpace2dFactor.cpp:93-97) instead of the g2o edge's
// aliased value (TypeOccupiedSpace2D.h: the six-jet block makes the functor's pose[6] read the range point's x).
BA_HD void laser_jacobian(const double* tq, const double* Tcr, const Vec3& P, const GridView& g, double J[6], const bool true_w = false) {
    double r, c, f, dfdr, dfdc; Mat3 R; Vec3 m;
    const double w = true_w ? tq[6] : P.x;
    laser_grid_coords(tq, w, Tcr, P, g, r, c, R, m);
    bicubic(g, r, c, f, dfdr, dfdc);
    const double x = tq[3], y = tq[4], z = tq[5];
    const double d0[6] = { -R.m00, -R.m10, -R.m20,
                           2.0 * y * m.y + 2.0 * z * m.z, -4.0 * y * m.x + 2.0 * x * m.y - 2.0 * w * m.z, -4.0 * z * m.x + 2.0 * w * m.y + 2.0 * x * m.z };
    const double d1[6] = { -R.m01, -R.m11, -R.m21,
                           2.0 * y * m.x - 4.0 * x * m.y + 2.0 * w * m.z, 2.0 * x * m.x + 2.0 * z * m.z, -2.0 * w * m.x - 4.0 * z * m.y + 2.0 * y * m.z };
    for (int i = 0; i < 6; ++i) J[i] = dfdr * (-d0[i] / g.resolution) + dfdc * (-d1[i] / g.resolution);
}

}  // namespace visfs_ba
