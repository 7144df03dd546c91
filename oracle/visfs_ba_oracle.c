/*
 * visfs_ba_oracle.c — CPU oracle (plain C) for VISFS's sliding-window BA.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see visfs_ba_oracle.h).  PARITY UNPINNED.
 *
 * Reference citations are relative to /root/reference/.  [g2o-upstream] marks
 * statements restated from g2o's published sources (not vendored by the reference).
 */
#define _GNU_SOURCE            /* sched_setaffinity / CPU_SET for oracle_omp_pin (bench.py's cpu_baseline leg) */
#define _POSIX_C_SOURCE 200809L
#include "visfs_ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ===================================================================== */
/* small fixed-size helpers (Eigen semantics restated)                    */
/* ===================================================================== */

/* Eigen::Quaternion::toRotationMatrix().  q = [x y z w]. */
static void quat_to_R(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* Eigen::Quaternion(Matrix3) (quaternionbase_assign_impl<Other,3,3>). */
static void R_to_quat(const double m[9], double q[4]) {
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

static void quat_mul(const double a[4], const double b[4], double o[4]) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

static void quat_normalize(double q[4]) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* CameraPose::normalizeRotation (OptimizeTypeDefine.h:36-41) == QuaternionPositify (Math.h:308-317) */
static void quat_positify(double q[4]) {
    if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    quat_normalize(q);
}

/* Eigen::Quaternion::inverse(): conjugate / squaredNorm. */
static void quat_inv(const double q[4], double o[4]) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    o[0] = -q[0] / n2; o[1] = -q[1] / n2; o[2] = -q[2] / n2; o[3] = q[3] / n2;
}

/* Eigen::Quaternion::_transformVector: v + w*(2 qv x v) + qv x (2 qv x v). */
static void quat_rot(const double q[4], const double v[3], double o[3]) {
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = A[r * 3] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
}
/* skewSymmetric (Math.h:294-301) */
static void skew(const double v[3], double S[9]) {
    S[0] = 0.0;   S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0.0;   S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0.0;
}

/* 3x4 row-major isometry helpers (Eigen::Isometry3d product / inverse(Isometry)) */
static void iso_mul(const double A[12], const double B[12], double C[12]) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            C[r * 4 + c] = A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c] + A[r * 4 + 2] * B[8 + c];
        C[r * 4 + 3] = A[r * 4] * B[3] + A[r * 4 + 1] * B[7] + A[r * 4 + 2] * B[11] + A[r * 4 + 3];
    }
}
static void iso_inv(const double A[12], double C[12]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 4 + c] = A[c * 4 + r];
    for (int r = 0; r < 3; ++r)
        C[r * 4 + 3] = -(C[r * 4] * A[3] + C[r * 4 + 1] * A[7] + C[r * 4 + 2] * A[11]);
}

/* ===================================================================== */
/* VISFS-owned arithmetic                                                 */
/* ===================================================================== */

void oracle_pose_from_Rt(const double R[9], const double t[3], double tq[7]) {
    /* CameraPose(const Matrix3d&, const Vector3d&): OptimizeTypeDefine.h:30-34 */
    double q[4];
    R_to_quat(R, q);
    quat_positify(q);
    tq[0] = t[0]; tq[1] = t[1]; tq[2] = t[2];
    tq[3] = q[0]; tq[4] = q[1]; tq[5] = q[2]; tq[6] = q[3];
}

void oracle_pose_to_Rt(const double tq[7], double R[9], double t[3]) {
    /* toHomogeneousMatrix: OptimizeTypeDefine.h:74-81 */
    quat_to_R(tq + 3, R);
    t[0] = tq[0]; t[1] = tq[1]; t[2] = tq[2];
}

void oracle_pose_update(double tq[7], const double d[6]) {
    /* CameraPose::update, OptimizeTypeDefine.cpp:7-14; deltaQ, Math.h:277-287 (first order, NOT normalised) */
    tq[0] += d[0]; tq[1] += d[1]; tq[2] += d[2];
    const double dq[4] = { d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0 };
    double q[4];
    quat_mul(dq, tq + 3, q);          /* q_ = dq * q_ */
    quat_normalize(q);                /* no re-positify */
    tq[3] = q[0]; tq[4] = q[1]; tq[5] = q[2]; tq[6] = q[3];
}

void oracle_stereo_edge(const double tq[7], const double pw[3], const double uvr[3],
                        const double intr[5], double e[3], double Ji[9], double Jj[18]) {
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3], bf = intr[4];
    double R[9], pc[3];
    quat_to_R(tq + 3, R);
    /* CameraPose::map, OptimizeTypeDefine.h:45-47 */
    mat3_vec(R, pw, pc);
    pc[0] += tq[0]; pc[1] += tq[1]; pc[2] += tq[2];
    /* project, OptimizeTypeDefine.h:180-187 */
    const double invZ = 1.0 / pc[2];
    double res[3];
    res[0] = pc[0] * invZ * fx + cx;
    res[1] = pc[1] * invZ * fy + cy;
    res[2] = res[0] - bf * invZ;
    /* computeError, :121-126 */
    e[0] = uvr[0] - res[0]; e[1] = uvr[1] - res[1]; e[2] = uvr[2] - res[2];
    if (!Ji && !Jj) return;
    /* linearizeOplus, :134-178 (vertex 0 = point → Xi, vertex 1 = pose → Xj) */
    const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
    if (Ji) {
        Ji[0] = -fx * R[0] / z + fx * x * R[6] / z_2;
        Ji[1] = -fx * R[1] / z + fx * x * R[7] / z_2;
        Ji[2] = -fx * R[2] / z + fx * x * R[8] / z_2;
        Ji[3] = -fy * R[3] / z + fy * y * R[6] / z_2;
        Ji[4] = -fy * R[4] / z + fy * y * R[7] / z_2;
        Ji[5] = -fy * R[5] / z + fy * y * R[8] / z_2;
        Ji[6] = Ji[0] - bf * R[6] / z_2;
        Ji[7] = Ji[1] - bf * R[7] / z_2;
        Ji[8] = Ji[2] - bf * R[8] / z_2;
    }
    if (Jj) {
        Jj[0] = -1. / z * fx;
        Jj[1] = 0.;
        Jj[2] = x / z_2 * fx;
        Jj[3] = x * y / z_2 * fx;
        Jj[4] = -(1. + (x * x / z_2)) * fx;
        Jj[5] = y / z * fx;
        Jj[6] = 0.;
        Jj[7] = -1. / z * fy;
        Jj[8] = y / z_2 * fy;
        Jj[9] = (1. + y * y / z_2) * fy;
        Jj[10] = -x * y / z_2 * fy;
        Jj[11] = -x / z * fy;
        Jj[12] = Jj[0];
        Jj[13] = 0.;
        Jj[14] = Jj[2] - bf / z_2;
        Jj[15] = Jj[3] - bf * y / z_2;
        Jj[16] = Jj[4] + bf * x / z_2;
        Jj[17] = Jj[5];
    }
}

/* QuaternionLeft / QuaternionRight bottom-right 3x3 products need the full 4x4 (Math.h:324-345). Layout [w; x y z]. */
static void quat_left4(const double qin[4], double M[16]) {
    double q[4] = { qin[0], qin[1], qin[2], qin[3] };
    quat_positify(q);
    const double w = q[3], v[3] = { q[0], q[1], q[2] };
    double S[9];
    skew(v, S);
    M[0] = w; M[1] = -v[0]; M[2] = -v[1]; M[3] = -v[2];
    for (int r = 0; r < 3; ++r) {
        M[(r + 1) * 4] = v[r];
        for (int c = 0; c < 3; ++c) M[(r + 1) * 4 + c + 1] = (r == c ? w : 0.0) + S[r * 3 + c];
    }
}
static void quat_right4(const double qin[4], double M[16]) {
    double q[4] = { qin[0], qin[1], qin[2], qin[3] };
    quat_positify(q);
    const double w = q[3], v[3] = { q[0], q[1], q[2] };
    double S[9];
    skew(v, S);
    M[0] = w; M[1] = -v[0]; M[2] = -v[1]; M[3] = -v[2];
    for (int r = 0; r < 3; ++r) {
        M[(r + 1) * 4] = v[r];
        for (int c = 0; c < 3; ++c) M[(r + 1) * 4 + c + 1] = (r == c ? w : 0.0) - S[r * 3 + c];
    }
}

void oracle_odo_edge(const double tq1[7], const double tq2[7], const double m[7],
                     double e[6], double Ji[36], double Jj[36]) {
    const double* P1 = tq1; const double* Q1 = tq1 + 3;
    const double* P2 = tq2; const double* Q2 = tq2 + 3;
    const double* mP = m;   const double* mQ = m + 3;
    double Q2i[4], Q12[4], mQi[4], nP2[3] = { -P2[0], -P2[1], -P2[2] };
    quat_inv(Q2, Q2i);
    quat_mul(Q1, Q2i, Q12);                      /* sQ1*sQ2.inverse() */
    quat_inv(mQ, mQi);
    /* computeError, OptimizeTypeDefine.cpp:47-48 */
    double r[3];
    quat_rot(Q12, nP2, r);
    e[0] = r[0] + P1[0] - mP[0]; e[1] = r[1] + P1[1] - mP[1]; e[2] = r[2] + P1[2] - mP[2];
    double t1[4], t2[4];
    quat_mul(mQi, Q1, t1);
    quat_mul(t1, Q2i, t2);                       /* mQ12.inverse()*sQ1*sQ2.inverse() */
    e[3] = 2 * t2[0]; e[4] = 2 * t2[1]; e[5] = 2 * t2[2];
    if (!Ji && !Jj) return;
    /* linearizeOplus ("Left update"), OptimizeTypeDefine.cpp:64-73 */
    double a[3], b[3], S[9];
    if (Ji) {
        memset(Ji, 0, 36 * sizeof(double));
        Ji[0] = Ji[7] = Ji[14] = 1.0;
        quat_rot(Q2i, nP2, a);                   /* sQ2.inverse()*(-sP2) */
        quat_rot(Q1, a, b);                      /* sQ1*( ... ) */
        skew(b, S);
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) Ji[rr * 6 + 3 + c] = -S[rr * 3 + c];
        double Q1i[4], Q21[4], L[16], Rm[16];
        quat_inv(Q1, Q1i);
        quat_mul(Q2, Q1i, Q21);                  /* sQ2*sQ1.inverse() */
        quat_left4(Q21, L);
        quat_right4(mQ, Rm);
        for (int rr = 0; rr < 3; ++rr)
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += L[(rr + 1) * 4 + k] * Rm[k * 4 + (c + 1)];
                Ji[(rr + 3) * 6 + 3 + c] = s;
            }
    }
    if (Jj) {
        memset(Jj, 0, 36 * sizeof(double));
        double R12[9], R1[9], R2i[9], T[9], U[9];
        quat_to_R(Q12, R12);
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) Jj[rr * 6 + c] = -R12[rr * 3 + c];
        quat_to_R(Q1, R1);
        quat_to_R(Q2i, R2i);
        skew(nP2, S);
        mat3_mul(R1, R2i, T);
        mat3_mul(T, S, U);
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) Jj[rr * 6 + 3 + c] = U[rr * 3 + c];
        double L[16];
        quat_left4(t2, L);
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) Jj[(rr + 3) * 6 + 3 + c] = -L[(rr + 1) * 4 + c + 1];
    }
}

/* ===================================================================== */
/* EdgeOccupiedObservation (TypeOccupiedSpace2D.h:75-185)                     */
/* ===================================================================== */
#define ORACLE_K_PADDING (2147483647 / 4)            /* kPadding = INT_MAX / 4, TypeOccupiedSpace2D.h:20 */
#define ORACLE_K_MAX_COST (1.0 - 0.1)                /* Map::kMaxCorrespondenceCost = 1 - kMinProbability, ProbabilityValues.h:41-44 */

/* GridArrayAdapter::GetValue (TypeOccupiedSpace2D.h:28-37) over Grid2D::getCorrespondenceCost (Grid2d.h:33-36). */
static double grid_value(const visfs_ba_grid* g, int row, int col) {
    const int num_rows = g->num_y_cells + 2 * ORACLE_K_PADDING, num_cols = g->num_x_cells + 2 * ORACLE_K_PADDING;
    if (row < ORACLE_K_PADDING || col < ORACLE_K_PADDING || row >= num_rows - ORACLE_K_PADDING || col >= num_cols - ORACLE_K_PADDING)
        return ORACLE_K_MAX_COST;
    const int x = col - ORACLE_K_PADDING, y = row - ORACLE_K_PADDING;      /* Array2i(column - kPadding, row - kPadding) */
    return (double)g->correspondence_cost[g->num_x_cells * y + x];         /* toFlatIndex, Grid2d.h:93-95 */
}

/* [ceres-upstream] CubicHermiteSpline<1> (ceres/cubic_interpolation.h): Catmull-Rom, Horner form. */
static void cubic_hermite(double p0, double p1, double p2, double p3, double x, double* f, double* dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    const double d = p1;
    if (f) *f = d + x * (c + x * (b + x * a));
    if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

/* [ceres-upstream] BiCubicInterpolator::Evaluate(r, c, f, dfdr, dfdc): rows first, then the column spline. */
void oracle_bicubic(const visfs_ba_grid* g, double r, double c, double* f, double* dfdr, double* dfdc) {
    const int row = (int)floor(r), col = (int)floor(c);
    double fr[4], dfr[4];
    for (int i = 0; i < 4; ++i) {
        const double p0 = grid_value(g, row - 1 + i, col - 1), p1 = grid_value(g, row - 1 + i, col);
        const double p2 = grid_value(g, row - 1 + i, col + 1), p3 = grid_value(g, row - 1 + i, col + 2);
        cubic_hermite(p0, p1, p2, p3, c - col, &fr[i], &dfr[i]);
    }
    cubic_hermite(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
    if (dfdc) cubic_hermite(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, NULL);
}

/* A ceres::Jet restricted to the six pose directions: value + 6 partials. */
typedef struct { double a; double v[6]; } jet6;
static jet6 jc(double a) { jet6 r; r.a = a; for (int i = 0; i < 6; ++i) r.v[i] = 0.0; return r; }
static jet6 jadd(jet6 x, jet6 y) { jet6 r; r.a = x.a + y.a; for (int i = 0; i < 6; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
static jet6 jsub(jet6 x, jet6 y) { jet6 r; r.a = x.a - y.a; for (int i = 0; i < 6; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
static jet6 jmul(jet6 x, jet6 y) { jet6 r; r.a = x.a * y.a; for (int i = 0; i < 6; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
static jet6 jneg(jet6 x) { jet6 r; r.a = -x.a; for (int i = 0; i < 6; ++i) r.v[i] = -x.v[i]; return r; }
static jet6 jscale(jet6 x, double k) { jet6 r; r.a = x.a * k; for (int i = 0; i < 6; ++i) r.v[i] = x.v[i] * k; return r; }

/* The functor body (TypeOccupiedSpace2D.h:97-124) on jets: Quaternion(w,x,y,z).toRotationMatrix() WITHOUT normalisation
 * (Eigen), Tiw = [R | t], Twc = Tiw.inverse() (Isometry: R^T, -R^T t) * Tcr, Po = Twc * P, then the grid coordinates. */
static void laser_functor(const jet6 pose[7], const double P[3], const double Tcr[12], const visfs_ba_grid* g, jet6* rr, jet6* cc) {
    const jet6 w = pose[6], x = pose[3], y = pose[4], z = pose[5];
    const jet6 tx = jscale(x, 2.0), ty = jscale(y, 2.0), tz = jscale(z, 2.0);
    const jet6 twx = jmul(tx, w), twy = jmul(ty, w), twz = jmul(tz, w);
    const jet6 txx = jmul(tx, x), txy = jmul(ty, x), txz = jmul(tz, x);
    const jet6 tyy = jmul(ty, y), tyz = jmul(tz, y), tzz = jmul(tz, z);
    const jet6 one = jc(1.0);
    jet6 R[9];
    R[0] = jsub(one, jadd(tyy, tzz)); R[1] = jsub(txy, twz);            R[2] = jadd(txz, twy);
    R[3] = jadd(txy, twz);            R[4] = jsub(one, jadd(txx, tzz)); R[5] = jsub(tyz, twx);
    R[6] = jsub(txz, twy);            R[7] = jadd(tyz, twx);            R[8] = jsub(one, jadd(txx, tyy));
    /* inverse: linear = R^T, translation = -(R^T t) */
    jet6 Ri[9], ti[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ri[3 * r + c] = R[3 * c + r];
    for (int r = 0; r < 3; ++r)
        ti[r] = jneg(jadd(jadd(jmul(Ri[3 * r], pose[0]), jmul(Ri[3 * r + 1], pose[1])), jmul(Ri[3 * r + 2], pose[2])));
    /* Twc = inverse * Tcr */
    jet6 L[9], T[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            L[3 * r + c] = jadd(jadd(jscale(Ri[3 * r], Tcr[c]), jscale(Ri[3 * r + 1], Tcr[4 + c])), jscale(Ri[3 * r + 2], Tcr[8 + c]));
        T[r] = jadd(jadd(jadd(jscale(Ri[3 * r], Tcr[3]), jscale(Ri[3 * r + 1], Tcr[7])), jscale(Ri[3 * r + 2], Tcr[11])), ti[r]);
    }
    jet6 Po[2];
    for (int r = 0; r < 2; ++r)
        Po[r] = jadd(jadd(jadd(jscale(L[3 * r], P[0]), jscale(L[3 * r + 1], P[1])), jscale(L[3 * r + 2], P[2])), T[r]);
    /* (max - Po) / resolution - 0.5 + kPadding, TypeOccupiedSpace2D.h:115-118 */
    const double pad = (double)ORACLE_K_PADDING;
    jet6 dx = jsub(jc(g->max_x), Po[0]), dy = jsub(jc(g->max_y), Po[1]);
    jet6 qx, qy;
    qx.a = dx.a / g->resolution; qy.a = dy.a / g->resolution;            /* Jet / double divides value and partials */
    for (int i = 0; i < 6; ++i) { qx.v[i] = dx.v[i] / g->resolution; qy.v[i] = dy.v[i] / g->resolution; }
    *rr = jadd(jsub(qx, jc(0.5)), jc(pad));
    *cc = jadd(jsub(qy, jc(0.5)), jc(pad));
}

/* computeError (TypeOccupiedSpace2D.h:126-131) and linearizeOplus (:145-179).
 * The reference differentiates with ceres::internal::AutoDifferentiate over StaticParameterDims<6, 3>: the pose block
 * holds SIX jets (t1 t2 t3 qx qy qz) and the functor's `pose[6]` therefore reads the next jet in the array — the first
 * coordinate of the range point.  The Jacobian the reference uses is that of the functor with q.w := point.x (no
 * normalisation), taken w.r.t. (t, qx, qy, qz) and fed to the 6-dof oplus as is.  Reproduced here; the error itself
 * (computeError) uses the true 7-vector. */
static void laser_edge_impl(const double tq[7], const double Tcr[12], const double P[3], const visfs_ba_grid* g, double* e, double J[6], int true_w);
void oracle_laser_edge(const double tq[7], const double Tcr[12], const double P[3], const visfs_ba_grid* g, double* e, double J[6]) {
    laser_edge_impl(tq, Tcr, P, g, e, J, 0);
}
/* true_w: the Ceres factor (OccupiedSpace2dFactor.cpp:22-49, :93-97): AutoDiffCostFunction<..., DYNAMIC, 7> differentiates the same
 * functor over the FULL pose, q.w included, and PoseLocalParameterization's [I6; 0] Jacobian then drops the q.w column — the first six
 * partials with the pose's own q.w, no aliasing. */
static void laser_edge_impl(const double tq[7], const double Tcr[12], const double P[3], const visfs_ba_grid* g, double* e, double J[6], int true_w) {
    jet6 pose[7], r, c;
    if (e) {
        for (int i = 0; i < 7; ++i) pose[i] = jc(tq[i]);
        laser_functor(pose, P, Tcr, g, &r, &c);
        oracle_bicubic(g, r.a, c.a, e, NULL, NULL);
    }
    if (J) {
        for (int i = 0; i < 6; ++i) { pose[i] = jc(tq[i]); pose[i].v[i] = 1.0; }
        pose[6] = jc(true_w ? tq[6] : P[0]);                              /* the aliasing described above (g2o edge only) */
        laser_functor(pose, P, Tcr, g, &r, &c);
        double f, dfdr, dfdc;
        oracle_bicubic(g, r.a, c.a, &f, &dfdr, &dfdc);
        for (int i = 0; i < 6; ++i) J[i] = dfdr * r.v[i] + dfdc * c.v[i];  /* BiCubicInterpolator::Evaluate(JetT) */
    }
}

void oracle_huber(double e2, double delta, double rho[2]) {
    /* [g2o-upstream] RobustKernelHuber::robustify */
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.0; }
    else { const double sqrte = sqrt(e2); rho[0] = 2 * sqrte * delta - dsqr; rho[1] = delta / sqrte; }
}

/* [ceres-upstream] HuberLoss(a)::Evaluate on s = ||residual||^2 (loss_function.cc), and the Corrector for rho'' <= 0 (corrector.cc):
 * residual and Jacobian are scaled by sqrt(rho'), the cost is rho / 2.  Same function of s as g2o's kernel for a > 0; unlike the
 * g2o branch the Ceres branch attaches the loss whatever robustKernelDelta is (Optimizer.cpp:370,469), so a <= 0 is restated too. */
static void ceres_huber(double s, double a, double rho[2]) {
    const double b = a * a;
    if (s > b) { const double r = sqrt(s); rho[0] = 2.0 * a * r - b; rho[1] = fmax(DBL_MIN, a / r); }
    else { rho[0] = s; rho[1] = 1.0; }
}

/* ===================================================================== */
/* graph build / write-back                                               */
/* ===================================================================== */

static int find_id(const uint64_t* ids, int n, uint64_t id) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (ids[mid] == id) return mid;
        if (ids[mid] < id) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

static void iso_to_tq(const double T[12], double tq[7]) {
    const double R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
    const double t[3] = { T[3], T[7], T[11] };
    oracle_pose_from_Rt(R, t, tq);
}

int oracle_pack_window(const visfs_ba_params* params, const visfs_ba_window* w,
                       double* pose_tq, uint8_t* pose_fixed, uint8_t* point_used,
                       int32_t* obs_point, int32_t* obs_pose, double* obs_uvr, int32_t* obs_ref,
                       int32_t* odo_from, int32_t* odo_to, double* odo_tq,
                       visfs_ba_graph* g, int32_t* n_mono_skipped) {
    (void)params;
    memset(g, 0, sizeof(*g));
    /* poses: Optimizer.cpp:100-114 */
    for (int i = 0; i < w->n_poses; ++i) {
        double Twc[12], Tcw[12];
        iso_mul(w->pose_Twr + 12 * i, w->Trc, Twc);
        iso_inv(Twc, Tcw);
        iso_to_tq(Tcw, pose_tq + 7 * i);
        pose_fixed[i] = (w->pose_ids[i] == w->root_id);
    }
    /* links: Optimizer.cpp:123-150 */
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
        iso_to_tq(T2, odo_tq + 7 * ne);           /* g2o::SE3Quat(R,t) normalises like CameraPose */
        odo_from[ne] = a; odo_to[ne] = b;
        ++ne;
    }
    /* landmarks + stereo edges: Optimizer.cpp:153-223 */
    memset(point_used, 0, (size_t)w->n_points);
    int no = 0, mono = 0;
    for (int k = 0; k < w->n_refs; ++k) {
        const int p = find_id(w->point_ids, w->n_points, w->ref_feature[k]);
        if (p < 0) continue;                                             /* :158 */
        point_used[p] = 1;
        const int c = find_id(w->pose_ids, w->n_poses, w->ref_pose[k]);
        if (c < 0 || w->ref_pose[k] == 0) continue;                      /* :172 */
        const double depth = (double)w->ref_depth[k];                    /* :174 */
        double baseLine = 0.0;
        if (w->n_cameras > 1) baseLine = (double)w->baseline;            /* :181-183 */
        if (isfinite(depth) && depth > 0.0 && baseLine > 0.0) {
            const float disparity = (float)(baseLine * w->fx / depth);   /* :187 */
            obs_uvr[3 * no + 0] = (double)w->ref_u[k];
            obs_uvr[3 * no + 1] = (double)w->ref_v[k];
            obs_uvr[3 * no + 2] = (double)(w->ref_u[k] - disparity);     /* float - float, :188 */
            obs_point[no] = p; obs_pose[no] = c;
            if (obs_ref) obs_ref[no] = k;
            ++no;
        } else {
            ++mono;   /* reference dereferences an uninitialised pointer here (:179,:197-210): skipped */
        }
    }
    if (n_mono_skipped) *n_mono_skipped = mono;
    g->n_poses = w->n_poses; g->n_points = w->n_points; g->n_obs = no; g->n_odo = ne;
    g->pose_tq = pose_tq; g->pose_fixed = pose_fixed;
    g->point_xyz = w->point_xyz; g->point_fixed = w->point_fixed;
    g->obs_point = obs_point; g->obs_pose = obs_pose; g->obs_uvr = obs_uvr;
    g->odo_from = odo_from; g->odo_to = odo_to; g->odo_tq = odo_tq;
    g->fx = w->fx; g->fy = w->fy; g->cx = w->cx; g->cy = w->cy;
    g->bf = ((w->n_cameras > 1) ? (double)w->baseline : 0.0) * w->fx;    /* :195 */
    /* range points: Optimizer.cpp:225-258 — every point of every cloud hangs off the newest pose */
    if (w->n_laser_points > 0 && w->grid != NULL && w->laser_xyz != NULL) {
        g->n_laser = w->n_laser_points; g->laser_pose = w->n_poses - 1;
        g->laser_xyz = w->laser_xyz; g->grid = w->grid;
    }
    memcpy(g->Tcr, Tcr, 96);
    return VISFS_BA_OK;
}

void oracle_unpack_pose(const double* tq, const double* Trc, double* Twr) {
    /* Optimizer.cpp:324-329 */
    double R[9], t[3];
    oracle_pose_to_Rt(tq, R, t);
    const double Tcw[12] = { R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2] };
    double Twc[12], Tcr[12];
    iso_inv(Tcw, Twc);
    iso_inv(Trc, Tcr);
    iso_mul(Twc, Tcr, Twr);
}

/* ===================================================================== */
/* the solver: [g2o-upstream] SparseOptimizer + BlockSolver_6_3 + LM       */
/* ===================================================================== */

struct oracle_sys {
    visfs_ba_params prm;
    int Np, Nl, No, Ne, npf, n6, nthreads;
    double intr[5];
    /* graph (owned copies) */
    double *pose0, *pt0;
    uint8_t *pose_fixed, *pt_fixed;
    int32_t *obs_pt, *obs_pose, *odo_i, *odo_j;
    double *obs_uvr, *odo_tq;
    /* laser occupied-space edges (owned copies) */
    int Nz, laser_pose;
    double *laser_xyz, Tcr[12];
    visfs_ba_grid grid;
    float *grid_cost;
    int *pose_idx;              /* free index or -1 (hessianIndex) */
    int *lm_ptr;                /* CSR over obs by landmark */
    int *po_ptr, *po_obs;       /* CSR over obs by FREE pose, ascending edge id (the OpenMP passes walk a pose's edges in the serial order) */
    /* estimates */
    double *pose, *pt, *pose_trial, *pt_trial;
    uint8_t *obs_level;         /* 0 active level, 1 outlier */
    uint8_t *obs_edge_ok;       /* !allVerticesFixed */
    /* linearisation products */
    double *err, *chi2, *wgt, *W, *Hll, *bl, *Hpp, *bp;
    double *odo_err;
    /* per trial */
    double *Dinv, *S, *bs, *dxp, *dxl, *chol;
    double lambda_used;
    /* Ceres flavour (Optimizer/Framework=1): weights of the stereo / laser terms (1/var for g2o's e^T Omega e, 1/var^2 for
     * Ceres' ||Omega e||^2), and the per-variable multipliers of the damping: H_ii + lambda m_i (m = 1 for g2o) */
    int ceres;
    double w_px, w_laser;
    double *ml, *mp;            /* [3 Nl], [n6] */
    double *s2l, *s2p;          /* Jacobi scaling squared, fixed at iteration zero [ceres-upstream] */
    double pcg_residual;        /* LinearSolverPCG::_residual */
    double *pcg_r, *pcg_d, *pcg_q, *pcg_s, *pcg_J;
    /* outputs */
    double *final_chi2;
    uint8_t *outlier;
};

static void* xcalloc(size_t n, size_t sz) { void* p = calloc(n ? n : 1, sz); if (!p) abort(); return p; }

oracle_sys* oracle_sys_create(const visfs_ba_params* prm, const visfs_ba_graph* g, int nthreads) {
    oracle_sys* s = (oracle_sys*)xcalloc(1, sizeof(*s));
    s->prm = *prm;
    s->ceres = (prm->framework == 1);
    /* Optimizer.cpp:405-422: the Ceres branch never adds a wheel-odometry factor (links between two window poses fall in the
     * "TODO" arm, and the other arm needs both ids in the window) */
    s->Np = g->n_poses; s->Nl = g->n_points; s->No = g->n_obs; s->Ne = s->ceres ? 0 : g->n_odo;
    s->nthreads = nthreads < 1 ? 1 : nthreads;
    /* g2o: chi2 = e^T Omega e with Omega = I/var (Optimizer.cpp:153); Ceres: residual = info * e with the SAME matrix, so the
     * squared norm carries 1/var^2 (StereoObservationFactor.cpp:25-26; OccupiedSpace2dFactor.cpp:45) */
    s->w_px = 1.0 / prm->pixel_variance; s->w_laser = 1.0 / prm->laser_covariance;
    if (s->ceres) { s->w_px *= s->w_px; s->w_laser *= s->w_laser; }
    s->intr[0] = g->fx; s->intr[1] = g->fy; s->intr[2] = g->cx; s->intr[3] = g->cy; s->intr[4] = g->bf;
    const int Np = s->Np, Nl = s->Nl, No = s->No, Ne = s->Ne;
    s->pose0 = xcalloc((size_t)Np * 7, 8); memcpy(s->pose0, g->pose_tq, (size_t)Np * 56);
    s->pt0 = xcalloc((size_t)Nl * 3, 8);   memcpy(s->pt0, g->point_xyz, (size_t)Nl * 24);
    s->pose_fixed = xcalloc(Np, 1); memcpy(s->pose_fixed, g->pose_fixed, Np);
    s->pt_fixed = xcalloc(Nl, 1);   memcpy(s->pt_fixed, g->point_fixed, Nl);
    s->obs_pt = xcalloc(No, 4);   memcpy(s->obs_pt, g->obs_point, (size_t)No * 4);
    s->obs_pose = xcalloc(No, 4); memcpy(s->obs_pose, g->obs_pose, (size_t)No * 4);
    s->obs_uvr = xcalloc((size_t)No * 3, 8); memcpy(s->obs_uvr, g->obs_uvr, (size_t)No * 24);
    s->odo_i = xcalloc(Ne, 4); s->odo_j = xcalloc(Ne, 4); s->odo_tq = xcalloc((size_t)Ne * 7, 8);
    if (Ne) { memcpy(s->odo_i, g->odo_from, (size_t)Ne * 4); memcpy(s->odo_j, g->odo_to, (size_t)Ne * 4); memcpy(s->odo_tq, g->odo_tq, (size_t)Ne * 56); }
    /* laser edges are active only if their pose is free (allVerticesFixed otherwise: the range points are fixed) */
    s->Nz = 0;
    if (g->n_laser > 0 && g->grid && g->laser_pose >= 0 && g->laser_pose < Np && !g->pose_fixed[g->laser_pose]) {
        s->Nz = g->n_laser; s->laser_pose = g->laser_pose;
        s->laser_xyz = xcalloc((size_t)s->Nz * 3, 8); memcpy(s->laser_xyz, g->laser_xyz, (size_t)s->Nz * 24);
        memcpy(s->Tcr, g->Tcr, 96);
        s->grid = *g->grid;
        const size_t cells = (size_t)g->grid->num_x_cells * g->grid->num_y_cells;
        s->grid_cost = xcalloc(cells, 4); memcpy(s->grid_cost, g->grid->correspondence_cost, cells * 4);
        s->grid.correspondence_cost = s->grid_cost;
    }
    /* buildIndexMapping: non-fixed poses in id (= index) order */
    s->pose_idx = xcalloc(Np, sizeof(int));
    int npf = 0;
    for (int i = 0; i < Np; ++i) s->pose_idx[i] = s->pose_fixed[i] ? -1 : npf++;
    s->npf = npf; s->n6 = 6 * npf;
    s->lm_ptr = xcalloc((size_t)Nl + 1, sizeof(int));
    for (int k = 0; k < No; ++k) s->lm_ptr[s->obs_pt[k] + 1]++;
    for (int l = 0; l < Nl; ++l) s->lm_ptr[l + 1] += s->lm_ptr[l];
    s->po_ptr = xcalloc((size_t)npf + 1, sizeof(int)); s->po_obs = xcalloc(No, sizeof(int));
    for (int k = 0; k < No; ++k) { const int a = s->pose_idx[s->obs_pose[k]]; if (a >= 0) s->po_ptr[a + 1]++; }
    for (int a = 0; a < npf; ++a) s->po_ptr[a + 1] += s->po_ptr[a];
    { int* fill = xcalloc((size_t)npf + 1, sizeof(int)); memcpy(fill, s->po_ptr, ((size_t)npf + 1) * sizeof(int));
      for (int k = 0; k < No; ++k) { const int a = s->pose_idx[s->obs_pose[k]]; if (a >= 0) s->po_obs[fill[a]++] = k; }
      free(fill); }
    s->pose = xcalloc((size_t)Np * 7, 8); s->pt = xcalloc((size_t)Nl * 3, 8);
    s->pose_trial = xcalloc((size_t)Np * 7, 8); s->pt_trial = xcalloc((size_t)Nl * 3, 8);
    s->obs_level = xcalloc(No, 1); s->obs_edge_ok = xcalloc(No, 1);
    for (int k = 0; k < No; ++k) s->obs_edge_ok[k] = !(s->pose_fixed[s->obs_pose[k]] && s->pt_fixed[s->obs_pt[k]]);
    s->err = xcalloc((size_t)No * 3, 8); s->chi2 = xcalloc(No, 8); s->wgt = xcalloc(No, 8); s->W = xcalloc((size_t)No * 18, 8);
    s->Hll = xcalloc((size_t)Nl * 6, 8); s->bl = xcalloc((size_t)Nl * 3, 8);
    s->Hpp = xcalloc((size_t)s->n6 * s->n6, 8); s->bp = xcalloc(s->n6, 8);
    s->odo_err = xcalloc((size_t)Ne * 6, 8);
    s->Dinv = xcalloc((size_t)Nl * 6, 8);
    s->ml = xcalloc((size_t)Nl * 3, 8); s->s2l = xcalloc((size_t)Nl * 3, 8);
    s->mp = xcalloc(s->n6, 8); s->s2p = xcalloc(s->n6, 8);
    for (int i = 0; i < 3 * Nl; ++i) s->ml[i] = 1.0;
    for (int i = 0; i < s->n6; ++i) s->mp[i] = 1.0;
    s->S = xcalloc((size_t)s->n6 * s->n6, 8); s->bs = xcalloc(s->n6, 8); s->chol = xcalloc((size_t)s->n6 * s->n6, 8);
    s->dxp = xcalloc(s->n6, 8); s->dxl = xcalloc((size_t)Nl * 3, 8);
    s->pcg_r = xcalloc(s->n6, 8); s->pcg_d = xcalloc(s->n6, 8); s->pcg_q = xcalloc(s->n6, 8); s->pcg_s = xcalloc(s->n6, 8);
    s->pcg_J = xcalloc((size_t)npf * 36, 8);
    s->final_chi2 = xcalloc(No, 8); s->outlier = xcalloc(No, 1);
    oracle_sys_reset(s);
    return s;
}

void oracle_sys_reset(oracle_sys* s) {
    memcpy(s->pose, s->pose0, (size_t)s->Np * 56);
    memcpy(s->pt, s->pt0, (size_t)s->Nl * 24);
    memset(s->obs_level, 0, s->No);
    memset(s->outlier, 0, s->No);
    memset(s->final_chi2, 0, (size_t)s->No * 8);
    s->pcg_residual = -1.0;
}

void oracle_sys_destroy(oracle_sys* s) {
    if (!s) return;
    free(s->pose0); free(s->pt0); free(s->pose_fixed); free(s->pt_fixed); free(s->obs_pt); free(s->obs_pose);
    free(s->obs_uvr); free(s->odo_i); free(s->odo_j); free(s->odo_tq); free(s->pose_idx); free(s->lm_ptr); free(s->po_ptr); free(s->po_obs);
    free(s->pose); free(s->pt); free(s->pose_trial); free(s->pt_trial); free(s->obs_level); free(s->obs_edge_ok);
    free(s->err); free(s->chi2); free(s->wgt); free(s->W); free(s->Hll); free(s->bl); free(s->Hpp); free(s->bp);
    free(s->odo_err); free(s->Dinv); free(s->S); free(s->bs); free(s->chol); free(s->dxp); free(s->dxl);
    free(s->pcg_r); free(s->pcg_d); free(s->pcg_q); free(s->pcg_s); free(s->pcg_J); free(s->final_chi2); free(s->outlier);
    free(s->laser_xyz); free(s->grid_cost); free(s->ml); free(s->mp); free(s->s2l); free(s->s2p);
    free(s);
}

int oracle_sys_free_poses(const oracle_sys* s) { return s->npf; }

static inline int edge_active(const oracle_sys* s, int k) { return s->obs_level[k] == 0 && s->obs_edge_ok[k]; }

/* computeActiveErrors + activeRobustChi2 at (pose, pt). Fills err/chi2 when store != 0. */
static double active_robust_chi2(oracle_sys* s, const double* pose, const double* pt, int store) {
    const double inv_var = s->w_px;                       /* Omega = I3 / pixelVariance, Optimizer.cpp:153 (squared for Ceres) */
    const double delta = s->prm.robust_kernel_delta;
    const int ceres = s->ceres;
    double total = 0.0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+:total) num_threads(s->nthreads) if (s->nthreads > 1)
#endif
    for (int k = 0; k < s->No; ++k) {
        if (!edge_active(s, k)) { if (store) { s->chi2[k] = 0.0; s->err[3*k] = s->err[3*k+1] = s->err[3*k+2] = 0.0; } continue; }
        double e[3];
        oracle_stereo_edge(pose + 7 * s->obs_pose[k], pt + 3 * s->obs_pt[k], s->obs_uvr + 3 * k, s->intr, e, NULL, NULL);
        /* chi2() = e . (Omega e) */
        const double c = e[0] * (inv_var * e[0]) + e[1] * (inv_var * e[1]) + e[2] * (inv_var * e[2]);
        if (store) { s->err[3*k] = e[0]; s->err[3*k+1] = e[1]; s->err[3*k+2] = e[2]; s->chi2[k] = c; }
        if (ceres) { double rho[2]; ceres_huber(c, delta, rho); total += rho[0]; }     /* the loss is attached unconditionally (:370,469) */
        else if (delta > 0.0) { double rho[2]; oracle_huber(c, delta, rho); total += rho[0]; }
        else total += c;
    }
    /* odometry edges: no robust kernel (Optimizer.cpp:135-142) */
    const double inv_cov = 1.0 / s->prm.odometry_covariance;   /* Optimizer.cpp:117-121 */
    for (int k = 0; k < s->Ne; ++k) {
        const int i = s->odo_i[k], j = s->odo_j[k];
        if (s->pose_fixed[i] && s->pose_fixed[j]) continue;      /* allVerticesFixed */
        double e[6];
        oracle_odo_edge(pose + 7 * i, pose + 7 * j, s->odo_tq + 7 * k, e, NULL, NULL);
        if (store) memcpy(s->odo_err + 6 * k, e, 48);
        double c = 0.0;
        for (int d = 0; d < 6; ++d) c += e[d] * (inv_cov * e[d]);
        total += c;
    }
    /* laser edges: information 1 / laserCovariance, no robust kernel (Optimizer.cpp:232,244-249) */
    const double inv_laser = s->w_laser;
    for (int k = 0; k < s->Nz; ++k) {
        double e;
        oracle_laser_edge(pose + 7 * s->laser_pose, s->Tcr, s->laser_xyz + 3 * k, &s->grid, &e, NULL);
        total += e * (inv_laser * e);
    }
    return total;
}

/* H(6x6 block at free poses a,b) += A^T * (w) * B where A is m x 6, B is m x 6 (row-major). */
static void hpp_add(oracle_sys* s, int a, int b, const double* A, const double* B, int m, double w) {
    double* H = s->Hpp + (size_t)(6 * a) * s->n6 + 6 * b;
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double acc = 0.0;
            for (int k = 0; k < m; ++k) acc += A[k * 6 + r] * w * B[k * 6 + c];
            H[(size_t)r * s->n6 + c] += acc;
        }
}

void oracle_sys_linearize(oracle_sys* s, double* robust_chi2, double* max_diag) {
    const double chi = active_robust_chi2(s, s->pose, s->pt, 1);
    if (robust_chi2) *robust_chi2 = chi;
    const double inv_var = s->w_px;
    const double delta = s->prm.robust_kernel_delta;
    memset(s->Hll, 0, (size_t)s->Nl * 48); memset(s->bl, 0, (size_t)s->Nl * 24);
    memset(s->Hpp, 0, (size_t)s->n6 * s->n6 * 8); memset(s->bp, 0, (size_t)s->n6 * 8);
    /* buildSystem: per edge linearizeOplus + constructQuadraticForm [g2o-upstream base_binary_edge.hpp] */
    /* landmark-major pass: W (Hpl), Hll, bl */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(s->nthreads) if (s->nthreads > 1)
#endif
    for (int l = 0; l < s->Nl; ++l) {
        for (int k = s->lm_ptr[l]; k < s->lm_ptr[l + 1]; ++k) {
            double* Wk = s->W + 18 * k;
            for (int q = 0; q < 18; ++q) Wk[q] = 0.0;
            s->wgt[k] = 0.0;
            if (!edge_active(s, k)) continue;
            const int ip = s->obs_pose[k];
            double e[3], Ji[9], Jj[18];
            oracle_stereo_edge(s->pose + 7 * ip, s->pt + 3 * l, s->obs_uvr + 3 * k, s->intr, e, Ji, Jj);
            double rho1 = 1.0;
            if (s->ceres) { double rho[2]; ceres_huber(s->chi2[k], delta, rho); rho1 = rho[1]; }
            else if (delta > 0.0) { double rho[2]; oracle_huber(s->chi2[k], delta, rho); rho1 = rho[1]; }
            s->wgt[k] = rho1;
            const double wo = rho1 * inv_var;                 /* weightedOmega = rho' * Omega (diagonal) */
            const int pfree = !s->pose_fixed[ip], lfree = !s->pt_fixed[l];
            if (lfree) {
                double* H = s->Hll + 6 * l; double* b = s->bl + 3 * l;
                /* Hll += Ji^T (wo) Ji ; b_l += Ji^T (-(rho' Omega e)) */
                int q = 0;
                for (int r = 0; r < 3; ++r)
                    for (int c = r; c < 3; ++c, ++q)
                        H[q] += Ji[r] * wo * Ji[c] + Ji[3 + r] * wo * Ji[3 + c] + Ji[6 + r] * wo * Ji[6 + c];
                for (int r = 0; r < 3; ++r)
                    b[r] += -(Ji[r] * wo * e[0] + Ji[3 + r] * wo * e[1] + Ji[6 + r] * wo * e[2]);
            }
            if (pfree && lfree) {
                /* Hpl (pose row, point col) = Jj^T (wo) Ji, 6x3 */
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 3; ++c)
                        Wk[r * 3 + c] = Jj[r] * wo * Ji[c] + Jj[6 + r] * wo * Ji[3 + c] + Jj[12 + r] * wo * Ji[6 + c];
            }
        }
    }
    /* pose pass: Hpp, bp.  Every pose block sums its edges in edge order; with several threads the poses are shared out and each
     * thread walks ITS poses' edges in that same order (pose-major lists), so the sums are bit-identical to the serial pass */
    if (s->nthreads > 1) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(s->nthreads)
#endif
        for (int a = 0; a < s->npf; ++a) {
            for (int n = s->po_ptr[a]; n < s->po_ptr[a + 1]; ++n) {
                const int k = s->po_obs[n];
                if (!edge_active(s, k)) continue;
                const int ip = s->obs_pose[k];
                double e[3], Jj[18];
                oracle_stereo_edge(s->pose + 7 * ip, s->pt + 3 * s->obs_pt[k], s->obs_uvr + 3 * k, s->intr, e, NULL, Jj);
                const double wo = s->wgt[k] * inv_var;
                hpp_add(s, a, a, Jj, Jj, 3, wo);
                double* b = s->bp + 6 * a;
                for (int r = 0; r < 6; ++r) b[r] += -(Jj[r] * wo * e[0] + Jj[6 + r] * wo * e[1] + Jj[12 + r] * wo * e[2]);
            }
        }
    } else
    for (int k = 0; k < s->No; ++k) {
        if (!edge_active(s, k)) continue;
        const int ip = s->obs_pose[k];
        if (s->pose_fixed[ip]) continue;
        const int a = s->pose_idx[ip];
        double e[3], Jj[18];
        oracle_stereo_edge(s->pose + 7 * ip, s->pt + 3 * s->obs_pt[k], s->obs_uvr + 3 * k, s->intr, e, NULL, Jj);
        const double wo = s->wgt[k] * inv_var;
        hpp_add(s, a, a, Jj, Jj, 3, wo);
        double* b = s->bp + 6 * a;
        for (int r = 0; r < 6; ++r) b[r] += -(Jj[r] * wo * e[0] + Jj[6 + r] * wo * e[1] + Jj[12 + r] * wo * e[2]);
    }
    /* odometry edges (EdgePoseConstraint, no kernel): vertex 0 = from (Xi), vertex 1 = to (Xj) */
    const double inv_cov = 1.0 / s->prm.odometry_covariance;
    for (int k = 0; k < s->Ne; ++k) {
        const int i = s->odo_i[k], j = s->odo_j[k];
        const int fi = !s->pose_fixed[i], fj = !s->pose_fixed[j];
        if (!fi && !fj) continue;
        double e[6], Ji[36], Jj[36];
        oracle_odo_edge(s->pose + 7 * i, s->pose + 7 * j, s->odo_tq + 7 * k, e, Ji, Jj);
        const int a = s->pose_idx[i], b = s->pose_idx[j];
        if (fi) {
            hpp_add(s, a, a, Ji, Ji, 6, inv_cov);
            for (int r = 0; r < 6; ++r) { double acc = 0.0; for (int d = 0; d < 6; ++d) acc += Ji[d * 6 + r] * inv_cov * e[d]; s->bp[6 * a + r] -= acc; }
        }
        if (fj) {
            hpp_add(s, b, b, Jj, Jj, 6, inv_cov);
            for (int r = 0; r < 6; ++r) { double acc = 0.0; for (int d = 0; d < 6; ++d) acc += Jj[d * 6 + r] * inv_cov * e[d]; s->bp[6 * b + r] -= acc; }
        }
        if (fi && fj) {
            hpp_add(s, a, b, Ji, Jj, 6, inv_cov);
            hpp_add(s, b, a, Jj, Ji, 6, inv_cov);
        }
    }
    /* laser edges: the range point is fixed, only the pose block receives J^T Omega J and -J^T Omega e */
    const double inv_laser = s->w_laser;
    for (int k = 0; k < s->Nz; ++k) {
        double e, J[6];
        laser_edge_impl(s->pose + 7 * s->laser_pose, s->Tcr, s->laser_xyz + 3 * k, &s->grid, &e, J, s->ceres);
        const int a = s->pose_idx[s->laser_pose];
        hpp_add(s, a, a, J, J, 1, inv_laser);
        for (int r = 0; r < 6; ++r) s->bp[6 * a + r] -= J[r] * inv_laser * e;
    }
    /* computeLambdaInit's maxDiagonal over pose and landmark diagonals [g2o-upstream] */
    double md = 0.0;
    for (int i = 0; i < s->n6; ++i) { const double v = fabs(s->Hpp[(size_t)i * s->n6 + i]); if (v > md) md = v; }
    for (int l = 0; l < s->Nl; ++l) {
        if (s->pt_fixed[l]) continue;
        const double* H = s->Hll + 6 * l;
        if (fabs(H[0]) > md) md = fabs(H[0]);
        if (fabs(H[3]) > md) md = fabs(H[3]);
        if (fabs(H[5]) > md) md = fabs(H[5]);
    }
    if (max_diag) *max_diag = md;
}

/* symmetric 3x3 inverse from 6 unique entries (xx xy xz yy yz zz) via cofactors (Eigen's 3x3 inverse). */
static void sym3_inv(const double h[6], double o[6]) {
    const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id; o[5] = (a * d - b * b) * id;
}

static int landmark_has_active_edge(const oracle_sys* s, int l) {
    for (int k = s->lm_ptr[l]; k < s->lm_ptr[l + 1]; ++k) if (edge_active(s, k)) return 1;
    return 0;
}

/* dense Cholesky (lower, row-major) of the n x n SPD matrix A into L; returns 0 on failure */
static int cholesky(const double* A, double* L, int n) {
    for (int i = 0; i < n; ++i) {
        double* Li = L + (size_t)i * n;
        for (int j = 0; j <= i; ++j) {
            const double* Lj = L + (size_t)j * n;
            double sum = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) sum -= Li[k] * Lj[k];
            if (i == j) {
                if (!(sum > 0.0) || !isfinite(sum)) return 0;
                Li[i] = sqrt(sum);
            } else {
                Li[j] = sum / Lj[j];
            }
        }
    }
    return 1;
}
static void cholesky_solve(const double* L, int n, const double* b, double* x) {
    for (int i = 0; i < n; ++i) {
        double sum = b[i];
        const double* Li = L + (size_t)i * n;
        for (int k = 0; k < i; ++k) sum -= Li[k] * x[k];
        x[i] = sum / Li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double sum = x[i];
        for (int k = i + 1; k < n; ++k) sum -= L[(size_t)k * n + i] * x[k];
        x[i] = sum / L[(size_t)i * n + i];
    }
}

/* general 6x6 inverse by Gauss-Jordan with partial pivoting (Eigen: PartialPivLU inverse for n>4) */
static void inv6(const double* A, int lda, double* out) {
    double M[6][12];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { M[r][c] = A[(size_t)r * lda + c]; M[r][6 + c] = (r == c); }
    for (int k = 0; k < 6; ++k) {
        int p = k; double best = fabs(M[k][k]);
        for (int r = k + 1; r < 6; ++r) if (fabs(M[r][k]) > best) { best = fabs(M[r][k]); p = r; }
        if (p != k) for (int c = 0; c < 12; ++c) { double t = M[k][c]; M[k][c] = M[p][c]; M[p][c] = t; }
        const double piv = 1.0 / M[k][k];
        for (int c = 0; c < 12; ++c) M[k][c] *= piv;
        for (int r = 0; r < 6; ++r) if (r != k) { const double f = M[r][k]; if (f != 0.0) for (int c = 0; c < 12; ++c) M[r][c] -= f * M[k][c]; }
    }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out[r * 6 + c] = M[r][6 + c];
}

static void sym_matvec(const double* A, int n, const double* x, double* y, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) if (nthreads > 1 && n > 256)
#endif
    for (int i = 0; i < n; ++i) {
        const double* Ai = A + (size_t)i * n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += Ai[k] * x[k];
        y[i] = acc;
    }
}

/* [g2o-upstream] LinearSolverPCG::solve: block-Jacobi preconditioned CG, tolerance 1e-6, maxIter = rows,
 * absolute tolerance carried over in _residual between solves of one optimize() call. */
static int pcg_solve(oracle_sys* s, int* iters) {
    const int n = s->n6, nb = s->npf;
    double *r = s->pcg_r, *d = s->pcg_d, *q = s->pcg_q, *sv = s->pcg_s, *x = s->dxp;
    for (int b = 0; b < nb; ++b) inv6(s->S + (size_t)(6 * b) * n + 6 * b, n, s->pcg_J + 36 * b);
    memset(x, 0, (size_t)n * 8);
    memcpy(r, s->bs, (size_t)n * 8);
    for (int b = 0; b < nb; ++b) for (int rr = 0; rr < 6; ++rr) { double a = 0.0; for (int c = 0; c < 6; ++c) a += s->pcg_J[36 * b + rr * 6 + c] * r[6 * b + c]; d[6 * b + rr] = a; }
    double dn = 0.0; for (int i = 0; i < n; ++i) dn += r[i] * d[i];
    double d0 = 1e-6 * dn;
    if (s->pcg_residual > 0.0 && s->pcg_residual > d0) d0 = s->pcg_residual;
    const int maxIter = n;
    int it;
    for (it = 0; it < maxIter; ++it) {
        if (dn <= d0) break;
        sym_matvec(s->S, n, d, q, s->nthreads);
        double dq = 0.0; for (int i = 0; i < n; ++i) dq += d[i] * q[i];
        const double a = dn / dq;
        for (int i = 0; i < n; ++i) x[i] += a * d[i];
        for (int i = 0; i < n; ++i) r[i] -= a * q[i];
        for (int b = 0; b < nb; ++b) for (int rr = 0; rr < 6; ++rr) { double acc = 0.0; for (int c = 0; c < 6; ++c) acc += s->pcg_J[36 * b + rr * 6 + c] * r[6 * b + c]; sv[6 * b + rr] = acc; }
        const double dold = dn;
        dn = 0.0; for (int i = 0; i < n; ++i) dn += r[i] * sv[i];
        const double ba = dn / dold;
        for (int i = 0; i < n; ++i) d[i] = sv[i] + ba * d[i];
    }
    s->pcg_residual = 0.5 * dn;
    *iters = it;
    return 1;
}

/* [g2o-upstream] BlockSolver::setLambda + solve (Schur) */
static int schur_solve(oracle_sys* s, double lambda, int* pcg_iters) {
    const int n = s->n6, Nl = s->Nl;
    *pcg_iters = 0;
    s->lambda_used = lambda;
    /* Hschur = Hpp (+lambda on the diagonal); bschur = bp - coefficients */
    memcpy(s->S, s->Hpp, (size_t)n * n * 8);
    for (int i = 0; i < n; ++i) s->S[(size_t)i * n + i] += lambda * s->mp[i];      /* mp = 1 (g2o: lambda I) */
    memcpy(s->bs, s->bp, (size_t)n * 8);
    /* poses without any active edge are not part of g2o's active set: pin them (dx = 0) */
    for (int b = 0; b < s->npf; ++b) {
        int empty = 1;
        for (int r = 0; r < 6 && empty; ++r) if (s->Hpp[(size_t)(6 * b + r) * n + 6 * b + r] != 0.0) empty = 0;
        if (empty) for (int r = 0; r < 6; ++r) s->S[(size_t)(6 * b + r) * n + 6 * b + r] = 1.0;
    }
    const int nt = s->nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (int l = 0; l < Nl; ++l) {
        double* Di = s->Dinv + 6 * l;
        if (s->pt_fixed[l] || !landmark_has_active_edge(s, l)) { memset(Di, 0, 48); continue; }
        double h[6]; memcpy(h, s->Hll + 6 * l, 48);
        h[0] += lambda * s->ml[3 * l]; h[3] += lambda * s->ml[3 * l + 1]; h[5] += lambda * s->ml[3 * l + 2];
        sym3_inv(h, Di);
    }
    if (nt > 1) {
        /* block ROW i1 of the upper triangle and b_s[i1] belong to one thread, which walks pose i1's edges in edge (= landmark)
         * order: every block receives its landmarks' terms in the order of the serial loop below — bit-identical, no per-thread
         * copies of S to merge */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#endif
        for (int i1 = 0; i1 < s->npf; ++i1) {
            for (int n1 = s->po_ptr[i1]; n1 < s->po_ptr[i1 + 1]; ++n1) {
                const int k1 = s->po_obs[n1];
                const int l = s->obs_pt[k1];
                if (s->pt_fixed[l] || s->wgt[k1] == 0.0) continue;
                const double* Di = s->Dinv + 6 * l;
                const double D[9] = { Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5] };
                double Ddb[3]; mat3_vec(D, s->bl + 3 * l, Ddb);
                const double* B1 = s->W + 18 * k1;
                double BD[18];
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[r * 3 + c] = B1[r * 3] * D[c] + B1[r * 3 + 1] * D[3 + c] + B1[r * 3 + 2] * D[6 + c];
                for (int r = 0; r < 6; ++r) s->bs[6 * i1 + r] -= B1[r * 3] * Ddb[0] + B1[r * 3 + 1] * Ddb[1] + B1[r * 3 + 2] * Ddb[2];
                for (int k2 = s->lm_ptr[l]; k2 < s->lm_ptr[l + 1]; ++k2) {
                    const int i2 = s->pose_idx[s->obs_pose[k2]];
                    if (i2 < i1 || s->wgt[k2] == 0.0) continue;
                    if (i2 == i1 && k2 != k1) continue;
                    const double* B2 = s->W + 18 * k2;
                    double* H = s->S + (size_t)(6 * i1) * n + 6 * i2;
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 6; ++c)
                            H[(size_t)r * n + c] -= BD[r * 3] * B2[c * 3] + BD[r * 3 + 1] * B2[c * 3 + 1] + BD[r * 3 + 2] * B2[c * 3 + 2];
                }
            }
        }
    } else
    for (int l = 0; l < Nl; ++l) {
        if (s->pt_fixed[l]) continue;
        const double* Di = s->Dinv + 6 * l;
        const double D[9] = { Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5] };
        const double* db = s->bl + 3 * l;
        double Ddb[3]; mat3_vec(D, db, Ddb);
        for (int k1 = s->lm_ptr[l]; k1 < s->lm_ptr[l + 1]; ++k1) {
            const int i1 = s->pose_idx[s->obs_pose[k1]];
            if (i1 < 0 || s->wgt[k1] == 0.0) continue;
            const double* B1 = s->W + 18 * k1;
            double BD[18];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[r * 3 + c] = B1[r * 3] * D[c] + B1[r * 3 + 1] * D[3 + c] + B1[r * 3 + 2] * D[6 + c];
            for (int r = 0; r < 6; ++r) s->bs[6 * i1 + r] -= B1[r * 3] * Ddb[0] + B1[r * 3 + 1] * Ddb[1] + B1[r * 3 + 2] * Ddb[2];
            for (int k2 = s->lm_ptr[l]; k2 < s->lm_ptr[l + 1]; ++k2) {
                const int i2 = s->pose_idx[s->obs_pose[k2]];
                if (i2 < i1 || s->wgt[k2] == 0.0) continue;      /* upper triangle only */
                if (i2 == i1 && k2 != k1) continue;               /* one edge per (pose, landmark) pair */
                const double* B2 = s->W + 18 * k2;
                double* H = s->S + (size_t)(6 * i1) * n + 6 * i2;
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c)
                        H[(size_t)r * n + c] -= BD[r * 3] * B2[c * 3] + BD[r * 3 + 1] * B2[c * 3 + 1] + BD[r * 3 + 2] * B2[c * 3 + 2];
            }
        }
    }
    /* mirror the upper block triangle */
    for (int i = 0; i < n; ++i) for (int j = (i / 6 + 1) * 6; j < n; ++j) s->S[(size_t)j * n + i] = s->S[(size_t)i * n + j];
    int ok;
    if (s->prm.solver == 2) {
        ok = pcg_solve(s, pcg_iters);
    } else {
        ok = cholesky(s->S, s->chol, n);
        if (ok) cholesky_solve(s->chol, n, s->bs, s->dxp);
    }
    if (!ok) return 0;
    /* back-substitution: c_l = b_l - Hpl^T x_p ; x_l = Dinv c_l */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(s->nthreads) if (s->nthreads > 1)
#endif
    for (int l = 0; l < Nl; ++l) {
        double* xl = s->dxl + 3 * l;
        xl[0] = xl[1] = xl[2] = 0.0;
        if (s->pt_fixed[l] || !landmark_has_active_edge(s, l)) continue;
        double c[3] = { s->bl[3 * l], s->bl[3 * l + 1], s->bl[3 * l + 2] };
        for (int k = s->lm_ptr[l]; k < s->lm_ptr[l + 1]; ++k) {
            const int i = s->pose_idx[s->obs_pose[k]];
            if (i < 0 || s->wgt[k] == 0.0) continue;
            const double* B = s->W + 18 * k; const double* xp = s->dxp + 6 * i;
            for (int cc = 0; cc < 3; ++cc) { double acc = 0.0; for (int r = 0; r < 6; ++r) acc += B[r * 3 + cc] * xp[r]; c[cc] -= acc; }
        }
        const double* Di = s->Dinv + 6 * l;
        xl[0] = Di[0] * c[0] + Di[1] * c[1] + Di[2] * c[2];
        xl[1] = Di[1] * c[0] + Di[3] * c[1] + Di[4] * c[2];
        xl[2] = Di[2] * c[0] + Di[4] * c[1] + Di[5] * c[2];
    }
    return 1;
}

static void apply_update(oracle_sys* s) {
    /* SparseOptimizer::update → oplus on every non-fixed vertex (K8) */
    memcpy(s->pose_trial, s->pose, (size_t)s->Np * 56);
    memcpy(s->pt_trial, s->pt, (size_t)s->Nl * 24);
    for (int i = 0; i < s->Np; ++i) if (s->pose_idx[i] >= 0) oracle_pose_update(s->pose_trial + 7 * i, s->dxp + 6 * s->pose_idx[i]);
    for (int l = 0; l < s->Nl; ++l) if (!s->pt_fixed[l]) { s->pt_trial[3*l] += s->dxl[3*l]; s->pt_trial[3*l+1] += s->dxl[3*l+1]; s->pt_trial[3*l+2] += s->dxl[3*l+2]; }
}

static double compute_scale(const oracle_sys* s, double lambda) {
    /* [g2o-upstream] OptimizationAlgorithmLevenberg::computeScale: sum_j x_j (lambda x_j + b_j) */
    double scale = 0.0;
    for (int i = 0; i < s->n6; ++i) scale += s->dxp[i] * (lambda * s->mp[i] * s->dxp[i] + s->bp[i]);
    for (int l = 0; l < s->Nl; ++l) { if (s->pt_fixed[l]) continue; for (int c = 0; c < 3; ++c) scale += s->dxl[3*l+c] * (lambda * s->ml[3*l+c] * s->dxl[3*l+c] + s->bl[3*l+c]); }
    return scale;
}

void oracle_sys_trial(oracle_sys* s, double lambda, double* trial_chi2, double* scale, int32_t* pcg_iterations, int32_t* solver_ok) {
    int it = 0;
    const int ok = schur_solve(s, lambda, &it);
    if (pcg_iterations) *pcg_iterations = it;
    if (solver_ok) *solver_ok = ok;
    if (!ok) { if (trial_chi2) *trial_chi2 = DBL_MAX; if (scale) *scale = 0.0; return; }
    apply_update(s);
    if (scale) *scale = compute_scale(s, lambda);
    if (trial_chi2) *trial_chi2 = active_robust_chi2(s, s->pose_trial, s->pt_trial, 0);
}

int oracle_sys_fetch(oracle_sys* s, int32_t which, double* dst, size_t n) {
    const double* src = NULL; size_t m = 0;
    switch (which) {
        case VISFS_BA_BUF_OBS_ERR: src = s->err; m = (size_t)s->No * 3; break;
        case VISFS_BA_BUF_OBS_CHI2: src = s->chi2; m = s->No; break;
        case VISFS_BA_BUF_OBS_WEIGHT: src = s->wgt; m = s->No; break;
        case VISFS_BA_BUF_HPL: src = s->W; m = (size_t)s->No * 18; break;
        case VISFS_BA_BUF_HLL: src = s->Hll; m = (size_t)s->Nl * 6; break;
        case VISFS_BA_BUF_BL: src = s->bl; m = (size_t)s->Nl * 3; break;
        case VISFS_BA_BUF_HPP: src = s->Hpp; m = (size_t)s->n6 * s->n6; break;
        case VISFS_BA_BUF_BP: src = s->bp; m = s->n6; break;
        case VISFS_BA_BUF_S: src = s->S; m = (size_t)s->n6 * s->n6; break;
        case VISFS_BA_BUF_BS: src = s->bs; m = s->n6; break;
        case VISFS_BA_BUF_DX_POSE: src = s->dxp; m = s->n6; break;
        case VISFS_BA_BUF_DX_POINT: src = s->dxl; m = (size_t)s->Nl * 3; break;
        case VISFS_BA_BUF_POSE_TRIAL: src = s->pose_trial; m = (size_t)s->Np * 7; break;
        case VISFS_BA_BUF_POINT_TRIAL: src = s->pt_trial; m = (size_t)s->Nl * 3; break;
        default: return VISFS_BA_ERR_BAD_ARGUMENT;
    }
    if (n < m) return VISFS_BA_ERR_BAD_ARGUMENT;
    memcpy(dst, src, m * 8);
    return VISFS_BA_OK;
}

/* [g2o-upstream] SparseOptimizer::optimize(n) with OptimizationAlgorithmLevenberg / GaussNewton, written over an
 * abstract problem (linearise / damped trial / commit) so that the control flow exists ONCE: the solver drives it with the
 * real system, oracle_lm_script() with scripted trial outcomes (known-answer tests of the schedule, incl. its failure paths).
 * Returns the number of outer iterations executed. */
typedef struct {
    void (*begin)(void* ctx);                                                 /* algorithm->init(): LinearSolverPCG::init() */
    void (*linearize)(void* ctx, double* chi, double* max_diag);              /* computeActiveErrors + buildSystem */
    void (*trial)(void* ctx, double lambda, double* temp_chi, double* scale, int32_t* pcg_it, int32_t* ok);
    void (*commit)(void* ctx);                                                /* discardTop: the trial becomes the estimate */
    int gauss_newton;
} lm_problem;

static int lm_optimize_phase(const lm_problem* P, void* ctx, int n_iter, visfs_ba_stats* st, int phase) {
    P->begin(ctx);
    double lambda = 0.0, ni = 2.0;
    int done = 0;
    for (int it = 0; it < n_iter; ++it) {
        double currentChi, maxDiag;
        P->linearize(ctx, &currentChi, &maxDiag);
        if (it == 0 && phase == 0 && st) st->chi2_initial = currentChi;
        if (P->gauss_newton) {
            /* OptimizationAlgorithmGaussNewton::solve: lambda = 0, always take the step.  (g2o calls update(x) even when
             * solve() failed — x then holds whatever the failed solver left; restated as "no update") */
            int32_t pit = 0, ok = 0;
            double tc = 0.0, sc = 0.0;
            P->trial(ctx, 0.0, &tc, &sc, &pit, &ok);
            if (st) { st->pcg_iterations += pit; st->trials_run[phase]++; }
            if (ok) P->commit(ctx);
            ++done;
            if (st && st->n_trace < VISFS_BA_MAX_TRACE) { st->trace_lambda[st->n_trace] = 0.0; st->trace_chi2[st->n_trace] = currentChi; st->n_trace++; }
            if (!ok) break;                 /* Fail */
            continue;
        }
        if (it == 0) { lambda = 1e-5 * maxDiag; ni = 2.0; }      /* computeLambdaInit, tau = 1e-5 */
        double rho = 0.0, tempChi = currentChi;
        int qmax = 0;
        do {
            int32_t pit = 0, ok = 0;
            double scale = 0.0;
            P->trial(ctx, lambda, &tempChi, &scale, &pit, &ok);
            if (st) { st->pcg_iterations += pit; st->trials_run[phase]++; }
            if (!ok) { tempChi = DBL_MAX; scale = 0.0; }
            scale += 1e-3;
            rho = (currentChi - tempChi) / scale;
            if (rho > 0.0 && isfinite(tempChi)) {
                double alpha = 1.0 - pow(2.0 * rho - 1.0, 3.0);
                alpha = fmin(alpha, 2.0 / 3.0);
                const double scaleFactor = fmax(1.0 / 3.0, alpha);
                lambda *= scaleFactor;
                ni = 2.0;
                currentChi = tempChi;
                P->commit(ctx);                                        /* discardTop */
            } else {
                lambda *= ni;
                ni *= 2.0;                                             /* pop: estimates unchanged */
                if (!isfinite(lambda)) break;
            }
            ++qmax;
        } while (rho < 0.0 && qmax < 10);
        ++done;
        if (st && st->n_trace < VISFS_BA_MAX_TRACE) { st->trace_lambda[st->n_trace] = lambda; st->trace_chi2[st->n_trace] = currentChi; st->n_trace++; }
        if (qmax == 10 || rho == 0.0 || !isfinite(lambda)) break;     /* Terminate (a non-finite lambda left the trial loop above) */
    }
    return done;
}

/* the real system behind lm_problem */
static void sys_begin(void* c) { ((oracle_sys*)c)->pcg_residual = -1.0; }   /* LinearSolverPCG::init() at algorithm->init() */
static void sys_linearize(void* c, double* chi, double* md) { oracle_sys_linearize((oracle_sys*)c, chi, md); }
static void sys_trial(void* c, double lambda, double* tc, double* sc, int32_t* pit, int32_t* ok) {
    oracle_sys* s = (oracle_sys*)c;
    if (s->prm.trust_region == 1) {
        /* Gauss-Newton needs neither the trial chi2 nor computeScale */
        int it = 0;
        const int k = schur_solve(s, 0.0, &it);
        *pit = it; *ok = k;
        if (k) apply_update(s);
        return;
    }
    oracle_sys_trial(s, lambda, tc, sc, pit, ok);
}
static void sys_commit(void* c) {
    oracle_sys* s = (oracle_sys*)c;
    memcpy(s->pose, s->pose_trial, (size_t)s->Np * 56);
    memcpy(s->pt, s->pt_trial, (size_t)s->Nl * 24);
}
static int optimize_phase(oracle_sys* s, int n_iter, visfs_ba_stats* st, int phase) {
    const lm_problem P = { sys_begin, sys_linearize, sys_trial, sys_commit, s->prm.trust_region == 1 };
    return lm_optimize_phase(&P, s, n_iter, st, phase);
}

/* Scripted problem: trial t of the phase returns (temp_chi[t], scale[t], ok[t]); the committed chi2 is what linearise reports. */
typedef struct { int n, pos; const double* chi; const double* scale; const int32_t* ok; double committed, pending, max_diag; } lm_script;
static void scr_begin(void* c) { (void)c; }
static void scr_linearize(void* c, double* chi, double* md) { lm_script* q = (lm_script*)c; *chi = q->committed; *md = q->max_diag; }
static void scr_trial(void* c, double lambda, double* tc, double* sc, int32_t* pit, int32_t* ok) {
    lm_script* q = (lm_script*)c; (void)lambda;
    const int t = q->pos < q->n ? q->pos : q->n - 1;
    q->pos++;
    *tc = q->chi[t]; *sc = q->scale[t]; *ok = q->ok[t]; *pit = 0;
    q->pending = q->chi[t];
}
static void scr_commit(void* c) { lm_script* q = (lm_script*)c; q->committed = q->pending; }
int oracle_lm_script(int gauss_newton, int n_iter, double chi0, double max_diag0, int n_trials, const double* temp_chi,
                     const double* scale, const int32_t* ok, visfs_ba_stats* st) {
    lm_script q = { n_trials, 0, temp_chi, scale, ok, chi0, chi0, max_diag0 };
    const lm_problem P = { scr_begin, scr_linearize, scr_trial, scr_commit, gauss_newton };
    memset(st, 0, sizeof(*st));
    st->iterations_run[0] = lm_optimize_phase(&P, &q, n_iter, st, 0);
    st->chi2_final = q.committed;
    return q.pos;
}

/* Optimizer.cpp:283-303: chi2() > kernel->delta() (UNSQUARED) → level 1; s->chi2 holds computeActiveErrors' values */
static int mark_outliers(oracle_sys* s) {
    int n_out = 0;
    for (int k = 0; k < s->No; ++k) {
        if (s->obs_level[k] == 0 && s->obs_edge_ok[k] && s->chi2[k] > s->prm.robust_kernel_delta) {
            s->obs_level[k] = 1; s->outlier[k] = 1; ++n_out;
        }
    }
    return n_out;
}
/* stepping a solve by hand (tools/soak_diverge.py) */
void oracle_sys_commit(oracle_sys* s) { sys_commit(s); }
void oracle_sys_begin_phase(oracle_sys* s) { sys_begin(s); }
int oracle_sys_mark_outliers(oracle_sys* s) {
    (void)active_robust_chi2(s, s->pose, s->pt, 1);          /* computeActiveErrors at the estimate (:270) */
    memcpy(s->final_chi2, s->chi2, (size_t)s->No * 8);
    return s->prm.robust_kernel_delta > 0.0 ? mark_outliers(s) : 0;
}

/* ===================================================================== */
/* Optimizer/Framework=1: the Ceres branch (Optimizer.cpp:366-593)         */
/* ===================================================================== */
/* [ceres-upstream] TrustRegionMinimizer with LevenbergMarquardtStrategy, restated from the published algorithm of Ceres 2.0 / 2.1
 * (internal/ceres/trust_region_minimizer.cc, levenberg_marquardt_strategy.cc; the reference needs <= 2.1 for
 * ceres::LocalParameterization and pins no version).  Solver::Options defaults: initial_trust_region_radius 1e4, max 1e16, min 1e-32,
 * min_relative_decrease 1e-3, min / max_lm_diagonal 1e-6 / 1e32, function / gradient / parameter tolerance 1e-6 / 1e-10 / 1e-8,
 * max_num_consecutive_invalid_steps 5, jacobi_scaling on, monotonic steps.  The control flow exists once (ceres_tr_step /
 * ceres_tr_finalize) and is driven by the real system and by scripted outcomes (oracle_ceres_script). */
typedef struct {
    double radius, decrease_factor, cost, x_norm;
    int iter, invalid, done, reason;    /* reason: 1 max iterations, 2 gradient, 3 parameter, 4 function tolerance, 5 min radius, 6 invalid steps */
    /* DoglegStrategy (Optimizer/TrustRegion=1 under Framework=1): the multiplier of the Gauss-Newton regularisation and the norm of the
     * last dogleg step in the scaled space */
    int dogleg;
    double mu, dogleg_step_norm;
} ceres_tr;
static void ceres_tr_init(ceres_tr* t, double cost, double x_norm) {
    t->radius = 1e4; t->decrease_factor = 2.0; t->cost = cost; t->x_norm = x_norm; t->iter = 0; t->invalid = 0; t->done = 0; t->reason = 0;
    t->dogleg = 0; t->mu = 1e-8; t->dogleg_step_norm = 0.0;          /* kMinMu */
}
/* One pass of the minimizer loop after the linear solve: solve_ok = the strategy produced a finite step; model_cost_change =
 * -(J step)^T (f + J step / 2); cand_cost = cost at x (+) step (only read when the step is valid); step_norm = ||x - candidate||.
 * Returns 1 when the step is taken (the caller moves x, re-evaluates the Jacobian and reports ||g||_inf, ||x|| to ceres_tr_finalize). */
static int ceres_tr_step(ceres_tr* t, int solve_ok, double model_cost_change, double cand_cost, double step_norm) {
    if (!solve_ok || !(model_cost_change > 0.0)) {                       /* step_is_valid = model_cost_change > 0 */
        if (++t->invalid >= 5) { t->done = 1; t->reason = 6; }            /* HandleInvalidStep */
        else if (t->dogleg) t->mu *= 10.0;                                /* DoglegStrategy::StepIsInvalid: mu *= mu_increase_factor (reuse = false) */
        else t->radius *= 0.5;                                            /* LevenbergMarquardtStrategy::StepIsInvalid */
        return 0;
    }
    t->invalid = 0;
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;                       /* a failed evaluation of the candidate */
    if (step_norm <= 1e-8 * (t->x_norm + 1e-8)) { t->done = 1; t->reason = 3; return 0; }      /* ParameterToleranceReached */
    const double cost_change = t->cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * t->cost) { t->done = 1; t->reason = 4; return 0; }         /* FunctionToleranceReached: the step is NOT taken */
    const double rho = cost_change / model_cost_change;                  /* StepQuality, monotonic */
    if (rho > 1e-3) {
        t->cost = cand_cost;
        if (t->dogleg) {
            /* [ceres-upstream] DoglegStrategy::StepAccepted (dogleg_strategy.cc): decrease_threshold 0.25, increase_threshold 0.75 */
            if (rho < 0.25) t->radius *= 0.5;
            if (rho > 0.75) t->radius = fmax(t->radius, 3.0 * t->dogleg_step_norm);
            t->mu = fmax(1e-8, 2.0 * t->mu / 10.0);                       /* back towards a pure Gauss-Newton solve */
            return 1;
        }
        t->radius = t->radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3.0));               /* StepAccepted */
        t->radius = fmin(1e16, t->radius);
        t->decrease_factor = 2.0;
        return 1;
    }
    if (t->dogleg) { t->radius *= 0.5; return 0; }                        /* DoglegStrategy::StepRejected (reuse = true: the same Gauss-Newton step, a new interpolant) */
    t->radius = t->radius / t->decrease_factor;                          /* StepRejected */
    t->decrease_factor *= 2.0;
    return 0;
}
/* FinalizeIterationAndCheckIfMinimizerCanContinue (no time limit: see DESIGN.md): accepted = this iteration took its step. */
static void ceres_tr_finalize(ceres_tr* t, int max_iter, int accepted, double grad_max, double x_norm) {
    if (accepted) t->x_norm = x_norm;
    if (t->done) return;
    if (t->iter >= max_iter) { t->done = 1; t->reason = 1; return; }
    if (accepted && grad_max <= 1e-10) { t->done = 1; t->reason = 2; return; }
    if (t->radius < 1e-32) { t->done = 1; t->reason = 5; return; }
}

/* ||x|| and ||g||_inf of the reduced program: non-constant parameter blocks that appear in a residual block — free poses (7 numbers)
 * and free landmarks (3) with at least one observation; g in the tangent space the factors differentiate in. */
static int pose_has_residual(const oracle_sys* s, int i) {
    const int a = s->pose_idx[i];
    if (a < 0) return 0;
    return s->po_ptr[a + 1] > s->po_ptr[a] || (s->Nz > 0 && s->laser_pose == i);
}
static double ceres_x_norm2(const oracle_sys* s, const double* pose, const double* pt) {
    double n2 = 0.0;
    for (int i = 0; i < s->Np; ++i) if (pose_has_residual(s, i)) for (int c = 0; c < 7; ++c) n2 += pose[7 * i + c] * pose[7 * i + c];
    for (int l = 0; l < s->Nl; ++l) if (!s->pt_fixed[l] && s->lm_ptr[l + 1] > s->lm_ptr[l]) for (int c = 0; c < 3; ++c) n2 += pt[3 * l + c] * pt[3 * l + c];
    return n2;
}
static double ceres_step_norm2(const oracle_sys* s) {
    double n2 = 0.0;
    for (int i = 0; i < s->Np; ++i) if (pose_has_residual(s, i)) for (int c = 0; c < 7; ++c) { const double d = s->pose_trial[7 * i + c] - s->pose[7 * i + c]; n2 += d * d; }
    for (int l = 0; l < s->Nl; ++l) if (!s->pt_fixed[l] && s->lm_ptr[l + 1] > s->lm_ptr[l]) for (int c = 0; c < 3; ++c) { const double d = s->pt_trial[3 * l + c] - s->pt[3 * l + c]; n2 += d * d; }
    return n2;
}
static double ceres_grad_max(const oracle_sys* s) {
    double m = 0.0;
    for (int i = 0; i < s->n6; ++i) m = fmax(m, fabs(s->bp[i]));
    for (int l = 0; l < s->Nl; ++l) if (!s->pt_fixed[l]) for (int c = 0; c < 3; ++c) m = fmax(m, fabs(s->bl[3 * l + c]));
    return m;
}
/* LevenbergMarquardtStrategy::ComputeStep's diagonal in the UNSCALED variables: with the Jacobi scaling s_i = 1 / (1 + sqrt(H0_ii))
 * (H0 = J^T J at iteration zero) Ceres solves (S H S + D^2) y = S b with D_i^2 = clamp(H_ii s_i^2, 1e-6, 1e32) / radius and takes the
 * step S y; that is (H + diag(m_i) / radius) dx = b with m_i = clamp(H_ii s_i^2) / s_i^2. */
static void ceres_multipliers(oracle_sys* s) {
    for (int i = 0; i < s->n6; ++i) {
        const double h = s->Hpp[(size_t)i * s->n6 + i] * s->s2p[i];
        s->mp[i] = fmin(fmax(h, 1e-6), 1e32) / s->s2p[i];
    }
    for (int l = 0; l < s->Nl; ++l) {
        static const int dq[3] = { 0, 3, 5 };
        for (int c = 0; c < 3; ++c) {
            const double h = s->Hll[6 * l + dq[c]] * s->s2l[3 * l + c];
            s->ml[3 * l + c] = fmin(fmax(h, 1e-6), 1e32) / s->s2l[3 * l + c];
        }
    }
}
static int step_is_finite(const oracle_sys* s) {
    for (int i = 0; i < s->n6; ++i) if (!isfinite(s->dxp[i])) return 0;
    for (int i = 0; i < 3 * s->Nl; ++i) if (!isfinite(s->dxl[i])) return 0;
    return 1;
}

/* ceres::Solve(options, &problem, &summary) of Optimizer.cpp:504-527 + the outlier loop :529-540.  Every linear_solver_type the
 * branch can select (DENSE_SCHUR / DENSE_NORMAL_CHOLESKY / DENSE_QR) solves the same damped normal equations exactly: restated as
 * Schur elimination + dense Cholesky.  options.max_solver_time_in_seconds = 0.06 is NOT restated (the result would depend on the
 * machine).  DOGLEG (Optimizer/TrustRegion=1): ceres_dogleg_step above. */
/* [ceres-upstream] DoglegStrategy::ComputeStep, TRADITIONAL_DOGLEG (dogleg_strategy.cc of Ceres 2.0 / 2.1), in the unscaled variables.
 * Ceres works on the Jacobi-scaled Jacobian J' = J S and in the elliptical norm ||D y|| with D = sqrt(clamp(diag(J'^T J'), 1e-6, 1e32)):
 * with m_i = D_i^2 / S_i^2 (ceres_multipliers) the scaled gradient is g_i / sqrt(m_i), the Gauss-Newton step solves (H + mu M) dn = b
 * (b = -g; the regularised solve of ComputeGaussNewtonStep, lm_diagonal = D sqrt(mu)) and has scaled entries sqrt(m_i) dn_i, the Cauchy
 * point is -alpha v with v_i = g_i / m_i and alpha = ||g_s||^2 / ||J v||^2, and a scaled step A g_s + B gn_s is the step A v + B dn.
 * Returns 0 when the linear solver failed; fills s->dxp / s->dxl with the dogleg step, *mcc with the model cost change
 * -(J step)^T (f + J step / 2) = -(g . step + step^T H step / 2) and t->dogleg_step_norm. */
/* [ceres-upstream] DoglegStrategy::ComputeTraditionalDoglegStep (dogleg_strategy.cc) on the inner products of the scaled space:
 * S1 = ||g_s||^2, S2 = ||gn_s||^2, S3 = g_s . gn_s, JV2 = ||J v||^2 with v = g / m (alpha = S1 / JV2: the Cauchy point is -alpha g_s).
 * out = { A, B, dogleg_step_norm, model_cost_change }: the scaled step is A g_s + B gn_s, i.e. A v + B dn in the unscaled variables;
 * model_cost_change = -(g . step + step^T H step / 2) (TrustRegionMinimizer::ComputeTrustRegionStep: the model of the unregularised
 * problem), written with v^T H v = ||J v||^2 and H dn = -g - mu M dn. */
void oracle_dogleg_combine(double S1, double S2, double S3, double JV2, double radius, double mu, double out[4]) {
    const double gnorm = sqrt(S1), gn_norm = sqrt(S2), alpha = S1 / JV2;
    double A, B, norm;
    if (gn_norm <= radius) { A = 0.0; B = 1.0; norm = gn_norm; }                     /* case 1: the Gauss-Newton step lies inside */
    else if (gnorm * alpha >= radius) { A = -radius / gnorm; B = 0.0; norm = radius; } /* case 2: the Cauchy point lies outside */
    else {                                                                          /* case 3: the boundary point of the segment Cauchy -> Gauss-Newton */
        const double b_dot_a = -alpha * S3;
        const double a_sq = pow(alpha * gnorm, 2.0);
        const double bma_sq = a_sq - 2.0 * b_dot_a + pow(gn_norm, 2.0);
        const double c = b_dot_a - a_sq;
        const double d = sqrt(c * c + bma_sq * (pow(radius, 2.0) - a_sq));
        const double beta = (c <= 0.0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
        A = -alpha * (1.0 - beta); B = beta;
        norm = sqrt(A * A * S1 + 2.0 * A * B * S3 + B * B * S2);
    }
    const double sHs = A * A * JV2 + 2.0 * A * B * (-S1 - mu * S3) + B * B * (-S3 - mu * S2);
    out[0] = A; out[1] = B; out[2] = norm; out[3] = -(A * S1 + B * S3 + 0.5 * sHs);
}
static int ceres_dogleg_step(oracle_sys* s, ceres_tr* t, double* mcc) {
    int pit = 0;
    if (!schur_solve(s, t->mu, &pit) || !step_is_finite(s)) return 0;     /* (H + mu M) dn = b: s->dxp, s->dxl = dn */
    const int n6 = s->n6, Nl = s->Nl;
    /* ||g_s||^2, ||gn_s||^2, g_s . gn_s */
    double S1 = 0.0, S2 = 0.0, S3 = 0.0;
    for (int i = 0; i < n6; ++i) { const double g = -s->bp[i]; S1 += g * g / s->mp[i]; S2 += s->mp[i] * s->dxp[i] * s->dxp[i]; S3 += g * s->dxp[i]; }
    for (int l = 0; l < Nl; ++l) {
        if (s->pt_fixed[l] || !landmark_has_active_edge(s, l)) continue;
        for (int c = 0; c < 3; ++c) { const double g = -s->bl[3 * l + c], m = s->ml[3 * l + c], d = s->dxl[3 * l + c]; S1 += g * g / m; S2 += m * d * d; S3 += g * d; }
    }
    /* ||J v||^2 over the (robustified) residual blocks: v = g / m on the free variables */
    double JV2 = 0.0;
    const double delta = s->prm.robust_kernel_delta;
    for (int k = 0; k < s->No; ++k) {
        if (!edge_active(s, k)) continue;
        const int ip = s->obs_pose[k], l = s->obs_pt[k];
        double e[3], Ji[9], Jj[18];
        oracle_stereo_edge(s->pose + 7 * ip, s->pt + 3 * l, s->obs_uvr + 3 * k, s->intr, e, Ji, Jj);
        double rho[2]; ceres_huber(s->chi2[k], delta, rho);
        double r[3] = { 0.0, 0.0, 0.0 };
        if (!s->pt_fixed[l]) for (int q = 0; q < 3; ++q) for (int c = 0; c < 3; ++c) r[q] += Ji[3 * q + c] * (-s->bl[3 * l + c] / s->ml[3 * l + c]);
        const int a = s->pose_idx[ip];
        if (a >= 0) for (int q = 0; q < 3; ++q) for (int c = 0; c < 6; ++c) r[q] += Jj[6 * q + c] * (-s->bp[6 * a + c] / s->mp[6 * a + c]);
        JV2 += rho[1] * s->w_px * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    }
    if (s->Nz > 0 && s->pose_idx[s->laser_pose] >= 0) {
        const int a = s->pose_idx[s->laser_pose];
        for (int k = 0; k < s->Nz; ++k) {
            double e, J[6], r = 0.0;
            laser_edge_impl(s->pose + 7 * s->laser_pose, s->Tcr, s->laser_xyz + 3 * k, &s->grid, &e, J, s->ceres);
            for (int c = 0; c < 6; ++c) r += J[c] * (-s->bp[6 * a + c] / s->mp[6 * a + c]);
            JV2 += s->w_laser * r * r;
        }
    }
    double out[4];
    oracle_dogleg_combine(S1, S2, S3, JV2, t->radius, t->mu, out);
    const double A = out[0], B = out[1];
    t->dogleg_step_norm = out[2];
    *mcc = out[3];
    if (getenv("VISFS_ORACLE_TRACE")) fprintf(stderr, "[oracle dogleg] it %d radius %.6g mu %.3g |gn| %.6g |cauchy| %.6g -> A %.6g B %.6g mcc %.6g\n", t->iter, t->radius, t->mu, sqrt(S2), sqrt(S1) * S1 / JV2, A, B, *mcc);
    for (int i = 0; i < n6; ++i) s->dxp[i] = A * (-s->bp[i] / s->mp[i]) + B * s->dxp[i];
    for (int l = 0; l < Nl; ++l) {
        if (s->pt_fixed[l] || !landmark_has_active_edge(s, l)) { s->dxl[3 * l] = s->dxl[3 * l + 1] = s->dxl[3 * l + 2] = 0.0; continue; }
        for (int c = 0; c < 3; ++c) s->dxl[3 * l + c] = A * (-s->bl[3 * l + c] / s->ml[3 * l + c]) + B * s->dxl[3 * l + c];
    }
    return isfinite(*mcc) && step_is_finite(s);
}

/* [ceres-upstream] TrustRegionMinimizer::IterationZero with jacobi_scaling: column scale 1 / (1 + sqrt(diag(J^T J))), kept for the whole solve */
static void ceres_estimate_scale(oracle_sys* s) {
    for (int i = 0; i < s->n6; ++i) { const double q = 1.0 / (1.0 + sqrt(s->Hpp[(size_t)i * s->n6 + i])); s->s2p[i] = q * q; }
    for (int l = 0; l < s->Nl; ++l) { static const int dq[3] = { 0, 3, 5 }; for (int c = 0; c < 3; ++c) { const double q = 1.0 / (1.0 + sqrt(s->Hll[6 * l + dq[c]])); s->s2l[3 * l + c] = q * q; } }
}

/* Stage entry for tests (Optimizer/Framework=1 systems, after oracle_sys_linearize at iteration zero): ONE dogleg step with the given
 * radius and mu from the current linearisation — Jacobi scaling as at iteration zero, multipliers, ceres_dogleg_step, Plus.  The step is
 * left in the DX buffers, the trial state in the TRIAL buffers; out = { model cost change, dogleg step norm, trial cost (sum rho / 2) }. */
int oracle_sys_dogleg_trial(oracle_sys* s, double radius, double mu, double out[3]) {
    if (!s->ceres) return 0;
    const int saved_solver = s->prm.solver;
    s->prm.solver = 0;
    ceres_estimate_scale(s);
    ceres_multipliers(s);
    ceres_tr t;
    ceres_tr_init(&t, 0.0, 0.0);
    t.dogleg = 1; t.radius = radius; t.mu = mu; t.iter = 1;
    double mcc = 0.0;
    const int ok = ceres_dogleg_step(s, &t, &mcc);
    s->prm.solver = saved_solver;
    if (!ok) return 0;
    apply_update(s);
    out[0] = mcc; out[1] = t.dogleg_step_norm; out[2] = 0.5 * active_robust_chi2(s, s->pose_trial, s->pt_trial, 0);
    return 1;
}

static int ceres_optimize(oracle_sys* s, visfs_ba_stats* st) {
    const int dogleg = s->prm.trust_region == 1;                /* Optimizer.cpp:515-519: options.trust_region_strategy_type = ceres::DOGLEG */
    const int max_it = s->prm.iterations;
    const int saved_solver = s->prm.solver;
    s->prm.solver = 0;                                         /* schur_solve: dense Cholesky on the reduced system */
    double chi, md;
    oracle_sys_linearize(s, &chi, &md);                        /* IterationZero: cost = sum rho / 2, robustified Jacobian */
    st->chi2_initial = chi;
    { int n_ok = 0; for (int k = 0; k < s->No; ++k) n_ok += s->obs_edge_ok[k]; st->n_active_edges[0] = st->n_active_edges[1] = n_ok; }
    ceres_estimate_scale(s);
    ceres_tr t;
    ceres_tr_init(&t, 0.5 * chi, sqrt(ceres_x_norm2(s, s->pose, s->pt)));
    t.dogleg = dogleg;
    if (max_it <= 0) { t.done = 1; t.reason = 1; }             /* the first FinalizeIteration... sees iteration 0 >= max_num_iterations */
    else if (ceres_grad_max(s) <= 1e-10) { t.done = 1; t.reason = 2; }
    while (!t.done) {
        t.iter++;
        ceres_multipliers(s);
        int pit = 0;
        double mcc = 0.0, cand = 0.0, step_norm = 0.0;
        int ok;
        st->trials_run[0]++;
        if (dogleg) {
            /* (a rejected step makes Ceres reuse the stored Gauss-Newton step and gradient with the halved radius — recomputing them with
             * the same mu gives the same vectors; its inner retry loop on a failed factorisation, mu *= 10 up to 1, is folded into the
             * invalid-step rule: one retry per iteration) */
            ok = ceres_dogleg_step(s, &t, &mcc);
        } else {
            ok = schur_solve(s, 1.0 / t.radius, &pit);
            if (ok && !step_is_finite(s)) ok = 0;
            if (ok) mcc = 0.5 * compute_scale(s, 1.0 / t.radius);  /* = step^T b - step^T H step / 2 with (H + D) step = b */
        }
        if (ok && mcc > 0.0) {
            apply_update(s);                                   /* Plus: t + dt, (deltaQ(dtheta) * q).normalized() (LocalParameterization.cpp:10-24) */
            cand = 0.5 * active_robust_chi2(s, s->pose_trial, s->pt_trial, 0);
            step_norm = sqrt(ceres_step_norm2(s));
        }
        const int accepted = ceres_tr_step(&t, ok, mcc, cand, step_norm);
        double gmax = 0.0, xn = t.x_norm;
        if (accepted) {
            sys_commit(s);
            xn = sqrt(ceres_x_norm2(s, s->pose, s->pt));
            oracle_sys_linearize(s, &chi, &md);
            gmax = ceres_grad_max(s);
        }
        if (st->n_trace < VISFS_BA_MAX_TRACE) { st->trace_lambda[st->n_trace] = t.radius; st->trace_chi2[st->n_trace] = 2.0 * t.cost; st->n_trace++; }
        ceres_tr_finalize(&t, max_it, accepted, gmax, xn);
    }
    st->iterations_run[0] = t.iter;
    s->prm.solver = saved_solver;
    for (int i = 0; i < 3 * s->Nl; ++i) s->ml[i] = 1.0;
    for (int i = 0; i < s->n6; ++i) s->mp[i] = 1.0;
    /* :529-540: every stereo residual block (both-constant ones included), error . (pixelInfo error) > delta, unsquared */
    const double chi_final = active_robust_chi2(s, s->pose, s->pt, 0);
    st->chi2_phase1 = st->chi2_final = chi_final;
    const double iv = 1.0 / s->prm.pixel_variance;
    int n_out = 0;
    for (int k = 0; k < s->No; ++k) {
        double e[3];
        oracle_stereo_edge(s->pose + 7 * s->obs_pose[k], s->pt + 3 * s->obs_pt[k], s->obs_uvr + 3 * k, s->intr, e, NULL, NULL);
        const double c = e[0] * (iv * e[0]) + e[1] * (iv * e[1]) + e[2] * (iv * e[2]);
        s->final_chi2[k] = c;
        if (s->prm.robust_kernel_delta > 0.0 && c > s->prm.robust_kernel_delta) { s->outlier[k] = 1; ++n_out; }
    }
    st->n_outliers = n_out;
    return VISFS_BA_OK;
}

/* The trust-region schedule on SCRIPTED outcomes: iteration t's solve reports (ok[t], model_cost_change[t], cand_cost[t],
 * step_norm[t]); an accepted step then reports (grad_max[t], x_norm[t]).  Fills trace_lambda (radius after each iteration),
 * trace_chi2 (2 x cost), iterations_run[0]; returns the termination reason (ceres_tr::reason). */
int oracle_ceres_script(int max_iter, double cost0, double x_norm0, double grad_max0, int n, const int32_t* ok, const double* mcc,
                        const double* cand_cost, const double* step_norm, const double* grad_max, const double* x_norm, visfs_ba_stats* st) {
    memset(st, 0, sizeof(*st));
    ceres_tr t;
    ceres_tr_init(&t, cost0, x_norm0);
    if (max_iter <= 0) { t.done = 1; t.reason = 1; }
    else if (grad_max0 <= 1e-10) { t.done = 1; t.reason = 2; }
    while (!t.done) {
        const int q = t.iter < n ? t.iter : n - 1;
        t.iter++;
        st->trials_run[0]++;
        const int accepted = ceres_tr_step(&t, ok[q], mcc[q], cand_cost[q], step_norm[q]);
        if (st->n_trace < VISFS_BA_MAX_TRACE) { st->trace_lambda[st->n_trace] = t.radius; st->trace_chi2[st->n_trace] = 2.0 * t.cost; st->n_trace++; }
        ceres_tr_finalize(&t, max_iter, accepted, grad_max[q], x_norm[q]);
    }
    st->iterations_run[0] = t.iter;
    st->chi2_final = 2.0 * t.cost;
    return t.reason;
}

/* the same loop with DoglegStrategy's rules: iteration t's step had the scaled length dogleg_step_norm[t]; mu_trace[i] = mu after iteration i */
int oracle_dogleg_script(int max_iter, double cost0, double x_norm0, double grad_max0, int n, const int32_t* ok, const double* mcc,
                         const double* cand_cost, const double* step_norm, const double* dogleg_step_norm, const double* grad_max, const double* x_norm,
                         visfs_ba_stats* st, double* mu_trace) {
    memset(st, 0, sizeof(*st));
    ceres_tr t;
    ceres_tr_init(&t, cost0, x_norm0);
    t.dogleg = 1;
    if (max_iter <= 0) { t.done = 1; t.reason = 1; }
    else if (grad_max0 <= 1e-10) { t.done = 1; t.reason = 2; }
    while (!t.done) {
        const int q = t.iter < n ? t.iter : n - 1;
        t.iter++;
        st->trials_run[0]++;
        t.dogleg_step_norm = dogleg_step_norm[q];
        const int accepted = ceres_tr_step(&t, ok[q], mcc[q], cand_cost[q], step_norm[q]);
        if (st->n_trace < VISFS_BA_MAX_TRACE) { mu_trace[st->n_trace] = t.mu; st->trace_lambda[st->n_trace] = t.radius; st->trace_chi2[st->n_trace] = 2.0 * t.cost; st->n_trace++; }
        ceres_tr_finalize(&t, max_iter, accepted, grad_max[q], x_norm[q]);
    }
    st->iterations_run[0] = t.iter;
    st->chi2_final = 2.0 * t.cost;
    return t.reason;
}

int oracle_sys_optimize(oracle_sys* s, visfs_ba_stats* st, double* seconds) {
    visfs_ba_stats local;
    if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (s->ceres) {
        const int rc = ceres_optimize(s, st);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        st->status = rc;
        return rc;
    }
    const int half = s->prm.iterations / 2;
    st->iterations_run[0] = optimize_phase(s, half, st, 0);            /* Optimizer.cpp:265 */
    { int n_ok = 0; for (int k = 0; k < s->No; ++k) n_ok += s->obs_edge_ok[k]; st->n_active_edges[0] = st->n_active_edges[1] = n_ok; }
    st->pcg_iterations_phase[0] = st->pcg_iterations;
    /* :270-271 */
    double chi2 = active_robust_chi2(s, s->pose, s->pt, 1);
    if (half == 0) st->chi2_initial = chi2;
    st->chi2_phase1 = chi2; st->chi2_final = chi2;
    memcpy(s->final_chi2, s->chi2, (size_t)s->No * 8);
    int status = VISFS_BA_OK;
    if (isnan(chi2)) status = VISFS_BA_ERR_NAN_CHI2;                   /* :272-275 */
    else if (chi2 > 1000000000000.0 || !isfinite(chi2)) status = VISFS_BA_ERR_HUGE_CHI2_1;  /* :277-280 */
    if (status == VISFS_BA_OK && s->prm.robust_kernel_delta > 0.0) {
        const int n_out = mark_outliers(s);
        st->n_outliers = n_out;
        st->n_active_edges[1] = st->n_active_edges[0] - n_out;
        st->iterations_run[1] = optimize_phase(s, half, st, 1);        /* :310-311 */
        /* :315 — activeRobustChi2 with the errors of the LAST evaluated state; restated at the committed state */
        st->chi2_final = active_robust_chi2(s, s->pose, s->pt, 0);
        if (st->chi2_final > 1000000000000.0) status = VISFS_BA_ERR_HUGE_CHI2_2;
    }
    st->pcg_iterations_phase[1] = st->pcg_iterations - st->pcg_iterations_phase[0];
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    st->status = status;
    return status;
}

void oracle_sys_download(oracle_sys* s, double* pose_tq, double* point_xyz, uint8_t* obs_outlier, double* obs_chi2) {
    if (pose_tq) memcpy(pose_tq, s->pose, (size_t)s->Np * 56);
    if (point_xyz) memcpy(point_xyz, s->pt, (size_t)s->Nl * 24);
    if (obs_outlier) memcpy(obs_outlier, s->outlier, s->No);
    if (obs_chi2) memcpy(obs_chi2, s->final_chi2, (size_t)s->No * 8);
}

int oracle_solve_window(const visfs_ba_params* prm, const visfs_ba_window* w, visfs_ba_result* r, int nthreads) {
    r->n_poses_out = 0; r->n_outliers = 0; r->warn_mono_skipped = 0;
    r->iterations_run[0] = r->iterations_run[1] = 0;
    r->chi2_initial = r->chi2_phase1 = r->chi2_final = 0.0;
    if (prm->framework != 0 && prm->framework != 1) return r->status = VISFS_BA_ERR_UNSUPPORTED;
    /* guards: Optimizer.cpp:74, 360-364 (g2o) and :368, 586-590 (Ceres): the same conditions */
    if (!(w->n_poses >= 2 && prm->iterations > 0 && w->pose_ids[0] > 0)) {
        if (w->n_poses == 1 || prm->iterations <= 0) {
            for (int i = 0; i < w->n_poses; ++i) { r->pose_ids_out[i] = w->pose_ids[i]; memcpy(r->pose_Twr_out + 12 * i, w->pose_Twr + 12 * i, 96); }
            r->n_poses_out = w->n_poses;
            return r->status = VISFS_BA_PASSTHROUGH;
        }
        return r->status = VISFS_BA_ERR_TOO_FEW_POSES;
    }
    const int Np = w->n_poses, Nl = w->n_points, Nr = w->n_refs, Nk = w->n_links;
    double* pose_tq = xcalloc((size_t)Np * 7, 8); uint8_t* pose_fixed = xcalloc(Np, 1); uint8_t* used = xcalloc(Nl, 1);
    int32_t* op = xcalloc(Nr, 4); int32_t* oc = xcalloc(Nr, 4); int32_t* oref = xcalloc(Nr, 4); double* uvr = xcalloc((size_t)Nr * 3, 8);
    int32_t* of = xcalloc(Nk, 4); int32_t* ot = xcalloc(Nk, 4); double* otq = xcalloc((size_t)Nk * 7, 8);
    visfs_ba_graph g; int32_t mono = 0;
    oracle_pack_window(prm, w, pose_tq, pose_fixed, used, op, oc, uvr, oref, of, ot, otq, &g, &mono);
    r->warn_mono_skipped = mono;
    oracle_sys* s = oracle_sys_create(prm, &g, nthreads);
    visfs_ba_stats st;
    const int status = oracle_sys_optimize(s, &st, NULL);
    r->status = status;
    r->iterations_run[0] = st.iterations_run[0]; r->iterations_run[1] = st.iterations_run[1];
    r->chi2_initial = st.chi2_initial; r->chi2_phase1 = st.chi2_phase1; r->chi2_final = st.chi2_final;
    /* outliers are appended at :296 before the phase-2 abort check, so they are reported even on HUGE_CHI2_2 */
    if (status == VISFS_BA_OK || status == VISFS_BA_ERR_HUGE_CHI2_2) {
        int n = 0;
        for (int k = 0; k < g.n_obs; ++k) if (s->outlier[k] && n < r->outlier_capacity) {
            r->outlier_feature[n] = w->ref_feature[oref[k]]; r->outlier_pose[n] = w->ref_pose[oref[k]]; ++n;
        }
        r->n_outliers = n;
    }
    if (status == VISFS_BA_OK) {
        for (int i = 0; i < Np; ++i) { r->pose_ids_out[i] = w->pose_ids[i]; oracle_unpack_pose(s->pose + 7 * i, w->Trc, r->pose_Twr_out + 12 * i); }
        r->n_poses_out = Np;
        /* points: Optimizer.cpp:343-358 */
        for (int l = 0; l < Nl; ++l) {
            double* p = w->point_xyz + 3 * l;
            if (used[l]) {
                const double dx = p[0] - s->pt[3*l], dy = p[1] - s->pt[3*l+1], dz = p[2] - s->pt[3*l+2];
                if (sqrt(dx * dx + dy * dy + dz * dz) < 5.0) { p[0] = s->pt[3*l]; p[1] = s->pt[3*l+1]; p[2] = s->pt[3*l+2]; }
            } else { p[0] = p[1] = p[2] = NAN; }
        }
    }
    oracle_sys_destroy(s);
    free(pose_tq); free(pose_fixed); free(used); free(op); free(oc); free(oref); free(uvr); free(of); free(ot); free(otq);
    return status;
}


/* ---- measurement helper (bench.py's cpu_baseline leg): where the OpenMP threads run ----
 * The GPU box grants a few cores of a host with 16 last-level-cache domains; unpinned, the scheduler spreads the team over them and the
 * same binary measured 181 / 673 / 465 it/s in three rounds.  oracle_omp_pin(cpus, n): thread t of the team (libgomp keeps
 * its threads between regions) binds itself to cpus[t]; the calling thread's own mask is saved and given back by oracle_omp_unpin().
 * Returns the number of threads that could be bound (0 in the serial build). */
#if defined(_OPENMP) && defined(__linux__)
#include <sched.h>
static cpu_set_t g_saved_mask;
static int g_saved = 0;
int oracle_omp_pin(const int32_t* cpus, int n) {
    int bound = 0;
    if (n < 1) return 0;
    if (!g_saved && sched_getaffinity(0, sizeof(g_saved_mask), &g_saved_mask) == 0) g_saved = 1;
#pragma omp parallel num_threads(n) reduction(+:bound)
    {
        const int t = omp_get_thread_num();
        if (t < n) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpus[t], &set);
            if (sched_setaffinity(0, sizeof(set), &set) == 0) bound += 1;
        }
    }
    return bound;
}
void oracle_omp_unpin(void) {
    if (g_saved) (void)sched_setaffinity(0, sizeof(g_saved_mask), &g_saved_mask);   /* the calling thread only: the team stays where it is */
}
#else
int oracle_omp_pin(const int32_t* cpus, int n) { (void)cpus; (void)n; return 0; }
void oracle_omp_unpin(void) {}
#endif
