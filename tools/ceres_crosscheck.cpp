// ceres_crosscheck.cpp — runs the REAL Ceres Solver on a flat factor-graph dump (.vbag with Optimizer/Framework=1,
// visfs_amd/graphio.py) the way the reference's Ceres branch does, and writes the result (.vbar).
//
// Purpose: pin this repo's restatement of the Ceres branch of Optimizer::localOptimize (oracle/visfs_ba_oracle.c: ceres_optimize,
// "PARITY UNPINNED") against the library the reference actually calls.  Nothing in this image or on the GPU box can build it (no
// Eigen, no Ceres: SURVEY.md §8c, profiles/r02_probe_box.log) — it is the hand-off for whoever has a Ceres <= 2.1 (the reference
// uses ceres::LocalParameterization, removed in 2.2):
//
//   g++ -std=c++17 -O2 tools/ceres_crosscheck.cpp -I/usr/include/eigen3 -lceres -lglog -o ceres_crosscheck
//   python tools/dump_graphs.py                      # writes tests/golden/graphs/*.vbag (committed; the ceres_* ones are for this tool)
//   for f in tests/golden/graphs/ceres_*.vbag; do ./ceres_crosscheck $f ${f%.vbag}.vbar "$(pkg-config --modversion ceres)"; done
//   python tools/g2o_golden_import.py tests/golden/graphs/ceres_*.vbar   # → tests/golden/ref_ceres_*.npz + report vs the oracle
//
// Two ways to get the cost functions and parameterizations:
//   default                 : the classes below — written from the formulas of corelib/src/Optimizer/ceres/StereoObservationFactor.cpp:12-76
//                             and LocalParameterization.cpp:10-48 (same operation order where it matters for rounding), self-contained;
//   -DVISFS_REFERENCE_TYPES : include the reference's own headers instead (add -I<reference>/corelib/include -I<reference>/utilite/include
//                             and compile <reference>/corelib/src/Optimizer/ceres/{StereoObservationFactor,LocalParameterization}.cpp
//                             beside this file) — then every arithmetic instruction of the factors is the reference's.
// The driver follows Optimizer.cpp:369-403 (pose blocks, root constant), :427-478 (point blocks, stereo residual blocks under ONE
// HuberLoss(robustKernelDelta)), :504-527 (options) and :529-540 (outlier loop) call for call, with ONE deliberate difference:
// options.max_solver_time_in_seconds = 0.06 is NOT set unless --time-cap is given — a wall-clock cap makes the result depend on the
// machine, and neither the oracle nor the GPU path restates it.  Wheel-odometry links never become factors in that
// branch (:405-422) and laser factors are outside the .vbag format, so neither appears here.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <ceres/ceres.h>

#ifdef VISFS_REFERENCE_TYPES
#include "Optimizer/ceres/LocalParameterization.h"
#include "Optimizer/ceres/StereoObservationFactor.h"
using StereoFactor = VISFS::Optimizer::StereoObservationFactor;
using PoseParam = VISFS::Optimizer::PoseLocalParameterization;
using PointParam = VISFS::Optimizer::PointLocalParameterization;
#else
namespace {
// utilite/include/Math.h:277-287: first order, not normalised
inline Eigen::Quaterniond deltaQ(const Eigen::Vector3d& theta) {
    Eigen::Quaterniond dq;
    const Eigen::Vector3d half = theta / 2.0;
    dq.w() = 1.0; dq.x() = half.x(); dq.y() = half.y(); dq.z() = half.z();
    return dq;
}
// LocalParameterization.cpp:10-32: x = [t, qx qy qz qw]; Plus: t + dt, (deltaQ(dtheta) * q).normalized(); Jacobian [I6; 0]
class PoseParam : public ceres::LocalParameterization {
public:
    bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
        Eigen::Map<const Eigen::Vector3d> p0(x);
        Eigen::Map<const Eigen::Quaterniond> q0(x + 3);
        Eigen::Map<const Eigen::Vector3d> dp(delta);
        const Eigen::Quaterniond dq = deltaQ(Eigen::Map<const Eigen::Vector3d>(delta + 3));
        Eigen::Map<Eigen::Vector3d> p(x_plus_delta);
        Eigen::Map<Eigen::Quaterniond> q(x_plus_delta + 3);
        p = p0 + dp;
        q = (dq * q0).normalized();
        return true;
    }
    bool ComputeJacobian(const double*, double* jacobian) const override {
        Eigen::Map<Eigen::Matrix<double, 7, 6, Eigen::RowMajor>> j(jacobian);
        j.topRows<6>().setIdentity();
        j.bottomRows<1>().setZero();
        return true;
    }
    int GlobalSize() const override { return 7; }
    int LocalSize() const override { return 6; }
};
// LocalParameterization.cpp:34-48
class PointParam : public ceres::LocalParameterization {
public:
    bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
        for (int i = 0; i < 3; ++i) x_plus_delta[i] = x[i] + delta[i];
        return true;
    }
    bool ComputeJacobian(const double*, double* jacobian) const override {
        Eigen::Map<Eigen::Matrix<double, 3, 3, Eigen::RowMajor>> j(jacobian);
        j.setIdentity();
        return true;
    }
    int GlobalSize() const override { return 3; }
    int LocalSize() const override { return 3; }
};
// StereoObservationFactor.cpp:12-76: residual = info * (obs - project(R Pw + t)); Jacobians premultiplied by info
class StereoFactor : public ceres::SizedCostFunction<3, 3, 7> {
public:
    StereoFactor(double fx, double fy, double cx, double cy, double bf, const Eigen::Matrix3d& info, const Eigen::Vector3d& obs)
        : fx_(fx), fy_(fy), cx_(cx), cy_(cy), bf_(bf), info_(info), obs_(obs) {}
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        const Eigen::Vector3d Pw(parameters[0][0], parameters[0][1], parameters[0][2]);
        const Eigen::Vector3d tcw(parameters[1][0], parameters[1][1], parameters[1][2]);
        const Eigen::Quaterniond qcw(parameters[1][6], parameters[1][3], parameters[1][4], parameters[1][5]);
        const Eigen::Matrix3d R = qcw.toRotationMatrix();
        const Eigen::Vector3d Pc = R * Pw + tcw;
        const double x = Pc[0], y = Pc[1], z = Pc[2], z_2 = z * z;
        const double invZ = 1.0 / z;
        Eigen::Vector3d proj;
        proj[0] = x * invZ * fx_ + cx_;
        proj[1] = y * invZ * fy_ + cy_;
        proj[2] = proj[0] - bf_ * invZ;
        Eigen::Map<Eigen::Vector3d> residual(residuals);
        residual = obs_ - proj;
        residual = info_ * residual;
        if (jacobians) {
            if (jacobians[0]) {
                Eigen::Map<Eigen::Matrix<double, 3, 3, Eigen::RowMajor>> J(jacobians[0]);
                for (int c = 0; c < 3; ++c) {
                    J(0, c) = -fx_ * R(0, c) / z + fx_ * x * R(2, c) / z_2;
                    J(1, c) = -fy_ * R(1, c) / z + fy_ * y * R(2, c) / z_2;
                    J(2, c) = J(0, c) - bf_ * R(2, c) / z_2;
                }
                J = info_ * J;
            }
            if (jacobians[1]) {
                Eigen::Map<Eigen::Matrix<double, 3, 7, Eigen::RowMajor>> J(jacobians[1]);
                J(0, 0) = -1. / z * fx_; J(0, 1) = 0.; J(0, 2) = x / z_2 * fx_; J(0, 3) = x * y / z_2 * fx_;
                J(0, 4) = -(1. + (x * x / z_2)) * fx_; J(0, 5) = y / z * fx_; J(0, 6) = 0.;
                J(1, 0) = 0.; J(1, 1) = -1. / z * fy_; J(1, 2) = y / z_2 * fy_; J(1, 3) = (1. + y * y / z_2) * fy_;
                J(1, 4) = -x * y / z_2 * fy_; J(1, 5) = -x / z * fy_; J(1, 6) = 0.;
                J(2, 0) = J(0, 0); J(2, 1) = 0.; J(2, 2) = J(0, 2) - bf_ / z_2; J(2, 3) = J(0, 3) - bf_ * y / z_2;
                J(2, 4) = J(0, 4) + bf_ * x / z_2; J(2, 5) = J(0, 5); J(2, 6) = 0.;
                J.leftCols<6>() = info_ * J.leftCols<6>();
            }
        }
        return true;
    }
private:
    double fx_, fy_, cx_, cy_, bf_;
    Eigen::Matrix3d info_;
    Eigen::Vector3d obs_;
};
}  // namespace
#endif

namespace {

struct Dump {           // the scalar groups are read straight into consecutive members: keep their order and types
    int32_t framework, solver, trust_region, iterations;
    double pixel_variance, odometry_covariance, laser_covariance, robust_kernel_delta;
    int32_t n_poses, n_points, n_obs, n_odo;
    double fx, fy, cx, cy, bf;
    std::vector<double> pose_tq, point_xyz, obs_uvr, odo_tq;
    std::vector<uint8_t> pose_fixed, point_fixed;
    std::vector<int32_t> obs_point, obs_pose, odo_from, odo_to;
};
template <typename T> bool rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), sizeof(T) * n); return bool(f); }
bool load(const char* path, Dump& d) {
    std::ifstream f(path, std::ios::binary);
    char magic[8]; uint32_t ver = 0;
    if (!f || !rd(f, magic, 8) || std::memcmp(magic, "VISFSBAG", 8) != 0 || !rd(f, &ver, 1) || ver != 1) return false;
    if (!rd(f, &d.framework, 4) || !rd(f, &d.pixel_variance, 4) || !rd(f, &d.n_poses, 4) || !rd(f, &d.fx, 5)) return false;
    d.pose_tq.resize(7 * (size_t)d.n_poses); d.pose_fixed.resize(d.n_poses);
    d.point_xyz.resize(3 * (size_t)d.n_points); d.point_fixed.resize(d.n_points);
    d.obs_point.resize(d.n_obs); d.obs_pose.resize(d.n_obs); d.obs_uvr.resize(3 * (size_t)d.n_obs);
    d.odo_from.resize(d.n_odo); d.odo_to.resize(d.n_odo); d.odo_tq.resize(7 * (size_t)d.n_odo);
    return rd(f, d.pose_tq.data(), d.pose_tq.size()) && rd(f, d.pose_fixed.data(), d.pose_fixed.size()) &&
           rd(f, d.point_xyz.data(), d.point_xyz.size()) && rd(f, d.point_fixed.data(), d.point_fixed.size()) &&
           rd(f, d.obs_point.data(), d.obs_point.size()) && rd(f, d.obs_pose.data(), d.obs_pose.size()) &&
           rd(f, d.obs_uvr.data(), d.obs_uvr.size()) && rd(f, d.odo_from.data(), d.odo_from.size()) &&
           rd(f, d.odo_to.data(), d.odo_to.size()) && (d.n_odo == 0 || rd(f, d.odo_tq.data(), d.odo_tq.size()));
}
template <typename T> void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s graph.vbag result.vbar [ceres-version-string] [--time-cap]\n", argv[0]); return 2; }
    bool time_cap = false;
    std::string version = "(version not given)";
    for (int a = 3; a < argc; ++a) { if (std::string(argv[a]) == "--time-cap") time_cap = true; else version = argv[a]; }
    Dump d;
    if (!load(argv[1], d)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    if (d.framework != 1) { std::fprintf(stderr, "only Optimizer/Framework=1 dumps (the g2o branch: tools/g2o_crosscheck.cpp)\n"); return 2; }

    // Optimizer.cpp:369-403: one 7-block per pose (the dump holds Tcw as [t, qx qy qz qw], positified at packing), the root constant
    ceres::Problem problem;
    ceres::LossFunction* loss = new ceres::HuberLoss(d.robust_kernel_delta);                         // :370 — ONE loss object for all blocks
    std::vector<double> poses = d.pose_tq, points = d.point_xyz;
    for (int i = 0; i < d.n_poses; ++i) {
        problem.AddParameterBlock(poses.data() + 7 * (size_t)i, 7, new PoseParam());
        if (d.pose_fixed[i]) problem.SetParameterBlockConstant(poses.data() + 7 * (size_t)i);
    }
    // :427-478: one 3-block per landmark of the dump (the packer keeps those that appear in wordReferences), constant when flagged
    const Eigen::Matrix3d info = Eigen::Matrix3d::Identity() / d.pixel_variance;                      // :431
    for (int l = 0; l < d.n_points; ++l) {
        problem.AddParameterBlock(points.data() + 3 * (size_t)l, 3, new PointParam());
        if (d.point_fixed[l]) problem.SetParameterBlockConstant(points.data() + 3 * (size_t)l);
    }
    // the dump's observations are in the reference's insertion order (feature-major, then pose-major)
    for (int k = 0; k < d.n_obs; ++k) {
        const Eigen::Vector3d obs(d.obs_uvr[3 * (size_t)k], d.obs_uvr[3 * (size_t)k + 1], d.obs_uvr[3 * (size_t)k + 2]);
        problem.AddResidualBlock(new StereoFactor(d.fx, d.fy, d.cx, d.cy, d.bf, info, obs), loss,
                                 points.data() + 3 * (size_t)d.obs_point[k], poses.data() + 7 * (size_t)d.obs_pose[k]);   // :468-469
    }
    // :504-527
    ceres::Solver::Options options;
    ceres::Solver::Summary summary;
    if (d.solver == 0) options.linear_solver_type = ceres::DENSE_SCHUR;
    else if (d.solver == 1) options.linear_solver_type = ceres::DENSE_NORMAL_CHOLESKY;
    else if (d.solver == 2) options.linear_solver_type = ceres::DENSE_QR;
    if (d.trust_region == 0) options.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
    else if (d.trust_region == 1) options.trust_region_strategy_type = ceres::DOGLEG;
    options.max_num_iterations = d.iterations;
    if (time_cap) options.max_solver_time_in_seconds = 0.06;
    options.num_threads = 2;
    ceres::Solve(options, &problem, &summary);

    // :529-540
    std::vector<uint8_t> outlier(d.n_obs, 0);
    std::vector<double> edge_chi2(d.n_obs, 0.0);
    int n_outliers = 0;
    for (int k = 0; k < d.n_obs; ++k) {
        const double* p = poses.data() + 7 * (size_t)d.obs_pose[k];
        const double* w = points.data() + 3 * (size_t)d.obs_point[k];
        const Eigen::Vector3d tcw(p[0], p[1], p[2]);
        const Eigen::Quaterniond qcw(p[6], p[3], p[4], p[5]);
        const Eigen::Vector3d pc = qcw * Eigen::Vector3d(w[0], w[1], w[2]) + tcw;
        const double invZ = 1. / pc[2];
        Eigen::Vector3d e;
        e[0] = pc[0] * invZ * d.fx + d.cx;
        e[1] = pc[1] * invZ * d.fy + d.cy;
        e[2] = e[0] - d.bf * invZ;
        e = Eigen::Vector3d(d.obs_uvr[3 * (size_t)k], d.obs_uvr[3 * (size_t)k + 1], d.obs_uvr[3 * (size_t)k + 2]) - e;
        edge_chi2[k] = e.dot(info * e);
        if (d.robust_kernel_delta > 0.0 && edge_chi2[k] > d.robust_kernel_delta) { outlier[k] = 1; ++n_outliers; }
    }
    // result: iterations = passes of the minimizer loop (successful + unsuccessful), chi2 = 2 x cost (sum of rho)
    const int iterations = summary.iterations.empty() ? 0 : (int)summary.iterations.size() - 1;
    const double chi2_initial = 2.0 * summary.initial_cost, chi2_final = 2.0 * summary.final_cost;
    std::string prov = std::string("ceres ") + version + ", " + ceres::LinearSolverTypeToString(options.linear_solver_type) +
                       (time_cap ? ", 60 ms cap" : ", no time cap") + ", " + summary.message +
#ifdef VISFS_REFERENCE_TYPES
                       ", reference factors (corelib/src/Optimizer/ceres)";
#else
                       ", re-typed factors (tools/ceres_crosscheck.cpp)";
#endif
    std::ofstream f(argv[2], std::ios::binary);
    const uint32_t ver = 1;
    f.write("VISFSBAR", 8); wr(f, &ver, 1);
    const int32_t head[4] = { 0, iterations, 0, n_outliers };
    wr(f, head, 4);
    const double chis[3] = { chi2_initial, chi2_final, chi2_final };
    wr(f, chis, 3);
    const int32_t sizes[3] = { d.n_poses, d.n_points, d.n_obs };
    wr(f, sizes, 3);
    wr(f, poses.data(), poses.size()); wr(f, points.data(), points.size());
    wr(f, outlier.data(), outlier.size()); wr(f, edge_chi2.data(), edge_chi2.size());
    const uint32_t n = (uint32_t)prov.size();
    wr(f, &n, 1); f.write(prov.data(), n);
    std::printf("%s: %s; iterations %d, outliers %d, 2 x cost %.9g -> %.9g\n", argv[1], summary.BriefReport().c_str(), iterations, n_outliers,
                chi2_initial, chi2_final);
    return f ? 0 : 1;
}
