// g2o_crosscheck.cpp — runs the REAL g2o on a flat factor-graph dump (.vbag, visfs_amd/graphio.py) and writes the result (.vbar).
//
// Purpose: pin this repo's CPU oracle (oracle/visfs_ba_oracle.c, "PARITY UNPINNED") against the library the reference actually
// calls.  Nothing in this image can build it (no Eigen, no g2o: SURVEY.md §8c; the GPU box has none either:
// profiles/r02_probe_box.log) — it is the hand-off for whoever has a g2o checkout of the reference's vintage
// (release 20201223_git … before `number_t` was removed):
//
//   g++ -std=c++17 -O2 tools/g2o_crosscheck.cpp -I/usr/include/eigen3 -lg2o_core -lg2o_stuff -lg2o_types_sba \
//       -lg2o_types_slam3d -lg2o_solver_eigen -lg2o_solver_pcg [-DG2O_HAVE_CSPARSE -lg2o_solver_csparse -lg2o_csparse_extension] \
//       -o g2o_crosscheck
//   python tools/dump_graphs.py                      # writes tests/golden/graphs/*.vbag (committed)
//   for f in tests/golden/graphs/*.vbag; do ./g2o_crosscheck $f ${f%.vbag}.vbar; done
//   python tools/g2o_golden_import.py tests/golden/graphs/*.vbar   # → tests/golden/g2o_*.npz, provenance "g2o", + report vs oracle
//
// Two ways to get the vertex / edge types:
//   default                 : the types below — written from the formulas of corelib/include/Optimizer/g2o/OptimizeTypeDefine.h:16-225
//                             and corelib/src/Optimizer/g2o/OptimizeTypeDefine.cpp:7-14,35-88 (same operation order where
//                             it matters for rounding), self-contained;
//   -DVISFS_REFERENCE_TYPES : include the reference's own header instead (add -I<reference>/corelib/include
//                             -I<reference>/utilite/include and compile <reference>/corelib/src/Optimizer/g2o/OptimizeTypeDefine.cpp
//                             beside this file) — then every arithmetic instruction on the path is the reference's.
// The driver below follows Optimizer.cpp:75-97 (solver / algorithm choice), :100-223 (graph), :261-318 (two phases, chi2 guards,
// outlier pass) call for call.  Laser edges are outside the .vbag format.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <list>
#include <memory>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include <g2o/config.h>
#include <g2o/core/base_binary_edge.h>
#include <g2o/core/base_vertex.h>
#include <g2o/core/block_solver.h>
#include <g2o/core/optimization_algorithm_gauss_newton.h>
#include <g2o/core/optimization_algorithm_levenberg.h>
#include <g2o/core/robust_kernel_impl.h>
#include <g2o/core/sparse_optimizer.h>
#include <g2o/solvers/eigen/linear_solver_eigen.h>
#include <g2o/solvers/pcg/linear_solver_pcg.h>
#include <g2o/types/sba/types_sba.h>
#include <g2o/types/slam3d/se3quat.h>
#ifdef G2O_HAVE_CSPARSE
#include <g2o/solvers/csparse/linear_solver_csparse.h>
#endif
#ifdef G2O_HAVE_CHOLMOD
#include <g2o/solvers/cholmod/linear_solver_cholmod.h>
#endif

#ifdef VISFS_REFERENCE_TYPES
#include "Optimizer/g2o/OptimizeTypeDefine.h"
using PoseState = VISFS::Optimizer::CameraPose;
using PoseVertex = VISFS::Optimizer::VertexPose;
using StereoEdge = VISFS::Optimizer::EdgeStereo;
using OdometryEdge = VISFS::Optimizer::EdgePoseConstraint;
#else
namespace xcheck {

using Eigen::Matrix3d;
using Eigen::Quaterniond;
using Eigen::Vector3d;

static Matrix3d hat(const Vector3d& v) {
    Matrix3d m;
    m << 0.0, -v.z(), v.y(), v.z(), 0.0, -v.x(), -v.y(), v.x(), 0.0;
    return m;
}
static Quaterniond positive(Quaterniond q) {            // w >= 0, unit norm (Math.h:308-317)
    if (q.w() < 0.0) q.coeffs() *= -1.0;
    q.normalize();
    return q;
}
// 4x4 left / right product matrices of a (positified) quaternion in [w; x y z] order (Math.h:324-345)
static Eigen::Matrix4d quatLeft(const Quaterniond& qin) {
    const Quaterniond q = positive(qin);
    Eigen::Matrix4d L;
    L(0, 0) = q.w();
    L.block<1, 3>(0, 1) = -q.vec().transpose();
    L.block<3, 1>(1, 0) = q.vec();
    L.block<3, 3>(1, 1) = q.w() * Matrix3d::Identity() + hat(q.vec());
    return L;
}
static Eigen::Matrix4d quatRight(const Quaterniond& qin) {
    const Quaterniond q = positive(qin);
    Eigen::Matrix4d R;
    R(0, 0) = q.w();
    R.block<1, 3>(0, 1) = -q.vec().transpose();
    R.block<3, 1>(1, 0) = q.vec();
    R.block<3, 3>(1, 1) = q.w() * Matrix3d::Identity() - hat(q.vec());
    return R;
}

// T_cw as translation + unit quaternion with w >= 0 at construction (OptimizeTypeDefine.h:16-86)
class PoseState {
public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    PoseState() : t_(Vector3d::Zero()), q_(Quaterniond::Identity()) {}
    PoseState(const Quaterniond& q, const Vector3d& t) : t_(t), q_(q) { q_ = positive(q_); }
    PoseState(const Matrix3d& R, const Vector3d& t) : t_(t), q_(R) { q_ = positive(q_); }
    // boxplus (OptimizeTypeDefine.cpp:7-14 with deltaQ, Math.h:277-287): t += d[0:3]; q = normalize((1, d[3:6] / 2) * q)
    void update(const double* d) {
        t_ += Vector3d(d[0], d[1], d[2]);
        Vector3d half(d[3], d[4], d[5]);
        half /= 2.0;
        Quaterniond dq;
        dq.w() = 1.0; dq.x() = half.x(); dq.y() = half.y(); dq.z() = half.z();
        q_ = dq * q_;
        q_.normalize();                                      // not re-positified
    }
    Vector3d map(const Vector3d& pw) const { return q_.toRotationMatrix() * pw + t_; }
    const Quaterniond& getRotation() const { return q_; }
    const Vector3d& getTranslation() const { return t_; }
private:
    Vector3d t_;
    Quaterniond q_;
};

class PoseVertex : public g2o::BaseVertex<6, PoseState> {
public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    void setToOriginImpl() override { _estimate = PoseState(); }
    void oplusImpl(const double* u) override { _estimate.update(u); updateCache(); }
};

// 3-D stereo residual (u_l, v_l, u_r), vertex 0 = landmark, vertex 1 = pose (OptimizeTypeDefine.h:111-191)
class StereoEdge : public g2o::BaseBinaryEdge<3, Vector3d, g2o::VertexPointXYZ, PoseVertex> {
public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    Vector3d project(const Vector3d& pc) const {
        const double invZ = 1.0 / pc[2];
        Vector3d r;
        r[0] = pc[0] * invZ * fx + cx;
        r[1] = pc[1] * invZ * fy + cy;
        r[2] = r[0] - bf * invZ;
        return r;
    }
    void computeError() override {
        const PoseVertex* pose = static_cast<const PoseVertex*>(_vertices[1]);
        const g2o::VertexPointXYZ* point = static_cast<const g2o::VertexPointXYZ*>(_vertices[0]);
        _error = Vector3d(_measurement) - project(pose->estimate().map(point->estimate()));
    }
    void linearizeOplus() override {
        const PoseVertex* pose = static_cast<const PoseVertex*>(_vertices[1]);
        const g2o::VertexPointXYZ* point = static_cast<const g2o::VertexPointXYZ*>(_vertices[0]);
        const Vector3d pc = pose->estimate().map(point->estimate());
        const Matrix3d R = pose->estimate().getRotation().toRotationMatrix();
        const double x = pc[0], y = pc[1], z = pc[2], z2 = z * z;
        // d e / d P_w = -d(pi)/d(P_c) R, row by row with the divisions of the original (:145-155)
        for (int c = 0; c < 3; ++c) {
            _jacobianOplusXi(0, c) = -fx * R(0, c) / z + fx * x * R(2, c) / z2;
            _jacobianOplusXi(1, c) = -fy * R(1, c) / z + fy * y * R(2, c) / z2;
            _jacobianOplusXi(2, c) = _jacobianOplusXi(0, c) - bf * R(2, c) / z2;
        }
        // d e / d (dt, dtheta): -d(pi)/d(P_c) [I | -[P_c]x]  — the SE(3)-left form in P_c, as written at :157-176
        _jacobianOplusXj(0, 0) = -1. / z * fx;
        _jacobianOplusXj(0, 1) = 0.;
        _jacobianOplusXj(0, 2) = x / z2 * fx;
        _jacobianOplusXj(0, 3) = x * y / z2 * fx;
        _jacobianOplusXj(0, 4) = -(1. + (x * x / z2)) * fx;
        _jacobianOplusXj(0, 5) = y / z * fx;
        _jacobianOplusXj(1, 0) = 0.;
        _jacobianOplusXj(1, 1) = -1. / z * fy;
        _jacobianOplusXj(1, 2) = y / z2 * fy;
        _jacobianOplusXj(1, 3) = (1. + y * y / z2) * fy;
        _jacobianOplusXj(1, 4) = -x * y / z2 * fy;
        _jacobianOplusXj(1, 5) = -x / z * fy;
        _jacobianOplusXj(2, 0) = _jacobianOplusXj(0, 0);
        _jacobianOplusXj(2, 1) = 0.;
        _jacobianOplusXj(2, 2) = _jacobianOplusXj(0, 2) - bf / z2;
        _jacobianOplusXj(2, 3) = _jacobianOplusXj(0, 3) - bf * y / z2;
        _jacobianOplusXj(2, 4) = _jacobianOplusXj(0, 4) + bf * x / z2;
        _jacobianOplusXj(2, 5) = _jacobianOplusXj(0, 5);
    }
};

// 6-D relative-pose residual between two T_cw states, measurement T_c1c2 (OptimizeTypeDefine.h:193-225, .cpp:35-88)
class OdometryEdge : public g2o::BaseBinaryEdge<6, g2o::SE3Quat, PoseVertex, PoseVertex> {
public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    void computeError() override {
        const PoseVertex* a = static_cast<const PoseVertex*>(_vertices[0]);
        const PoseVertex* b = static_cast<const PoseVertex*>(_vertices[1]);
        const Vector3d P1 = a->estimate().getTranslation(), P2 = b->estimate().getTranslation();
        const Quaterniond Q1 = a->estimate().getRotation(), Q2 = b->estimate().getRotation();
        const Vector3d mP = _measurement.translation();
        const Quaterniond mQ = _measurement.rotation();
        _error.block<3, 1>(0, 0) = Q1 * Q2.inverse() * (-P2) + P1 - mP;
        _error.block<3, 1>(3, 0) = 2 * (mQ.inverse() * Q1 * Q2.inverse()).vec();
    }
    void linearizeOplus() override {
        const PoseVertex* a = static_cast<const PoseVertex*>(_vertices[0]);
        const PoseVertex* b = static_cast<const PoseVertex*>(_vertices[1]);
        const Vector3d P2 = b->estimate().getTranslation();
        const Quaterniond Q1 = a->estimate().getRotation(), Q2 = b->estimate().getRotation();
        const Quaterniond mQ = _measurement.rotation();
        _jacobianOplusXi.setZero();
        _jacobianOplusXi.block<3, 3>(0, 0) = Matrix3d::Identity();
        _jacobianOplusXi.block<3, 3>(0, 3) = -hat(Q1 * (Q2.inverse() * (-P2)));
        _jacobianOplusXi.block<3, 3>(3, 3) = (quatLeft(Q2 * Q1.inverse()) * quatRight(mQ)).bottomRightCorner<3, 3>();
        _jacobianOplusXj.setZero();
        _jacobianOplusXj.block<3, 3>(0, 0) = -(Q1 * Q2.inverse()).toRotationMatrix();
        _jacobianOplusXj.block<3, 3>(0, 3) = Q1.toRotationMatrix() * Q2.inverse().toRotationMatrix() * hat(-P2);
        _jacobianOplusXj.block<3, 3>(3, 3) = -(quatLeft(mQ.inverse() * Q1 * Q2.inverse())).bottomRightCorner<3, 3>();
    }
};

}  // namespace xcheck
using PoseState = xcheck::PoseState;
using PoseVertex = xcheck::PoseVertex;
using StereoEdge = xcheck::StereoEdge;
using OdometryEdge = xcheck::OdometryEdge;
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
    if (argc < 3) { std::fprintf(stderr, "usage: %s graph.vbag result.vbar [g2o-version-string]\n", argv[0]); return 2; }
    Dump d;
    if (!load(argv[1], d)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    if (d.framework != 0) { std::fprintf(stderr, "only Optimizer/Framework=0 (g2o)\n"); return 2; }

    // status codes of include/visfs_ba.h
    enum { OK = 0, ERR_NAN_CHI2 = 3, ERR_HUGE_CHI2_1 = 4, ERR_HUGE_CHI2_2 = 5 };
    int status = OK, iters[2] = { 0, 0 }, n_outliers = 0;
    double chi2_initial = 0.0, chi2_phase1 = 0.0, chi2_final = 0.0;

    g2o::SparseOptimizer optimizer;                                                    // Optimizer.cpp:75
    using PoseMatrix = g2o::BlockSolver_6_3::PoseMatrixType;
    std::unique_ptr<g2o::BlockSolver_6_3::LinearSolverType> linear;                     // :76-91
    std::string solver_name;
    if (d.solver == 3) { linear = std::make_unique<g2o::LinearSolverEigen<PoseMatrix>>(); solver_name = "LinearSolverEigen"; }
    else if (d.solver == 2) { linear = std::make_unique<g2o::LinearSolverPCG<PoseMatrix>>(); solver_name = "LinearSolverPCG"; }
#ifdef G2O_HAVE_CHOLMOD
    else if (d.solver == 1) { linear = std::make_unique<g2o::LinearSolverCholmod<PoseMatrix>>(); solver_name = "LinearSolverCholmod"; }
#endif
#ifdef G2O_HAVE_CSPARSE
    else if (d.solver == 0) { linear = std::make_unique<g2o::LinearSolverCSparse<PoseMatrix>>(); solver_name = "LinearSolverCSparse"; }
#endif
    if (!linear) { std::fprintf(stderr, "this g2o build has no linear solver for Optimizer/Solver=%d\n", d.solver); return 2; }
    auto block = std::make_unique<g2o::BlockSolver_6_3>(std::move(linear));
    if (d.trust_region == 0) optimizer.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(std::move(block)));   // :93-97
    else optimizer.setAlgorithm(new g2o::OptimizationAlgorithmGaussNewton(std::move(block)));

    // pose vertices: id = index + 1 (ids must be > 0 and ascend with the index: buildIndexMapping orders by id)   :100-114
    std::vector<PoseVertex*> poses(d.n_poses);
    for (int i = 0; i < d.n_poses; ++i) {
        const double* tq = d.pose_tq.data() + 7 * (size_t)i;
        PoseVertex* v = new PoseVertex();
        v->setEstimate(PoseState(Eigen::Quaterniond(tq[6], tq[3], tq[4], tq[5]), Eigen::Vector3d(tq[0], tq[1], tq[2])));
        v->setId(i + 1);
        v->setFixed(d.pose_fixed[i] != 0);
        optimizer.addVertex(v);
        poses[i] = v;
    }
    // odometry edges (:117-150): vertex 0 = from, vertex 1 = to, Omega = I6 / odometryCovariance, no robust kernel
    Eigen::Matrix<double, 6, 6> infoOdo = Eigen::Matrix<double, 6, 6>::Zero();
    for (int i = 0; i < 6; ++i) infoOdo(i, i) = 1. / d.odometry_covariance;
    for (int e = 0; e < d.n_odo; ++e) {
        const double* m = d.odo_tq.data() + 7 * (size_t)e;
        OdometryEdge* edge = new OdometryEdge();
        edge->setVertex(0, poses[d.odo_from[e]]);
        edge->setVertex(1, poses[d.odo_to[e]]);
        edge->setMeasurement(g2o::SE3Quat(Eigen::Quaterniond(m[6], m[3], m[4], m[5]), Eigen::Vector3d(m[0], m[1], m[2])));
        edge->setInformation(infoOdo);
        if (!optimizer.addEdge(edge)) { delete edge; std::fprintf(stderr, "addEdge failed (odometry %d)\n", e); return 1; }
    }
    // landmarks + stereo edges (:153-223): observations are sorted by (point, pose) — the reference's insertion order
    const Eigen::Matrix3d pixelInfo = Eigen::Matrix3d::Identity() / d.pixel_variance;
    const int stepVertexId = d.n_poses + 1;
    std::vector<g2o::VertexPointXYZ*> points(d.n_points, nullptr);
    std::vector<StereoEdge*> visual(d.n_obs, nullptr);
    std::vector<uint8_t> has_edge(d.n_points, 0);
    for (int k = 0; k < d.n_obs; ++k) has_edge[d.obs_point[k]] = 1;
    int k = 0;
    for (int l = 0; l < d.n_points; ++l) {
        // (the reference creates a vertex for every feature that has references AND a 3-D point; a landmark without any
        //  observation in the dump has no edge and stays out of the active set either way)
        g2o::VertexPointXYZ* v = new g2o::VertexPointXYZ();
        v->setEstimate(Eigen::Vector3d(d.point_xyz[3 * (size_t)l], d.point_xyz[3 * (size_t)l + 1], d.point_xyz[3 * (size_t)l + 2]));
        v->setId(stepVertexId + l);
        v->setFixed(d.point_fixed[l] != 0);
        v->setMarginalized(true);
        optimizer.addVertex(v);
        points[l] = v;
        for (; k < d.n_obs && d.obs_point[k] == l; ++k) {
            StereoEdge* es = new StereoEdge();
            es->setMeasurement(Eigen::Vector3d(d.obs_uvr[3 * (size_t)k], d.obs_uvr[3 * (size_t)k + 1], d.obs_uvr[3 * (size_t)k + 2]));
            es->setInformation(pixelInfo);
            es->fx = d.fx; es->fy = d.fy; es->cx = d.cx; es->cy = d.cy; es->bf = d.bf;
            es->setVertex(0, v);
            es->setVertex(1, poses[d.obs_pose[k]]);
            if (d.robust_kernel_delta > 0.0) {
                g2o::RobustKernelHuber* kernel = new g2o::RobustKernelHuber;
                kernel->setDelta(d.robust_kernel_delta);
                es->setRobustKernel(kernel);
            }
            optimizer.addEdge(es);
            visual[k] = es;
        }
    }

    std::vector<uint8_t> outlier(d.n_obs, 0);
    std::vector<double> edge_chi2(d.n_obs, 0.0);
    optimizer.setVerbose(false);                                                         // :261-265
    optimizer.initializeOptimization();
    optimizer.computeActiveErrors();
    chi2_initial = optimizer.activeRobustChi2();
    iters[0] = optimizer.optimize(d.iterations / 2);
    optimizer.computeActiveErrors();                                                     // :270-280
    double chi2 = optimizer.activeRobustChi2();
    chi2_phase1 = chi2_final = chi2;
    if (std::isnan(chi2)) status = ERR_NAN_CHI2;
    else if (chi2 > 1000000000000.0 || !std::isfinite(chi2)) status = ERR_HUGE_CHI2_1;
    if (status == OK) {
        // an edge whose two vertices are fixed is not in the active set: its _error is never computed — report 0 for it
        for (int q = 0; q < d.n_obs; ++q)
            if (visual[q] && !(d.pose_fixed[d.obs_pose[q]] && d.point_fixed[d.obs_point[q]])) edge_chi2[q] = visual[q]->chi2();
        if (d.robust_kernel_delta > 0.0) {                                               // :283-312
            for (int q = 0; q < d.n_obs; ++q) {
                StereoEdge* e = visual[q];
                if (!e || (d.pose_fixed[d.obs_pose[q]] && d.point_fixed[d.obs_point[q]])) continue;
                if (e->level() == 0 && e->chi2() > e->robustKernel()->delta()) { e->setLevel(1); outlier[q] = 1; ++n_outliers; }
            }
            optimizer.initializeOptimization(0);
            iters[1] = optimizer.optimize(d.iterations / 2);
        }
        chi2_final = optimizer.activeRobustChi2();                                       // :315-318
        if (chi2_final > 1000000000000.0) status = ERR_HUGE_CHI2_2;
    }

    std::vector<double> pose_out(7 * (size_t)d.n_poses), point_out(3 * (size_t)d.n_points);
    for (int i = 0; i < d.n_poses; ++i) {
        const Eigen::Vector3d t = poses[i]->estimate().getTranslation();
        const Eigen::Quaterniond q = poses[i]->estimate().getRotation();
        double* o = pose_out.data() + 7 * (size_t)i;
        o[0] = t.x(); o[1] = t.y(); o[2] = t.z(); o[3] = q.x(); o[4] = q.y(); o[5] = q.z(); o[6] = q.w();
    }
    for (int l = 0; l < d.n_points; ++l) {
        const Eigen::Vector3d p = points[l]->estimate();
        point_out[3 * (size_t)l] = p.x(); point_out[3 * (size_t)l + 1] = p.y(); point_out[3 * (size_t)l + 2] = p.z();
    }
    std::string prov = std::string("g2o ") + (argc > 3 ? argv[3] : "(version not given)") + ", " + solver_name +
#ifdef VISFS_REFERENCE_TYPES
                       ", reference vertex/edge types (OptimizeTypeDefine.h)";
#else
                       ", re-typed vertex/edge types (tools/g2o_crosscheck.cpp)";
#endif
    std::ofstream f(argv[2], std::ios::binary);
    const uint32_t ver = 1;
    f.write("VISFSBAR", 8); wr(f, &ver, 1);
    const int32_t head[4] = { status, iters[0], iters[1], n_outliers };
    wr(f, head, 4);
    const double chis[3] = { chi2_initial, chi2_phase1, chi2_final };
    wr(f, chis, 3);
    const int32_t sizes[3] = { d.n_poses, d.n_points, d.n_obs };
    wr(f, sizes, 3);
    wr(f, pose_out.data(), pose_out.size()); wr(f, point_out.data(), point_out.size());
    wr(f, outlier.data(), outlier.size()); wr(f, edge_chi2.data(), edge_chi2.size());
    const uint32_t n = (uint32_t)prov.size();
    wr(f, &n, 1); f.write(prov.data(), n);
    std::printf("%s: status %d, iterations %d + %d, outliers %d, chi2 %.6g -> %.6g -> %.6g\n", argv[1], status, iters[0], iters[1],
                n_outliers, chi2_initial, chi2_phase1, chi2_final);
    return f ? 0 : 1;
}
