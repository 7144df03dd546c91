// Optimizer.cpp — packs the reference's std::map arguments into the C ABI and unpacks the results.
// Mirrors the boundary behaviour of corelib/src/Optimizer/Optimizer.cpp:58-364 (g2o branch); every arithmetic
// step of that function happens behind visfs_ba_solve_window on the GPU.
#include "Optimizer.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/visfs_ba.h"

namespace VISFS {
namespace Optimizer {

namespace {
// Parameters::parse for the keys the optimiser reads (Optimizer.cpp:47-54): string → typed value, default kept when absent.
void parseInt(const ParametersMap& p, const char* key, int& v) {
    auto it = p.find(key);
    if (it != p.end()) v = std::atoi(it->second.c_str());
}
void parseDouble(const ParametersMap& p, const char* key, double& v) {
    auto it = p.find(key);
    if (it != p.end()) v = std::atof(it->second.c_str());
}
void toRowMajor3x4(const Eigen::Isometry3d& T, double* o) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) o[4 * r + c] = T(r, c);
}
}  // namespace

Optimizer::Optimizer(const ParametersMap& _parameters) :
    framework_(0), solver_(0), trustRegion_(0), iterations_(10),          // Parameters.h:184-187
    pixelVariance_(1.5), odometryCovariance_(0.00005), laserCovariance_(0.1), robustKernelDelta_(8.0),   // :188-191
    device_(0), handle_(nullptr), lastStatus_(VISFS_BA_OK) {
    parseInt(_parameters, "Optimizer/Framework", framework_);
    parseInt(_parameters, "Optimizer/Solver", solver_);
    parseInt(_parameters, "Optimizer/TrustRegion", trustRegion_);
    parseInt(_parameters, "Optimizer/Iterations", iterations_);
    parseDouble(_parameters, "Optimizer/PixelVariance", pixelVariance_);
    parseDouble(_parameters, "Optimizer/OdometryCovariance", odometryCovariance_);
    parseDouble(_parameters, "Optimizer/LaserCovariance", laserCovariance_);
    parseDouble(_parameters, "Optimizer/RobustKernelDelta", robustKernelDelta_);
    parseInt(_parameters, "Optimizer/Device", device_);
    visfs_ba_params prm;
    prm.framework = framework_; prm.solver = solver_; prm.trust_region = trustRegion_; prm.iterations = iterations_;
    prm.pixel_variance = pixelVariance_; prm.odometry_covariance = odometryCovariance_;
    prm.laser_covariance = laserCovariance_; prm.robust_kernel_delta = robustKernelDelta_;
    lastStatus_ = visfs_ba_create(&prm, device_, &handle_);
    if (lastStatus_ != VISFS_BA_OK) {
        handle_ = nullptr;
        createError_ = visfs_ba_create_error();          // missing device, wrong architecture or an allocation failure: say which
        std::fprintf(stderr, "VISFS::Optimizer (MI355X backend): cannot open device %d (status %d): %s — localOptimize will fail (there is no CPU fallback)\n",
                     device_, lastStatus_, createError_.c_str());
    }
}

Optimizer::~Optimizer() {
    if (handle_) visfs_ba_destroy(handle_);
}

const char* Optimizer::lastError() const { return handle_ ? visfs_ba_last_error(handle_) : createError_.c_str(); }

std::map<std::size_t, Eigen::Isometry3d> Optimizer::localOptimize(
    std::size_t _rootId,
    const std::map<std::size_t, Eigen::Isometry3d>& _poses,
    const std::map<std::size_t, std::tuple<std::size_t, std::size_t, Eigen::Isometry3d>>& _links,
    const std::vector<std::shared_ptr<GeometricCamera>>& _cameraModels,
    std::map<std::size_t, std::tuple<Eigen::Vector3d, bool>>& _points3D,
    const std::map<std::size_t, std::map<std::size_t, FeatureBA>>& _wordReferences,
    const std::vector<Sensor::PointCloud>& _pointClouds,
    const std::shared_ptr<const Map::Submap2D>& _submap,
    std::vector<std::tuple<std::size_t, std::size_t>>& _outliers) {

    std::map<std::size_t, Eigen::Isometry3d> optimizedPoses;
    if (_cameraModels.empty() || !handle_) {                // the reference asserts cameraModels.size() >= 1 (Optimizer.cpp:69)
        lastStatus_ = handle_ ? VISFS_BA_ERR_BAD_ARGUMENT : VISFS_BA_ERR_DEVICE;
        return optimizedPoses;
    }
    // ---- flatten (std::map iteration order == the reference's vertex / edge insertion order)
    std::vector<uint64_t> poseIds, linkFrom, linkTo, pointIds, refFeature, refPose;
    std::vector<double> poseTwr, linkT, pointXyz;
    std::vector<uint8_t> pointFixed;
    std::vector<float> refU, refV, refDepth;
    poseIds.reserve(_poses.size()); poseTwr.resize(_poses.size() * 12);
    std::size_t n = 0;
    for (auto it = _poses.begin(); it != _poses.end(); ++it, ++n) { poseIds.push_back(it->first); toRowMajor3x4(it->second, &poseTwr[12 * n]); }
    linkT.resize(_links.size() * 12);
    n = 0;
    for (auto it = _links.begin(); it != _links.end(); ++it, ++n) {
        linkFrom.push_back(std::get<0>(it->second)); linkTo.push_back(std::get<1>(it->second));
        toRowMajor3x4(std::get<2>(it->second), &linkT[12 * n]);
    }
    pointXyz.resize(_points3D.size() * 3);
    n = 0;
    for (auto it = _points3D.begin(); it != _points3D.end(); ++it, ++n) {
        pointIds.push_back(it->first);
        const Eigen::Vector3d& p = std::get<0>(it->second);
        pointXyz[3 * n] = p[0]; pointXyz[3 * n + 1] = p[1]; pointXyz[3 * n + 2] = p[2];
        pointFixed.push_back(std::get<1>(it->second) ? 1 : 0);
    }
    for (auto it = _wordReferences.begin(); it != _wordReferences.end(); ++it)
        for (auto jt = it->second.begin(); jt != it->second.end(); ++jt) {
            refFeature.push_back(it->first); refPose.push_back(jt->first);
            refU.push_back(jt->second.kpt.pt.x); refV.push_back(jt->second.kpt.pt.y); refDepth.push_back(jt->second.depth);
        }
    // laser occupied-space factor (Optimizer.cpp:225-258): every point of every cloud, and the sub-map's grid as the
    // edge reads it — limits + getCorrespondenceCost(Array2i(x, y)) per cell (TypeOccupiedSpace2D.h:28-37)
    std::vector<double> laserXyz;
    std::vector<float> gridCost;
    visfs_ba_grid grid;
    std::memset(&grid, 0, sizeof(grid));
    const bool withLaser = !_pointClouds.empty() && _submap != nullptr && _submap->getGrid() != nullptr;
    if (withLaser) {
        for (const auto& pc : _pointClouds)
            for (const auto& pt : pc.points()) { laserXyz.push_back(pt.position[0]); laserXyz.push_back(pt.position[1]); laserXyz.push_back(pt.position[2]); }
        const auto* g2 = _submap->getGrid();
        const auto& lim = g2->limits();
        grid.resolution = lim.resolution(); grid.max_x = lim.max().x(); grid.max_y = lim.max().y();
        grid.num_x_cells = lim.cellLimits().numXcells; grid.num_y_cells = lim.cellLimits().numYcells;
        gridCost.resize(static_cast<std::size_t>(grid.num_x_cells) * grid.num_y_cells);
        for (int y = 0; y < grid.num_y_cells; ++y)
            for (int x = 0; x < grid.num_x_cells; ++x)
                gridCost[static_cast<std::size_t>(grid.num_x_cells) * y + x] = g2->getCorrespondenceCost(Eigen::Array2i(x, y));
        grid.correspondence_cost = gridCost.data();
    }

    const GeometricCamera& cam = *_cameraModels.front();
    const Eigen::Matrix3d K = cam.eigenKdouble();
    visfs_ba_window w;
    std::memset(&w, 0, sizeof(w));
    w.root_id = _rootId;
    w.n_poses = static_cast<int32_t>(poseIds.size()); w.pose_ids = poseIds.data(); w.pose_Twr = poseTwr.data();
    w.n_links = static_cast<int32_t>(linkFrom.size()); w.link_from = linkFrom.data(); w.link_to = linkTo.data(); w.link_T = linkT.data();
    w.n_cameras = static_cast<int32_t>(_cameraModels.size());
    w.fx = K(0, 0); w.fy = K(1, 1); w.cx = K(0, 2); w.cy = K(1, 2);
    w.baseline = cam.getBaseLine();
    toRowMajor3x4(cam.getTansformImageToRobot(), w.Trc);
    w.n_points = static_cast<int32_t>(pointIds.size()); w.point_ids = pointIds.data(); w.point_xyz = pointXyz.data(); w.point_fixed = pointFixed.data();
    w.n_refs = static_cast<int32_t>(refFeature.size()); w.ref_feature = refFeature.data(); w.ref_pose = refPose.data();
    w.ref_u = refU.data(); w.ref_v = refV.data(); w.ref_depth = refDepth.data();
    if (withLaser && !laserXyz.empty()) { w.n_laser_points = static_cast<int32_t>(laserXyz.size() / 3); w.laser_xyz = laserXyz.data(); w.grid = &grid; }

    std::vector<uint64_t> outIds(poseIds.size() + 1), outFeat(refFeature.size() + 1), outPose(refFeature.size() + 1);
    std::vector<double> outTwr((poseIds.size() + 1) * 12);
    visfs_ba_result r;
    std::memset(&r, 0, sizeof(r));
    r.pose_ids_out = outIds.data(); r.pose_Twr_out = outTwr.data();
    r.outlier_capacity = static_cast<int32_t>(refFeature.size() + 1);
    r.outlier_feature = outFeat.data(); r.outlier_pose = outPose.data();

    lastStatus_ = visfs_ba_solve_window(handle_, &w, &r);

    // ---- unpack: the reference appends outliers at :296 even when phase 2 aborts, and returns an empty map on failure
    for (int i = 0; i < r.n_outliers; ++i) _outliers.emplace_back(static_cast<std::size_t>(outFeat[i]), static_cast<std::size_t>(outPose[i]));
    if (lastStatus_ != VISFS_BA_OK && lastStatus_ != VISFS_BA_PASSTHROUGH) {
        std::fprintf(stderr, "VISFS::Optimizer (MI355X backend): localOptimize failed, status %d: %s\n", lastStatus_, visfs_ba_last_error(handle_));
        return optimizedPoses;
    }
    for (int i = 0; i < r.n_poses_out; ++i) {
        Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 4; ++c) T(rr, c) = outTwr[12 * i + 4 * rr + c];
        optimizedPoses.emplace(static_cast<std::size_t>(outIds[i]), T);
    }
    if (lastStatus_ == VISFS_BA_OK) {                          // Optimizer.cpp:343-358 (already applied to pointXyz by the library)
        n = 0;
        for (auto it = _points3D.begin(); it != _points3D.end(); ++it, ++n)
            it->second = std::make_tuple(Eigen::Vector3d(pointXyz[3 * n], pointXyz[3 * n + 1], pointXyz[3 * n + 2]), std::get<1>(it->second));
    }
    return optimizedPoses;
}

}  // namespace Optimizer
}  // namespace VISFS
