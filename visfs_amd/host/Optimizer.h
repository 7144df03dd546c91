// Optimizer.h — drop-in for VISFS::Optimizer::Optimizer backed by the MI355X bundle-adjustment library.
//
// Keeps the reference's API surface (corelib/include/Optimizer/Optimizer.h:29-73): the same constructor taking a
// ParametersMap and the same localOptimize signature, argument meaning and error convention (empty map on failure,
// input poses when there is a single pose or Iterations <= 0, points3D updated in place, outliers appended).  The
// implementation (Optimizer.cpp) packs the std::map inputs into the flat C ABI of include/visfs_ba.h and calls
// visfs_ba_solve_window; there is no g2o / Ceres and no CPU solve path behind it.
#ifndef VISFS_AMD_OPTIMIZER_H
#define VISFS_AMD_OPTIMIZER_H

#include <cstddef>
#include <map>
#include <string>
#include <memory>
#include <tuple>
#include <vector>

#ifdef VISFS_BA_WITH_REFERENCE_HEADERS
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core/core.hpp>
#include "Parameters.h"
#include "CameraModels/GeometricCamera.h"
#include "Sensor/PointCloud.h"
#include "Map/2d/Submap2D.h"
#else
#include "compat/visfs_types.h"
#endif

struct visfs_ba_handle;

namespace VISFS {
namespace Optimizer {

// One observation of a feature in a signature: key-point (pixels) + depth along the camera z axis.
struct FeatureBA {
    cv::KeyPoint kpt;
    float depth;
    FeatureBA(const cv::KeyPoint& _kpt, const float _depth) : kpt(_kpt), depth(_depth) {}
};

class Optimizer {
public:
    // Reads Optimizer/{Framework,Solver,TrustRegion,Iterations,PixelVariance,OdometryCovariance,LaserCovariance,
    // RobustKernelDelta} with the reference defaults (Parameters.h:184-191).  Extra key: Optimizer/Device (HIP ordinal, default 0).
    Optimizer(const ParametersMap& _parameters = ParametersMap());
    ~Optimizer();
    Optimizer(const Optimizer&) = delete;
    Optimizer& operator=(const Optimizer&) = delete;

    std::map<std::size_t, Eigen::Isometry3d> localOptimize(
        std::size_t _rootId,                                                                                   // fixed pose
        const std::map<std::size_t, Eigen::Isometry3d>& _poses,                                                 // Twr per signature id
        const std::map<std::size_t, std::tuple<std::size_t, std::size_t, Eigen::Isometry3d>>& _links,           // (from, to, T_r1r2)
        const std::vector<std::shared_ptr<GeometricCamera>>& _cameraModels,                                     // [left, right]
        std::map<std::size_t, std::tuple<Eigen::Vector3d, bool>>& _points3D,                                    // in/out: world xyz, fixed?
        const std::map<std::size_t, std::map<std::size_t, FeatureBA>>& _wordReferences,                         // feature → pose → observation
        const std::vector<Sensor::PointCloud>& _pointClouds,
        const std::shared_ptr<const Map::Submap2D>& _submap,
        std::vector<std::tuple<std::size_t, std::size_t>>& _outliers);                                          // appended: (feature id, signature id)

    // status of the last call (VISFS_BA_* of include/visfs_ba.h) and the library's message
    int lastStatus() const { return lastStatus_; }
    const char* lastError() const;

private:
    int framework_, solver_, trustRegion_, iterations_;
    double pixelVariance_, odometryCovariance_, laserCovariance_, robustKernelDelta_;
    int device_;
    visfs_ba_handle* handle_;
    int lastStatus_;
    std::string createError_;      // why visfs_ba_create failed (visfs_ba_create_error), kept for lastError()
};

}  // namespace Optimizer
}  // namespace VISFS

#endif  // VISFS_AMD_OPTIMIZER_H
