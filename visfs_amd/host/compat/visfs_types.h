// visfs_types.h — minimal stand-ins for the third-party / VISFS types that appear in the signature of
// VISFS::Optimizer::Optimizer::localOptimize (reference: corelib/include/Optimizer/Optimizer.h:29-73).
//
// This image has no Eigen, OpenCV or VISFS headers, so the shim (../Optimizer.h) is compiled and tested against
// these stand-ins.  They expose ONLY the members the shim touches — T(r,c) element access, Identity(), v[i],
// kpt.pt.x/.y, eigenKdouble(), getBaseLine(), getTansformImageToRobot() — and every one of those spellings is
// valid for the real types too, so building inside VISFS with -DVISFS_BA_WITH_REFERENCE_HEADERS uses the real
// headers with the same shim source.  They are NOT a re-implementation of Eigen or OpenCV.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() : v{ 0, 0, 0 } {}
    Vector3d(double x, double y, double z) : v{ x, y, z } {}
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
};
struct Matrix3d {
    double m[3][3];
    Matrix3d() : m{ { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } } {}
    double& operator()(int r, int c) { return m[r][c]; }
    const double& operator()(int r, int c) const { return m[r][c]; }
};
struct Isometry3d {
    double m[4][4];
    Isometry3d() : m{ { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } } {}
    static Isometry3d Identity() { return Isometry3d(); }
    double& operator()(int r, int c) { return m[r][c]; }
    const double& operator()(int r, int c) const { return m[r][c]; }
};
}  // namespace Eigen

namespace cv {
struct Point2f { float x = 0.f, y = 0.f; };
struct KeyPoint {
    Point2f pt;
    float size = 1.f;
    KeyPoint() {}
    KeyPoint(float x, float y, float s) { pt.x = x; pt.y = y; size = s; }
};
}  // namespace cv

namespace VISFS {
typedef std::map<std::string, std::string> ParametersMap;   // corelib/include/Parameters.h

// corelib/include/CameraModels/GeometricCamera.h: the three accessors read at Optimizer.cpp:104,176,182
class GeometricCamera {
public:
    GeometricCamera() {
        // image (x right, y down, z forward) → robot (x forward, y left, z up): GeometricCamera.h:15-19
        T_ = Eigen::Isometry3d::Identity();
        T_(0, 0) = 0.0; T_(0, 2) = 1.0; T_(1, 0) = -1.0; T_(1, 1) = 0.0; T_(2, 1) = -1.0; T_(2, 2) = 0.0;
    }
    virtual ~GeometricCamera() {}
    virtual Eigen::Matrix3d eigenKdouble() const { return K_; }
    virtual float getBaseLine() const { return baseline_; }
    Eigen::Isometry3d getTansformImageToRobot() const { return T_; }
    void set(double fx, double fy, double cx, double cy, float baseline) {
        K_(0, 0) = fx; K_(1, 1) = fy; K_(0, 2) = cx; K_(1, 2) = cy; K_(2, 2) = 1.0; baseline_ = baseline;
    }
    void setTransformImageToRobot(const Eigen::Isometry3d& T) { T_ = T; }
private:
    Eigen::Matrix3d K_;
    float baseline_ = 0.f;
    Eigen::Isometry3d T_;
};

namespace Sensor {
struct RangefinderPoint { Eigen::Vector3d position; };
class PointCloud {   // corelib/include/Sensor/PointCloud.h — only emptiness / size is inspected by the shim
public:
    const std::vector<RangefinderPoint>& points() const { return pts_; }
    std::vector<RangefinderPoint> pts_;
};
}  // namespace Sensor
namespace Map { class Submap2D {}; }
}  // namespace VISFS
