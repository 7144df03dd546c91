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
struct Vector2d {
    double v[2];
    Vector2d() : v{ 0, 0 } {}
    Vector2d(double x, double y) : v{ x, y } {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
};
struct Array2i {
    int v[2];
    Array2i() : v{ 0, 0 } {}
    Array2i(int x, int y) : v{ x, y } {}
    int x() const { return v[0]; }
    int y() const { return v[1]; }
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
class PointCloud {   // corelib/include/Sensor/PointCloud.h — points() is what the laser factor iterates (Optimizer.cpp:235-236)
public:
    const std::vector<RangefinderPoint>& points() const { return pts_; }
    std::vector<RangefinderPoint> pts_;
};
}  // namespace Sensor
namespace Map {
// corelib/include/Map/2d/{xyIndex,MapLimits,Grid2d,Submap2D}.h — the accessors the laser factor reads through
// GridArrayAdapter (TypeOccupiedSpace2D.h:22-48) and Optimizer.cpp:229-231.
struct CellLimits { int numXcells = 0; int numYcells = 0; };
class MapLimits {
public:
    MapLimits() {}
    MapLimits(double resolution, const Eigen::Vector2d& max, const CellLimits& cells) : resolution_(resolution), max_(max), cells_(cells) {}
    double resolution() const { return resolution_; }
    const Eigen::Vector2d& max() const { return max_; }
    const CellLimits& cellLimits() const { return cells_; }
private:
    double resolution_ = 0.05;
    Eigen::Vector2d max_;
    CellLimits cells_;
};
class Grid2D {
public:
    Grid2D(const MapLimits& limits, std::vector<float> cost) : limits_(limits), cost_(std::move(cost)) {}
    const MapLimits& limits() const { return limits_; }
    float getCorrespondenceCost(const Eigen::Array2i& c) const {       // Grid2d.h:33-36 (the stand-in stores the floats)
        if (c.x() < 0 || c.y() < 0 || c.x() >= limits_.cellLimits().numXcells || c.y() >= limits_.cellLimits().numYcells) return 0.9f;
        return cost_[static_cast<std::size_t>(limits_.cellLimits().numXcells) * c.y() + c.x()];
    }
private:
    MapLimits limits_;
    std::vector<float> cost_;
};
class Submap2D {
public:
    Submap2D() {}
    explicit Submap2D(std::shared_ptr<Grid2D> grid) : grid_(std::move(grid)) {}
    const Grid2D* getGrid() const { return grid_.get(); }
private:
    std::shared_ptr<Grid2D> grid_;
};
}  // namespace Map
}  // namespace VISFS
