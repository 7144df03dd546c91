// WindowMap.h — the sliding window of VISFS's LocalMap (corelib/include/LocalMap.h, corelib/src/LocalMap.cpp) as flat
// arrays that feed the bundle-adjustment backend directly.
//
// SURVEY §8f rows f1 (LocalMap → flat-graph packer / un-packer) and f2 (wheel-odometry link generation).  The reference
// rebuilds four nested std::maps per frame (getSignaturePoses / getSignatureLinks / getFeaturePosesAndObservations,
// LocalMap.cpp:228-294) and the optimiser walks them again (Optimizer.cpp:100-223).  Here the window keeps its signatures
// and feature tracks in sorted flat vectors and emits the `visfs_ba_window` of include/visfs_ba.h in one pass.
// Only the BA-relevant state of LocalMap is mirrored (no images, no occupancy sub-maps); behaviour that decides what
// enters the graph or how results flow back is reproduced rule by rule and cited below.
#ifndef VISFS_AMD_WINDOW_MAP_H
#define VISFS_AMD_WINDOW_MAP_H

#include <cstddef>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/visfs_ba.h"

namespace VISFS {

// One tracked word of a new signature: key-point in the left image, its match in the right image and its 3-D position in
// the signature's ROBOT frame (Signature::getWords / getKeyPointsMatchesImageRight / getWords3d; floats as in OpenCV).
struct WordObservation {
    uint64_t featureId;
    float u, v;             // kpt.pt
    float uRight, vRight;   // right-image match
    float x, y, z;          // words3d (may be non-finite)
    bool has3d;             // the id is present in words3d
};

struct CovisibleWord { uint64_t featureId; float u, v; };   // Signature::getCovisibleWords (key-points in the former signature)

struct SignatureInput {
    uint64_t id;
    double pose[12];        // Twr, row-major 3x4 (Signature::getPose)
    double wheelOdom[12];   // wheel-odometry pose, all zeros = not available (the reference's zero-matrix sentinel)
    std::vector<WordObservation> words;          // ascending featureId (std::map order)
    std::vector<CovisibleWord> covisibleWords;   // ascending featureId
};

class WindowMap {
public:
    enum FeatureState { NEW_ADDED = 0, STABLE = 1 };   // Feature::eFeatureState, LocalMap.h:28-31

    // Keys read: LocalMap/MapSize, Tracker/MaxFeatures, LocalMap/MinParallax, LocalMap/MinTranslation, Estimator/MinInliers
    // (LocalMap.cpp:11-46, defaults Parameters.h:148,161-163,171).
    explicit WindowMap(const std::map<std::string, std::string>& parameters = std::map<std::string, std::string>());

    // LocalMap::insertSignature (LocalMap.cpp:48-131).  translation: motion since the previous signature.
    bool insertSignature(const SignatureInput& signature, const double translation[3]);
    // LocalMap::removeSignature (LocalMap.cpp:133-168).
    void removeSignature();
    // LocalMap::checkMapAvaliable (LocalMap.cpp:296-302).
    bool checkMapAvaliable() const;
    // LocalMap::updateLocalMap (LocalMap.cpp:170-226): poses, NEW_ADDED-only landmark write-back, outlier observations
    // erased, features to block (c1 && c2 && c3) appended to errorVertex.  Arrays as produced by the BA backend.
    void updateLocalMap(int nPoses, const uint64_t* poseIds, const double* poseTwr,
                        int nPoints, const uint64_t* pointIds, const double* pointXyz,
                        int nOutliers, const uint64_t* outlierFeature, const uint64_t* outlierPose,
                        std::set<uint64_t>& errorVertex);

    // The BA window in one pass: getSignaturePoses + getSignatureLinks (f2) + getFeaturePosesAndObservations
    // (LocalMap.cpp:228-294) straight into the flat arrays of the C ABI.  Trc: image→robot transform of the camera model,
    // intrinsics and baseline as the estimator would pass them (Estimator.cpp:227-254); rootId = newest id - 1 (:252).
    // withLinks mirrors `sensorStrategy_ >= 2` (Estimator.cpp:235-236).  The returned struct points into this object's
    // buffers and stays valid until the next mutating call.
    const visfs_ba_window& buildWindow(const double Trc[12], double fx, double fy, double cx, double cy, float baseline,
                                       int nCameras, bool withLinks);
    // Apply a solved window (visfs_ba_result + the in/out points of the window built last) — calls updateLocalMap.
    void applyResult(const visfs_ba_result& result, std::set<uint64_t>& errorVertex);

    // ---- introspection (tests, estimator bookkeeping)
    bool isKeySignature() const { return keySignature_; }
    std::size_t signatureCount() const { return sigIds_.size(); }
    std::size_t featureCount() const { return features_.size(); }
    struct FeatureView {
        uint64_t id, startSignature, endSignature; int state; double pose[3];
        std::vector<uint64_t> obsSignature; std::vector<float> obs;   // per observation: u v uRight vRight x y z
    };
    std::vector<uint64_t> signatureIds() const { return sigIds_; }
    void signaturePose(std::size_t index, double out[12]) const;
    std::vector<FeatureView> features() const;
    void counters(int& newFeatures, int& signatures, float& parallax, double translation[3]) const;

private:
    struct Obs { uint64_t sig; float u, v, uRight, vRight, x, y, z; };
    struct Feature {
        uint64_t id, startSig, endSig;
        int state;
        double pose[3];
        std::vector<Obs> obs;        // ascending signature id (<= MapSize + 1 entries)
    };
    int findFeature(uint64_t id) const;      // index in features_ or -1
    int findSignature(uint64_t id) const;
    void clearCounters();

    bool keySignature_;
    int localMapSize_, maxFeature_, minInliers_;
    float minParallax_;
    double minTranslation_;
    int newFeatureCount_, signatureCount_;
    float parallaxCount_;
    double translationCount_[3];

    std::vector<uint64_t> sigIds_;           // ascending
    std::vector<double> sigPose_, sigWheel_; // [n][12]
    std::vector<Feature> features_;          // ascending id

    // buffers behind the window returned by buildWindow
    visfs_ba_window window_;
    std::vector<uint64_t> wPoseIds_, wLinkFrom_, wLinkTo_, wPointIds_, wRefFeature_, wRefPose_;
    std::vector<double> wPoseTwr_, wLinkT_, wPointXyz_;
    std::vector<uint8_t> wPointFixed_;
    std::vector<float> wRefU_, wRefV_, wRefDepth_;
};

}  // namespace VISFS

#endif
