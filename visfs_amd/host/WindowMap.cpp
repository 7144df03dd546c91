// WindowMap.cpp — see WindowMap.h.  Every rule cites the reference line it reproduces (corelib/src/LocalMap.cpp).
#include "WindowMap.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace VISFS {

namespace {
void isoMul(const double* A, const double* B, double* C) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) C[4 * r + c] = A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c];
        C[4 * r + 3] = A[4 * r] * B[3] + A[4 * r + 1] * B[7] + A[4 * r + 2] * B[11] + A[4 * r + 3];
    }
}
void isoInv(const double* A, double* C) {   // Eigen Transform::inverse(Isometry): R^T, -R^T t
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[4 * r + c] = A[4 * c + r];
    for (int r = 0; r < 3; ++r) C[4 * r + 3] = -(C[4 * r] * A[3] + C[4 * r + 1] * A[7] + C[4 * r + 2] * A[11]);
}
bool isZeroPose(const double* T) {          // isApprox(Isometry3d(Matrix4d::Zero())) holds only for the exact zero sentinel
    for (int i = 0; i < 12; ++i) if (T[i] != 0.0) return false;
    return true;
}
template <typename T>
void parse(const std::map<std::string, std::string>& p, const char* key, T& v) {
    auto it = p.find(key);
    if (it != p.end()) v = static_cast<T>(std::atof(it->second.c_str()));
}
}  // namespace

WindowMap::WindowMap(const std::map<std::string, std::string>& p) :
    keySignature_(true), localMapSize_(5), maxFeature_(300), minInliers_(12), minParallax_(60.f),
    minTranslation_(3 * 0.5 * 0.5), newFeatureCount_(0), signatureCount_(0), parallaxCount_(0.f) {
    translationCount_[0] = translationCount_[1] = translationCount_[2] = 0.0;
    parse(p, "LocalMap/MapSize", localMapSize_);
    parse(p, "Tracker/MaxFeatures", maxFeature_);
    parse(p, "LocalMap/MinParallax", minParallax_);
    parse(p, "LocalMap/MinTranslation", minTranslation_);
    // LocalMap.cpp:34-35: the squared form is applied AFTER the parse — i.e. twice when the key is absent (reproduced)
    minTranslation_ = 3 * minTranslation_ * minTranslation_;
    parse(p, "Estimator/MinInliers", minInliers_);
    std::memset(&window_, 0, sizeof(window_));
}

int WindowMap::findFeature(uint64_t id) const {
    auto it = std::lower_bound(features_.begin(), features_.end(), id, [](const Feature& f, uint64_t v) { return f.id < v; });
    return (it != features_.end() && it->id == id) ? static_cast<int>(it - features_.begin()) : -1;
}

int WindowMap::findSignature(uint64_t id) const {
    auto it = std::lower_bound(sigIds_.begin(), sigIds_.end(), id);
    return (it != sigIds_.end() && *it == id) ? static_cast<int>(it - sigIds_.begin()) : -1;
}

void WindowMap::clearCounters() {   // LocalMap.cpp:320-328
    newFeatureCount_ = 0; signatureCount_ = 0; parallaxCount_ = 0.f;
    translationCount_[0] = translationCount_[1] = translationCount_[2] = 0.0;
}

bool WindowMap::insertSignature(const SignatureInput& s, const double translation[3]) {
    bool any3d = false;
    for (const auto& w : s.words) any3d |= w.has3d;
    if (!any3d) return false;                                                     // :49-52 (words3d empty)
    for (const auto& w : s.words) {                                               // :60 ascending feature id
        const int fi = findFeature(w.featureId);
        const Obs ob{ s.id, w.u, w.v, w.uRight, w.vRight, w.x, w.y, w.z };
        if (fi < 0) {
            if (static_cast<int>(features_.size()) > maxFeature_) {               // :64-67 only ids beyond the newest are admitted
                if (w.featureId <= features_.back().id) continue;
            }
            if (!w.has3d) continue;                                               // :68-69
            if (!(std::isfinite(w.x) && std::isfinite(w.y) && std::isfinite(w.z))) continue;   // :71-73
            Feature f;
            f.id = w.featureId; f.startSig = s.id; f.endSig = s.id; f.state = NEW_ADDED;
            const double px = w.x, py = w.y, pz = w.z;                            // :76 Twr * Vector3d(float→double)
            for (int r = 0; r < 3; ++r) f.pose[r] = s.pose[4 * r] * px + s.pose[4 * r + 1] * py + s.pose[4 * r + 2] * pz + s.pose[4 * r + 3];
            f.obs.push_back(ob);                                                  // :77
            features_.insert(std::lower_bound(features_.begin(), features_.end(), f.id, [](const Feature& a, uint64_t v) { return a.id < v; }), f);
            ++newFeatureCount_;                                                   // :79
        } else {
            Feature& f = features_[fi];
            auto it = std::lower_bound(f.obs.begin(), f.obs.end(), s.id, [](const Obs& o, uint64_t v) { return o.sig < v; });
            if (it == f.obs.end() || it->sig != s.id) f.obs.insert(it, ob);       // :82 emplace: keeps an existing entry
            f.endSig = s.id;                                                      // :83
            if (static_cast<int>(f.obs.size()) > localMapSize_ && f.state == NEW_ADDED) f.state = STABLE;   // :84-88
        }
    }
    // :92 signatures_.emplace (keeps an existing entry)
    if (findSignature(s.id) < 0) {
        const std::size_t pos = std::lower_bound(sigIds_.begin(), sigIds_.end(), s.id) - sigIds_.begin();
        sigIds_.insert(sigIds_.begin() + pos, s.id);
        sigPose_.insert(sigPose_.begin() + 12 * pos, s.pose, s.pose + 12);
        sigWheel_.insert(sigWheel_.begin() + 12 * pos, s.wheelOdom, s.wheelOdom + 12);
    }
    // ---- key-signature policy :94-126
    keySignature_ = false;
    ++signatureCount_;
    for (int k = 0; k < 3; ++k) translationCount_[k] += std::fabs(translation[k]);
    const double t2 = translationCount_[0] * translationCount_[0] + translationCount_[1] * translationCount_[1] + translationCount_[2] * translationCount_[2];
    if (newFeatureCount_ > 0.2 * maxFeature_) {
        keySignature_ = true; clearCounters();
    } else if (signatureCount_ > 10 && t2 > minTranslation_) {
        keySignature_ = true; clearCounters();
    } else {
        // parallax over the words present in both the former and the new signature (:110-119), float arithmetic
        float parallaxSum = 0.f;
        int parallaxNum = 0;
        std::size_t a = 0;
        for (const auto& w : s.words) {
            while (a < s.covisibleWords.size() && s.covisibleWords[a].featureId < w.featureId) ++a;
            if (a < s.covisibleWords.size() && s.covisibleWords[a].featureId == w.featureId) {
                const float du = s.covisibleWords[a].u - w.u, dv = s.covisibleWords[a].v - w.v;
                parallaxSum += std::max(0.f, static_cast<float>(std::sqrt(du * du + dv * dv)));
                ++parallaxNum;
            }
        }
        parallaxCount_ += (parallaxSum / static_cast<float>(parallaxNum));          // 0/0 = NaN when nothing matches, as the reference
        if (parallaxCount_ >= minParallax_) { keySignature_ = true; clearCounters(); }
    }
    return true;
}

void WindowMap::removeSignature() {
    const int n = static_cast<int>(sigIds_.size());
    if (n != localMapSize_ + 1) return;                                           // :134-141 (the > MapSize+1 case is a fatal log upstream)
    const uint64_t rmId = keySignature_ ? sigIds_.front() : sigIds_[n - 2];       // :144-148
    const uint64_t firstId = sigIds_.front();                                     // signatures_.begin() BEFORE the erase (:157)
    std::size_t out = 0;
    for (std::size_t i = 0; i < features_.size(); ++i) {
        Feature& f = features_[i];
        auto it = std::lower_bound(f.obs.begin(), f.obs.end(), rmId, [](const Obs& o, uint64_t v) { return o.sig < v; });
        if (it != f.obs.end() && it->sig == rmId) f.obs.erase(it);                // :150-152
        const bool drop = f.obs.empty() && (f.state == STABLE || f.endSig < firstId);   // :157-158
        if (!drop) { if (out != i) features_[out] = std::move(features_[i]); ++out; }
    }
    features_.resize(out);
    const int si = findSignature(rmId);                                           // :164
    sigIds_.erase(sigIds_.begin() + si);
    sigPose_.erase(sigPose_.begin() + 12 * si, sigPose_.begin() + 12 * si + 12);
    sigWheel_.erase(sigWheel_.begin() + 12 * si, sigWheel_.begin() + 12 * si + 12);
}

bool WindowMap::checkMapAvaliable() const {
    return !(sigIds_.size() < 2 || static_cast<int>(features_.size()) < minInliers_);   // :297
}

void WindowMap::updateLocalMap(int nPoses, const uint64_t* poseIds, const double* poseTwr,
                               int nPoints, const uint64_t* pointIds, const double* pointXyz,
                               int nOutliers, const uint64_t* outlierFeature, const uint64_t* outlierPose,
                               std::set<uint64_t>& errorVertex) {
    for (int i = 0; i < nPoses; ++i) {                                            // :171-177
        const int si = findSignature(poseIds[i]);
        if (si >= 0) std::memcpy(&sigPose_[12 * si], poseTwr + 12 * i, 96);
    }
    for (int i = 0; i < nPoints; ++i) {                                           // :178-188 only NEW_ADDED landmarks take the BA value
        const int fi = findFeature(pointIds[i]);
        if (fi >= 0 && features_[fi].state == NEW_ADDED) std::memcpy(features_[fi].pose, pointXyz + 3 * i, 24);
    }
    for (int i = 0; i < nOutliers; ++i) {                                         // :190-225
        const int fi = findFeature(outlierFeature[i]);
        if (fi < 0) continue;
        Feature& f = features_[fi];
        auto it = std::lower_bound(f.obs.begin(), f.obs.end(), outlierPose[i], [](const Obs& o, uint64_t v) { return o.sig < v; });
        if (it == f.obs.end() || it->sig != outlierPose[i]) continue;
        f.obs.erase(it);
        const bool c1 = f.obs.empty();
        const bool c2 = f.state == NEW_ADDED;
        // third-newest signature (:211-214); the reference dereferences it unconditionally — callers guarantee >= 3
        const bool c3 = sigIds_.size() >= 3 && f.startSig < sigIds_[sigIds_.size() - 3];
        if (c1 && c2 && c3) errorVertex.insert(f.id);
    }
}

const visfs_ba_window& WindowMap::buildWindow(const double Trc[12], double fx, double fy, double cx, double cy, float baseline,
                                              int nCameras, bool withLinks) {
    const std::size_t n = sigIds_.size();
    wPoseIds_ = sigIds_;                                                          // getSignaturePoses :228-236
    wPoseTwr_ = sigPose_;
    wLinkFrom_.clear(); wLinkTo_.clear(); wLinkT_.clear();
    if (withLinks && n >= 2) {                                                    // getSignatureLinks :238-272 (f2)
        for (std::size_t i = 0; i + 1 < n; ++i) {
            const double* from = &sigWheel_[12 * i];
            const double* to = &sigWheel_[12 * (i + 1)];
            if (!isZeroPose(from) && !isZeroPose(to)) {                           // :256 zero sentinel = no wheel odometry
                double inv[12], T[12];
                isoInv(from, inv);
                isoMul(inv, to, T);                                               // :257 fromPose.inverse() * toPose
                wLinkFrom_.push_back(sigIds_[i]); wLinkTo_.push_back(sigIds_[i + 1]);
                wLinkT_.insert(wLinkT_.end(), T, T + 12);
            }
        }
    }
    // getFeaturePosesAndObservations :274-294
    double Tcr[12];
    isoInv(Trc, Tcr);                                                             // :275 robot → image
    wPointIds_.clear(); wPointXyz_.clear(); wPointFixed_.clear();
    wRefFeature_.clear(); wRefPose_.clear(); wRefU_.clear(); wRefV_.clear(); wRefDepth_.clear();
    for (const Feature& f : features_) {
        if (f.obs.size() <= 1) continue;                                          // :277 observedTimes > 1
        wPointIds_.push_back(f.id);
        wPointXyz_.insert(wPointXyz_.end(), f.pose, f.pose + 3);
        wPointFixed_.push_back(f.state == STABLE ? 1 : 0);                        // :278
        for (const Obs& o : f.obs) {
            // :283 depth = float(z of (Tcr * [x y z 1])) — the product is formed in double from the float coordinates
            const double z = Tcr[8] * static_cast<double>(o.x) + Tcr[9] * static_cast<double>(o.y) + Tcr[10] * static_cast<double>(o.z) + Tcr[11];
            wRefFeature_.push_back(f.id); wRefPose_.push_back(o.sig);
            wRefU_.push_back(o.u); wRefV_.push_back(o.v); wRefDepth_.push_back(static_cast<float>(z));
        }
    }
    visfs_ba_window& w = window_;
    std::memset(&w, 0, sizeof(w));
    w.root_id = n ? sigIds_.back() - 1 : 0;                                       // Estimator.cpp:252
    w.n_poses = static_cast<int32_t>(n); w.pose_ids = wPoseIds_.data(); w.pose_Twr = wPoseTwr_.data();
    w.n_links = static_cast<int32_t>(wLinkFrom_.size()); w.link_from = wLinkFrom_.data(); w.link_to = wLinkTo_.data(); w.link_T = wLinkT_.data();
    w.n_cameras = nCameras; w.fx = fx; w.fy = fy; w.cx = cx; w.cy = cy; w.baseline = baseline;
    std::memcpy(w.Trc, Trc, 96);
    w.n_points = static_cast<int32_t>(wPointIds_.size()); w.point_ids = wPointIds_.data(); w.point_xyz = wPointXyz_.data(); w.point_fixed = wPointFixed_.data();
    w.n_refs = static_cast<int32_t>(wRefFeature_.size()); w.ref_feature = wRefFeature_.data(); w.ref_pose = wRefPose_.data();
    w.ref_u = wRefU_.data(); w.ref_v = wRefV_.data(); w.ref_depth = wRefDepth_.data();
    w.n_laser_points = 0;
    return w;
}

void WindowMap::applyResult(const visfs_ba_result& r, std::set<uint64_t>& errorVertex) {
    updateLocalMap(r.n_poses_out, r.pose_ids_out, r.pose_Twr_out,
                   window_.n_points, window_.point_ids, window_.point_xyz,
                   r.n_outliers, r.outlier_feature, r.outlier_pose, errorVertex);
}

void WindowMap::signaturePose(std::size_t index, double out[12]) const { std::memcpy(out, &sigPose_[12 * index], 96); }

std::vector<WindowMap::FeatureView> WindowMap::features() const {
    std::vector<FeatureView> v;
    for (const Feature& f : features_) {
        FeatureView fv;
        fv.id = f.id; fv.startSignature = f.startSig; fv.endSignature = f.endSig; fv.state = f.state;
        std::memcpy(fv.pose, f.pose, 24);
        for (const Obs& o : f.obs) {
            fv.obsSignature.push_back(o.sig);
            const float a[7] = { o.u, o.v, o.uRight, o.vRight, o.x, o.y, o.z };
            fv.obs.insert(fv.obs.end(), a, a + 7);
        }
        v.push_back(fv);
    }
    return v;
}

void WindowMap::counters(int& newFeatures, int& signatures, float& parallax, double translation[3]) const {
    newFeatures = newFeatureCount_; signatures = signatureCount_; parallax = parallaxCount_;
    std::memcpy(translation, translationCount_, 24);
}

}  // namespace VISFS
