"""TEST INFRASTRUCTURE — a pure-Python restatement of VISFS::Map::LocalMap (corelib/src/LocalMap.cpp) as far as the
bundle-adjustment path needs it.  Dicts stand for the reference's std::maps (iterated in sorted key order), numpy
float32 for cv::Point2f / Point3f / KeyPoint fields, Python floats for Eigen doubles.  PARITY UNPINNED: the reference
cannot be built here (Eigen / OpenCV are absent) and holds no fixtures for this class; each rule cites its source line.
Only tests/ may import this file."""
import math

import numpy as np

NEW_ADDED, STABLE = 0, 1
f32 = np.float32


def iso_inv(T):
    """Eigen::Isometry3d::inverse(): (R^T, -R^T t); T as 12 row-major doubles."""
    R = [[T[4 * c + r] for c in range(3)] for r in range(3)]
    t = [-(R[r][0] * T[3] + R[r][1] * T[7] + R[r][2] * T[11]) for r in range(3)]
    return [R[0][0], R[0][1], R[0][2], t[0], R[1][0], R[1][1], R[1][2], t[1], R[2][0], R[2][1], R[2][2], t[2]]


def iso_mul(A, B):
    C = [0.0] * 12
    for r in range(3):
        for c in range(3):
            C[4 * r + c] = A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c] + A[4 * r + 2] * B[8 + c]
        C[4 * r + 3] = A[4 * r] * B[3] + A[4 * r + 1] * B[7] + A[4 * r + 2] * B[11] + A[4 * r + 3]
    return C


class LocalMapOracle:
    def __init__(self, parameters=None):
        p = parameters or {}
        # LocalMap.cpp:11-46 (defaults Parameters.h:148,161-163,171)
        self.key_signature = True
        self.map_size = int(p.get("LocalMap/MapSize", 5))
        self.max_feature = int(p.get("Tracker/MaxFeatures", 300))
        self.min_parallax = f32(p.get("LocalMap/MinParallax", 60.0))
        mt = 3 * 0.5 * 0.5                                   # initialiser list :17
        if "LocalMap/MinTranslation" in p:
            mt = float(p["LocalMap/MinTranslation"])       # :34
        self.min_translation = 3 * mt * mt                   # :35 (applied to whatever :34 left)
        self.min_inliers = int(p.get("Estimator/MinInliers", 12))
        self.new_feature_count = 0
        self.signature_count = 0
        self.parallax_count = f32(0)
        self.translation_count = [0.0, 0.0, 0.0]
        self.signatures = {}        # id -> dict(pose, wheel)
        self.features = {}          # id -> dict(start, end, state, pose, obs{sig -> 7 float32})

    def _clear(self):               # :320-328
        self.new_feature_count = 0; self.signature_count = 0; self.parallax_count = f32(0); self.translation_count = [0.0, 0.0, 0.0]

    def insert(self, sig_id, pose, wheel, translation, words, right, words3d, covisible):
        """words / right / covisible: {id: (u, v)} float32; words3d: {id: (x, y, z)} float32.  LocalMap.cpp:48-131."""
        if len(words3d) == 0:
            return False
        for fid in sorted(words):
            u, v = words[fid]
            if fid not in self.features:
                if len(self.features) > self.max_feature and fid <= max(self.features):   # :64-67
                    continue
                if fid not in words3d:                                                    # :68-69
                    continue
                x, y, z = words3d[fid]
                if not (math.isfinite(x) and math.isfinite(y) and math.isfinite(z)):      # :71-73
                    continue
                px, py, pz = float(x), float(y), float(z)
                world = [pose[4 * r] * px + pose[4 * r + 1] * py + pose[4 * r + 2] * pz + pose[4 * r + 3] for r in range(3)]   # :76
                self.features[fid] = dict(start=sig_id, end=sig_id, state=NEW_ADDED, pose=world,
                                          obs={sig_id: np.array([u, v, *right[fid], x, y, z], f32)})
                self.new_feature_count += 1
            else:
                f = self.features[fid]
                x, y, z = words3d[fid]                                                    # .at(): the caller guarantees presence
                f["obs"].setdefault(sig_id, np.array([u, v, *right[fid], x, y, z], f32))  # :82 emplace
                f["end"] = sig_id                                                         # :83
                if len(f["obs"]) > self.map_size and f["state"] == NEW_ADDED:             # :84-88
                    f["state"] = STABLE
        self.signatures.setdefault(sig_id, dict(pose=list(pose), wheel=list(wheel)))      # :92
        self.key_signature = False                                                        # :95
        self.signature_count += 1
        for k in range(3):
            self.translation_count[k] += abs(translation[k])
        t2 = sum(c * c for c in self.translation_count)
        if self.new_feature_count > 0.2 * self.max_feature:                               # :100
            self.key_signature = True; self._clear()
        elif self.signature_count > 10 and t2 > self.min_translation:                     # :103
            self.key_signature = True; self._clear()
        else:
            psum = f32(0); pnum = 0                                                       # :106-119
            for fid in sorted(words):
                if fid in covisible:
                    du = f32(covisible[fid][0]) - f32(words[fid][0]); dv = f32(covisible[fid][1]) - f32(words[fid][1])
                    psum = f32(psum + max(f32(0), f32(np.sqrt(f32(f32(du * du) + f32(dv * dv))))))
                    pnum += 1
            with np.errstate(invalid="ignore", divide="ignore"):
                self.parallax_count = f32(self.parallax_count + f32(psum / f32(pnum)))    # 0/0 → NaN as in the reference
            if self.parallax_count >= self.min_parallax:                                  # :121
                self.key_signature = True; self._clear()
        return True

    def remove(self):               # :133-168
        if len(self.signatures) != self.map_size + 1:
            return
        ids = sorted(self.signatures)
        rm = ids[0] if self.key_signature else ids[-2]
        first = ids[0]
        for fid in sorted(self.features):
            f = self.features[fid]
            f["obs"].pop(rm, None)
            if len(f["obs"]) == 0 and (f["state"] == STABLE or f["end"] < first):
                del self.features[fid]
        del self.signatures[rm]

    def available(self):            # :296-302
        return not (len(self.signatures) < 2 or len(self.features) < self.min_inliers)

    def update(self, poses, points, outliers):
        """poses {id: 12 doubles}, points {id: xyz}, outliers [(feature, signature)] → errorVertex set.  :170-226."""
        err = set()
        for sid, T in poses.items():
            if sid in self.signatures:
                self.signatures[sid]["pose"] = list(T)
        for fid, xyz in points.items():
            if fid in self.features and self.features[fid]["state"] == NEW_ADDED:
                self.features[fid]["pose"] = list(xyz)
        ids = sorted(self.signatures)
        for fid, sid in outliers:
            f = self.features.get(fid)
            if f is None or sid not in f["obs"]:
                continue
            del f["obs"][sid]
            c1 = len(f["obs"]) == 0
            c2 = f["state"] == NEW_ADDED
            c3 = len(ids) >= 3 and f["start"] < ids[-3]
            if c1 and c2 and c3:
                err.add(fid)
        return err

    def poses(self):                # :228-236
        return {sid: list(self.signatures[sid]["pose"]) for sid in sorted(self.signatures)}

    def links(self):                # :238-272
        ids = sorted(self.signatures)
        out = {}
        for i in range(1, len(ids)):
            a, b = self.signatures[ids[i - 1]]["wheel"], self.signatures[ids[i]]["wheel"]
            if any(v != 0.0 for v in a) and any(v != 0.0 for v in b):
                out[i] = (ids[i - 1], ids[i], iso_mul(iso_inv(a), b))
        return out

    def points_and_observations(self, Trc):   # :274-294
        Tcr = iso_inv(list(Trc))
        points, obs = {}, {}
        for fid in sorted(self.features):
            f = self.features[fid]
            if len(f["obs"]) > 1:
                points[fid] = (list(f["pose"]), f["state"] == STABLE)
                pts = {}
                for sid in sorted(f["obs"]):
                    o = f["obs"][sid]
                    z = Tcr[8] * float(o[4]) + Tcr[9] * float(o[5]) + Tcr[10] * float(o[6]) + Tcr[11]
                    pts[sid] = (o[0], o[1], f32(z))
                obs[fid] = pts
        return points, obs
