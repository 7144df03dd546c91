"""Synthetic front end for the sliding-window tests: a camera moving through a point cloud, producing per-frame
signatures (words, right matches, robot-frame 3-D words, covisible words, wheel odometry) the way
Estimator::process hands them to LocalMap::insertSignature (corelib/src/Estimator.cpp:391-395)."""
import numpy as np

from visfs_amd import synth

f32 = np.float32


class FrontEnd:
    def __init__(self, seed=1, n_tracks=60, track_len=(2, 12), p_no3d=0.05, p_nan=0.05, p_no_wheel=0.15, step=0.12, pix_noise=0.3,
                 fx=420.0, fy=420.0, cx=320.0, cy=240.0, baseline=0.12):
        self.rng = np.random.default_rng(seed)
        self.n_tracks, self.track_len, self.p_no3d, self.p_nan, self.p_no_wheel = n_tracks, track_len, p_no3d, p_nan, p_no_wheel
        self.step, self.pix_noise = step, pix_noise
        self.fx, self.fy, self.cx, self.cy, self.baseline = fx, fy, cx, cy, f32(baseline)
        # image → robot: optical frame (z forward) → robot frame (x forward), a small lever arm
        self.Trc = np.array([0, 0, 1, 0.1, -1, 0, 0, 0.02, 0, -1, 0, 0.3], float)
        self.next_feature = 1
        self.next_sig = 1
        self.tracks = {}       # id -> dict(world xyz, remaining, no3d)
        self.prev_words = {}
        self.Twr = np.hstack([np.eye(3), np.zeros((3, 1))])
        self.wheel = self.Twr.copy()

    @staticmethod
    def _inv(T):
        R, t = T[:, :3], T[:, 3]
        return np.hstack([R.T, (-R.T @ t)[:, None]])

    def _spawn(self):
        # a point 2..8 m in front of the camera, inside the image
        z = self.rng.uniform(2, 8); u = self.rng.uniform(20, 620); v = self.rng.uniform(20, 460)
        pc = np.array([(u - self.cx) / self.fx * z, (v - self.cy) / self.fy * z, z])
        Trc = self.Trc.reshape(3, 4)
        pr = Trc[:, :3] @ pc + Trc[:, 3]
        pw = self.Twr[:, :3] @ pr + self.Twr[:, 3]
        fid = self.next_feature
        self.next_feature += int(self.rng.integers(1, 3))
        self.tracks[fid] = dict(world=pw, remaining=int(self.rng.integers(*self.track_len)), no3d=self.rng.random() < self.p_no3d)

    def frame(self):
        rng = self.rng
        # motion: forward along robot x with a little yaw
        yaw = rng.normal(0, 0.02); c, s = np.cos(yaw), np.sin(yaw)
        dT = np.array([[c, -s, 0, self.step + rng.normal(0, 0.01)], [s, c, 0, rng.normal(0, 0.01)], [0, 0, 1, rng.normal(0, 0.003)]])
        new = np.hstack([self.Twr[:, :3] @ dT[:, :3], (self.Twr[:, :3] @ dT[:, 3] + self.Twr[:, 3])[:, None]])
        translation = new[:, 3] - self.Twr[:, 3]
        self.Twr = new
        wn = dT.copy(); wn[:, 3] += rng.normal(0, 0.005, 3)
        self.wheel = np.hstack([self.wheel[:, :3] @ wn[:, :3], (self.wheel[:, :3] @ wn[:, 3] + self.wheel[:, 3])[:, None]])
        while len(self.tracks) < self.n_tracks:
            self._spawn()
        Twr_noisy = self.Twr.copy(); Twr_noisy[:, 3] += rng.normal(0, 0.02, 3)   # the tracker's pose estimate
        Trw = self._inv(self.Twr); Tcr = self._inv(self.Trc.reshape(3, 4))
        words, right, words3d = {}, {}, {}
        for fid in sorted(self.tracks):
            t = self.tracks[fid]
            pr = Trw[:, :3] @ t["world"] + Trw[:, 3]
            pc = Tcr[:, :3] @ pr + Tcr[:, 3]
            t["remaining"] -= 1
            if pc[2] < 0.5:
                t["remaining"] = 0
                continue
            u = self.fx * pc[0] / pc[2] + self.cx + rng.normal(0, self.pix_noise)
            v = self.fy * pc[1] / pc[2] + self.cy + rng.normal(0, self.pix_noise)
            ur = u - self.fx * float(self.baseline) / pc[2] + rng.normal(0, self.pix_noise)
            words[fid] = (f32(u), f32(v)); right[fid] = (f32(ur), f32(v))
            if not t["no3d"]:
                p3 = pr + rng.normal(0, 0.03, 3)
                if rng.random() < self.p_nan:
                    p3 = np.array([np.nan, np.nan, np.nan])
                words3d[fid] = tuple(f32(x) for x in p3)
        for fid in [k for k, t in self.tracks.items() if t["remaining"] <= 0]:
            del self.tracks[fid]
        wheel = np.zeros(12) if rng.random() < self.p_no_wheel else self.wheel.reshape(12).copy()
        sig = dict(id=self.next_sig, pose=Twr_noisy.reshape(12).copy(), wheel=wheel, translation=translation.copy(),
                   words=words, right=right, words3d=words3d, covisible=dict(self.prev_words))
        self.next_sig += 1
        self.prev_words = dict(words)
        return sig


def insert_native(wm, s):
    ids = sorted(s["words"])
    uv = np.array([[*s["words"][i], *s["right"][i]] for i in ids], f32).reshape(-1, 4)
    xyz = np.array([s["words3d"].get(i, (0, 0, 0)) for i in ids], f32).reshape(-1, 3)
    has = np.array([i in s["words3d"] for i in ids], np.uint8)
    cids = sorted(s["covisible"])
    cuv = np.array([s["covisible"][i] for i in cids], f32).reshape(-1, 2)
    return wm.insert(s["id"], s["pose"], s["wheel"], s["translation"], ids, uv, xyz, has, cids, cuv)


def insert_oracle(lm, s):
    return lm.insert(s["id"], list(s["pose"]), list(s["wheel"]), list(s["translation"]), s["words"], s["right"], s["words3d"], s["covisible"])
