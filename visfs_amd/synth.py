"""Deterministic synthetic sliding windows for the BA hot path.

Produces the ROBOT-FRAME inputs of `Optimizer::localOptimize`
(reference: corelib/include/Optimizer/Optimizer.h:46-56) the way LocalMap would
(corelib/src/LocalMap.cpp:228-294): poses Twr keyed by signature id, wheel-odometry
links between consecutive signatures, world landmarks with a `fixed` flag, and per
(feature, pose) float key-points + float depth.  No arithmetic of the solver lives
here; this is input synthesis only.

RNG: SplitMix64 counter stream → uniform → Box-Muller, seed = 20261003 + config
index (+ window index), so every run (here, on the GPU box, any N) sees the same bits.
"""
import numpy as np

# BASELINE.json configs: (n_keyframes, n_landmarks, n_observations, with_odometry)
CONFIGS = {
    "C1": dict(n_kf=10, n_lm=500, n_obs=3000, odo=False, index=0),
    "C2": dict(n_kf=50, n_lm=5000, n_obs=50000, odo=False, index=1),
    "C3": dict(n_kf=50, n_lm=5000, n_obs=50000, odo=True, index=2),   # C2 + 49 wheel-odometry edges (no IMU factor exists in the reference)
    "C4": dict(n_kf=200, n_lm=30000, n_obs=300000, odo=False, index=3),
    # C4 with the world origin moved to the centroid of the trajectory (same measurements, same noise): the reference's pose
    # Jacobian is the SE(3)-left form in Pc while its update leaves t unrotated — an approximation whose error grows
    # with |t_cw|, i.e. with the distance of a key-frame from the world origin; 50 m of trajectory from the origin do not converge
    "C4C": dict(n_kf=200, n_lm=30000, n_obs=300000, odo=False, index=3, centre=True),
    # C4 started from 1e-3 rad (0.06 deg) of rotation error per pose instead of 1e-2: the update error of the a6 quirk is
    # |d_theta x t_cw|, so at 25-50 m from the origin only small rotation corrections keep the LM steps acceptable — phase 1
    # then converges in 10 accepted trials and the kernels see all 300 k edges (C4 itself: 69 % of the edges are culled)
    "C4R": dict(n_kf=200, n_lm=30000, n_obs=300000, odo=False, index=3, pose_noise_r=1e-3),
    "C5": dict(n_kf=50, n_lm=5000, n_obs=50000, odo=False, index=4),  # per window; 64 windows in the batch
    # production-sized window (Parameters.h:161,148: 6 signatures, <=300 features)
    "PROD": dict(n_kf=6, n_lm=300, n_obs=1500, odo=True, index=5),
    # wide band (measurement only): tracks of 30 key-frames give the reduced camera matrix a block half-bandwidth of 29 — beyond what the
    # banded factorisation holds in LDS, so Optimizer/Solver=0 takes the dense blocked Cholesky, the one place with an MFMA (fp64 SYRK)
    "WB": dict(n_kf=60, n_lm=1200, n_obs=36000, odo=False, index=6),
}
BASE_SEED = 20261003

# camera: 752x480, fx=fy=435.2, cx=367.2, cy=252.2, baseline 0.11f
WIDTH, HEIGHT = 752, 480
FX = FY = 435.2
CX, CY = 367.2, 252.2
BASELINE_F = np.float32(0.11)
# GeometricCamera.h:15-19: image→robot rotation, zero translation
TRC = np.array([[0.0, 0.0, 1.0, 0.0], [-1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 0.0, 0.0]])

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


class SplitMix64:
    """Counter-based SplitMix64: output i = mix(seed + (i+1)*GOLDEN)."""

    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self.count = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(self.count + 1, self.count + n + 1, dtype=np.uint64)
            z = self.seed + idx * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self.count += n
        return z

    def uniform(self, n):
        """(0,1) doubles with 53 random bits."""
        return ((self.u64(n) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)

    def normal(self, n):
        u1 = self.uniform(n)
        u2 = self.uniform(n)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _rotz(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.zeros(yaw.shape + (3, 3))
    R[..., 0, 0] = c; R[..., 0, 1] = -s; R[..., 1, 0] = s; R[..., 1, 1] = c; R[..., 2, 2] = 1.0
    return R


def _exp_so3(w):
    """Rodrigues, vectorised over leading dims. w: (...,3)"""
    th = np.linalg.norm(w, axis=-1)
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1] = -w[..., 2]; K[..., 0, 2] = w[..., 1]
    K[..., 1, 0] = w[..., 2]; K[..., 1, 2] = -w[..., 0]
    K[..., 2, 0] = -w[..., 1]; K[..., 2, 1] = w[..., 0]
    th2 = th * th
    small = th < 1e-8
    a = np.where(small, 1.0 - th2 / 6.0, np.sin(th) / np.where(small, 1.0, th))
    b = np.where(small, 0.5 - th2 / 24.0, (1.0 - np.cos(th)) / np.where(small, 1.0, th2))
    I = np.broadcast_to(np.eye(3), K.shape)
    return I + a[..., None, None] * K + b[..., None, None] * (K @ K)


def _iso(R, t):
    T = np.zeros(R.shape[:-2] + (3, 4))
    T[..., :3, :3] = R
    T[..., :3, 3] = t
    return T


def iso_mul(A, B):
    R = A[..., :3, :3] @ B[..., :3, :3]
    t = (A[..., :3, :3] @ B[..., :3, 3:4])[..., 0] + A[..., :3, 3]
    return _iso(R, t)


def iso_inv(A):
    Rt = np.swapaxes(A[..., :3, :3], -1, -2)
    t = -(Rt @ A[..., :3, 3:4])[..., 0]
    return _iso(Rt, t)


def make_window(config="C2", window_index=0, n_kf=None, n_lm=None, n_obs=None, odo=None,
                noise_px=0.5, outlier_frac=0.02, fixed_frac=0.2, pose_noise_t=0.05, pose_noise_r=0.01,
                point_noise=0.05, seed=None):
    """Build one synthetic window. Returns a dict of numpy arrays (see WindowBuffers) plus ground truth."""
    cfg = dict(CONFIGS[config]) if config in CONFIGS else dict(n_kf=10, n_lm=100, n_obs=600, odo=False, index=9)
    if n_kf is not None: cfg["n_kf"] = n_kf
    if n_lm is not None: cfg["n_lm"] = n_lm
    if n_obs is not None: cfg["n_obs"] = n_obs
    if odo is not None: cfg["odo"] = odo
    Np, Nl, No = cfg["n_kf"], cfg["n_lm"], cfg["n_obs"]
    pose_noise_r = cfg.get("pose_noise_r", pose_noise_r)
    rng = SplitMix64((BASE_SEED + cfg["index"] + 1000003 * window_index) if seed is None else seed)

    # --- trajectory (planar robot): x = 0.25k, y = 0.5 sin(0.2k), yaw = 0.1 cos(0.2k)
    k = np.arange(Np, dtype=np.float64)
    Twr_true = _iso(_rotz(0.1 * np.cos(0.2 * k)), np.stack([0.25 * k, 0.5 * np.sin(0.2 * k), np.zeros(Np)], -1))
    Twc_true = iso_mul(Twr_true, TRC[None])
    Tcw_true = iso_inv(Twc_true)
    bf = float(BASELINE_F) * FX

    # --- track layout: contiguous runs of key-frames, total length exactly No
    base, rem = divmod(No, Nl)
    length = np.full(Nl, base, dtype=np.int64)
    length[:rem] += 1
    if length.max() > Np:
        raise ValueError("n_obs / n_lm exceeds the number of key-frames")
    if length.min() < 1:
        raise ValueError("need at least one observation per landmark")
    start = (np.arange(Nl, dtype=np.int64) * (Np - length + 1)) // Nl
    mid = start + length // 2

    # --- landmarks: sample in the frustum of the middle key-frame, keep if visible in the whole run
    P = np.zeros((Nl, 3))
    todo = np.arange(Nl)
    margin = 8.0
    for _ in range(200):
        if todo.size == 0:
            break
        n = todo.size
        u = margin + rng.uniform(n) * (WIDTH - 2 * margin)
        v = margin + rng.uniform(n) * (HEIGHT - 2 * margin)
        z = 2.0 + rng.uniform(n) * 8.0                       # depth 2–10 m (Parameters.h:152-153)
        pc = np.stack([(u - CX) / FX * z, (v - CY) / FY * z, z], -1)
        Twc = Twc_true[mid[todo]]
        pw = (Twc[:, :3, :3] @ pc[:, :, None])[:, :, 0] + Twc[:, :3, 3]
        ok = np.ones(n, bool)
        # check both ends of the run and the middle third points (projection is monotone enough along the path)
        for off in (0, 1, 2, 3):
            kf = np.minimum(start[todo] + (length[todo] - 1) * off // 3, Np - 1)
            T = Tcw_true[kf]
            q = (T[:, :3, :3] @ pw[:, :, None])[:, :, 0] + T[:, :3, 3]
            zz = q[:, 2]
            uu = FX * q[:, 0] / np.where(zz > 0.2, zz, 1.0) + CX
            vv = FY * q[:, 1] / np.where(zz > 0.2, zz, 1.0) + CY
            ur = uu - bf / np.where(zz > 0.2, zz, 1.0)
            ok &= (zz > 0.5) & (uu > 2) & (uu < WIDTH - 2) & (vv > 2) & (vv < HEIGHT - 2) & (ur > 0)
        P[todo[ok]] = pw[ok]
        todo = todo[~ok]
    if todo.size:
        raise RuntimeError("landmark sampling did not converge")

    # --- observations, feature-major then pose-major (nested std::map order)
    obs_lm = np.repeat(np.arange(Nl), length)
    first = np.cumsum(length) - length
    obs_kf = start[obs_lm] + (np.arange(No) - first[obs_lm])
    T = Tcw_true[obs_kf]
    q = (T[:, :3, :3] @ P[obs_lm][:, :, None])[:, :, 0] + T[:, :3, 3]
    u = FX * q[:, 0] / q[:, 2] + CX
    v = FY * q[:, 1] / q[:, 2] + CY
    d = bf / q[:, 2]
    u = u + noise_px * rng.normal(No)
    v = v + noise_px * rng.normal(No)
    d = d + noise_px * rng.normal(No)
    gross = rng.uniform(No) < outlier_frac
    gu = 10.0 + 20.0 * rng.uniform(No)
    gv = 10.0 + 20.0 * rng.uniform(No)
    u = np.where(gross, u + gu, u)
    v = np.where(gross, v + gv, v)
    d = np.maximum(d, 0.5)
    ref_u = u.astype(np.float32)
    ref_v = v.astype(np.float32)
    ref_depth = (bf / d).astype(np.float32)                  # FeatureBA::depth is float (Optimizer.h:20-27)

    # --- initial state: perturbed poses (root unperturbed), perturbed landmarks, 20 % fixed & exact
    pose_ids = np.arange(1, Np + 1, dtype=np.uint64)         # ids must be > 0 (Optimizer.cpp:74)
    root_id = int(pose_ids[-1]) - 1                          # Estimator.cpp:252
    dt = pose_noise_t * rng.normal(3 * Np).reshape(Np, 3)
    dr = pose_noise_r * rng.normal(3 * Np).reshape(Np, 3)
    root_idx = Np - 2
    dt[root_idx] = 0.0; dr[root_idx] = 0.0
    Twr0 = _iso(Twr_true[:, :3, :3] @ _exp_so3(dr), Twr_true[:, :3, 3] + dt)
    fixed = rng.uniform(Nl) < fixed_frac
    P0 = P + np.where(fixed[:, None], 0.0, point_noise * rng.normal(3 * Nl).reshape(Nl, 3))

    if cfg.get("centre"):
        # a translation of the world: poses and landmarks move, measurements (and the random streams above) stay
        c = Twr_true[:, :3, 3].mean(axis=0)
        Twr_true = Twr_true.copy(); Twr_true[:, :3, 3] -= c
        Twr0 = Twr0.copy(); Twr0[:, :3, 3] -= c
        P = P - c; P0 = P0 - c
    w = dict(config=config, root_id=root_id, pose_ids=pose_ids, pose_Twr=Twr0.reshape(Np, 12),
             n_cameras=2, fx=FX, fy=FY, cx=CX, cy=CY, baseline=float(BASELINE_F), Trc=TRC.reshape(12),
             point_ids=np.arange(Nl, dtype=np.uint64), point_xyz=P0, point_fixed=fixed.astype(np.uint8),
             ref_feature=obs_lm.astype(np.uint64), ref_pose=pose_ids[obs_kf], ref_u=ref_u, ref_v=ref_v,
             ref_depth=ref_depth, n_laser_points=0, laser_xyz=np.zeros((0, 3)), grid=None,
             truth_Twr=Twr_true.reshape(Np, 12), truth_points=P, gross=gross)
    if cfg["odo"]:
        # LocalMap.cpp:238-272: consecutive pairs, T_r1r2 = from^-1 * to, + first-order noise N(0, 5e-5) per dof
        sig = np.sqrt(5e-5)
        T12 = iso_mul(iso_inv(Twr_true[:-1]), Twr_true[1:])
        nt = sig * rng.normal(3 * (Np - 1)).reshape(Np - 1, 3)
        nr = sig * rng.normal(3 * (Np - 1)).reshape(Np - 1, 3)
        T12 = _iso(T12[:, :3, :3] @ _exp_so3(nr), T12[:, :3, 3] + nt)
        w["link_from"] = pose_ids[:-1].copy()
        w["link_to"] = pose_ids[1:].copy()
        w["link_T"] = T12.reshape(Np - 1, 12)
    else:
        w["link_from"] = np.zeros(0, np.uint64)
        w["link_to"] = np.zeros(0, np.uint64)
        w["link_T"] = np.zeros((0, 12))
    return w


def make_grid(num_x=240, num_y=200, resolution=0.05, max_x=9.0, max_y=5.0, seed=7):
    """A probability grid as the laser factor reads it (Map::Grid2D through GridArrayAdapter): float correspondence costs
    in [0.1, 0.9], low on the walls of a rectangular room with a pillar, rising with the distance to them; a band of
    unknown cells (float(0.9)) on one side.  Cell (x, y) has its centre at (max_x - res (y + .5), max_y - res (x + .5))
    (MapLimits::getCellCenter, MapLimits.h:75-79)."""
    rng = SplitMix64(BASE_SEED + 777 + seed)
    xi, yi = np.meshgrid(np.arange(num_x), np.arange(num_y))          # arrays [num_y][num_x]
    wx = max_x - resolution * (yi + 0.5)
    wy = max_y - resolution * (xi + 0.5)
    x0, x1, y0, y1 = max_x - num_y * resolution + 0.6, max_x - 0.6, max_y - num_x * resolution + 0.8, max_y - 0.8
    d_wall = np.minimum.reduce([np.abs(wx - x0), np.abs(wx - x1), np.abs(wy - y0), np.abs(wy - y1)])
    d_pillar = np.abs(np.hypot(wx - (x0 + 2.0), wy - (y0 + 3.0)) - 0.4)
    d = np.minimum(d_wall, d_pillar)
    cost = np.clip(0.1 + 0.9 * d + 0.01 * rng.normal(num_x * num_y).reshape(num_y, num_x), 0.1, 0.9).astype(np.float32)
    cost[:, :6] = np.float32(0.9)                                      # unknown band
    return dict(resolution=resolution, max_x=max_x, max_y=max_y, cost=cost, room=(x0, x1, y0, y1))


def make_laser_window(n_kf=6, n_points=720, with_visual=False, seed=0, pose_noise_t=0.03, pose_noise_r=0.01):
    """Sensor strategy 4/5 (Estimator.cpp:243-250): poses + wheel-odometry links + one laser scan against the matching
    submap, no landmarks (with_visual=True keeps the stereo part of a small window as well)."""
    w = make_window("PROD", n_kf=n_kf, odo=True, seed=BASE_SEED + 4242 + seed, pose_noise_t=pose_noise_t, pose_noise_r=pose_noise_r)
    if not with_visual:
        for k, dt in (("point_ids", np.uint64), ("point_fixed", np.uint8), ("ref_feature", np.uint64), ("ref_pose", np.uint64),
                      ("ref_u", np.float32), ("ref_v", np.float32), ("ref_depth", np.float32)):
            w[k] = np.zeros(0, dt)
        w["point_xyz"] = np.zeros((0, 3))
    rng = SplitMix64(BASE_SEED + 999 + seed)
    g = make_grid(seed=seed)
    x0, x1, y0, y1 = g["room"]
    # the trajectory of make_window starts at the world origin: shift the room so that it lies inside
    off = np.array([x0 + 1.0, y0 + 2.0, 0.0])
    Ttrue = w["truth_Twr"].reshape(-1, 3, 4).copy(); Ttrue[:, :, 3] += off
    T0 = w["pose_Twr"].reshape(-1, 3, 4).copy(); T0[:, :, 3] += off
    w["truth_Twr"] = Ttrue.reshape(-1, 12); w["pose_Twr"] = T0.reshape(-1, 12)
    if with_visual:
        w["point_xyz"] = w["point_xyz"] + off; w["truth_points"] = w["truth_points"] + off
    # one scan from the newest (true) pose: rays to the room walls, hits expressed in that robot frame
    R, t = Ttrue[-1][:, :3], Ttrue[-1][:, 3]
    ang = 2 * np.pi * (np.arange(n_points) + 0.5) / n_points
    dirs = np.stack([np.cos(ang), np.sin(ang)], -1)
    with np.errstate(divide="ignore"):
        tx = np.where(dirs[:, 0] > 0, (x1 - t[0]) / dirs[:, 0], (x0 - t[0]) / dirs[:, 0])
        ty = np.where(dirs[:, 1] > 0, (y1 - t[1]) / dirs[:, 1], (y0 - t[1]) / dirs[:, 1])
    rng_len = np.minimum(tx, ty) + 0.01 * rng.normal(n_points)
    hits_w = np.stack([t[0] + rng_len * dirs[:, 0], t[1] + rng_len * dirs[:, 1], np.full(n_points, t[2])], -1)
    w["laser_xyz"] = (hits_w - t) @ R                      # R^T (p - t)
    w["n_laser_points"] = n_points
    w["grid"] = g
    return w


def pose_errors(Twr_a, Twr_b):
    """Max over poses of relative translation error and geodesic rotation angle (SURVEY §8d metric)."""
    A = np.asarray(Twr_a).reshape(-1, 3, 4)
    B = np.asarray(Twr_b).reshape(-1, 3, 4)
    dt = np.linalg.norm(A[:, :, 3] - B[:, :, 3], axis=1) / np.maximum(1.0, np.linalg.norm(B[:, :, 3], axis=1))
    Rrel = np.swapaxes(A[:, :, :3], 1, 2) @ B[:, :, :3]
    # geodesic angle from the chord ||R - I||_F = 2 sqrt(2) sin(theta/2): accurate near zero, unlike acos(trace)
    chord = np.linalg.norm(Rrel - np.eye(3), axis=(1, 2))
    ang = 2.0 * np.arcsin(np.clip(chord / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
    return float(dt.max()), float(ang.max())
