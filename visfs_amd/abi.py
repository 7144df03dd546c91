"""ctypes mirror of include/visfs_ba.h (the C ABI of the BA backend).

Only struct layouts and helpers to fill them from numpy arrays live here; no
arithmetic.  The product binding (visfs_amd.backend) and the test-side checker binding
under tests/ both use these definitions so that they are fed identical bytes.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 8
TUNE_LATENCY, TUNE_THROUGHPUT = 0, 1            # visfs_ba_set_tuning
MAX_TRACE = 64

# status codes (include/visfs_ba.h)
OK, PASSTHROUGH, ERR_TOO_FEW_POSES, ERR_NAN_CHI2, ERR_HUGE_CHI2_1, ERR_HUGE_CHI2_2, \
    ERR_BAD_ARGUMENT, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOT_LOADED = range(10)

# stage buffer ids
BUF_OBS_ERR, BUF_OBS_CHI2, BUF_OBS_WEIGHT, BUF_HPL, BUF_HLL, BUF_BL, BUF_HPP, BUF_BP, \
    BUF_S, BUF_BS, BUF_DX_POSE, BUF_DX_POINT, BUF_POSE_TRIAL, BUF_POINT_TRIAL = range(14)

_pd = C.POINTER(C.c_double)
_pf = C.POINTER(C.c_float)
_pu8 = C.POINTER(C.c_uint8)
_pi32 = C.POINTER(C.c_int32)
_pu64 = C.POINTER(C.c_uint64)


class Params(C.Structure):
    _fields_ = [("framework", C.c_int32), ("solver", C.c_int32), ("trust_region", C.c_int32),
                ("iterations", C.c_int32), ("pixel_variance", C.c_double),
                ("odometry_covariance", C.c_double), ("laser_covariance", C.c_double),
                ("robust_kernel_delta", C.c_double)]


def default_params(**kw):
    """Reference defaults, Parameters.h:184-191."""
    p = Params(0, 0, 0, 10, 1.5, 0.00005, 0.1, 8.0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Grid(C.Structure):
    _fields_ = [("resolution", C.c_double), ("max_x", C.c_double), ("max_y", C.c_double),
                ("num_x_cells", C.c_int32), ("num_y_cells", C.c_int32), ("correspondence_cost", _pf)]


class GridBuffers:
    """Owns the float cost array behind a `Grid` struct.  g: dict(resolution, max_x, max_y, cost[num_y][num_x])."""

    def __init__(self, g):
        self.cost = _arr(g["cost"], np.float32)
        assert self.cost.ndim == 2
        s = Grid()
        s.resolution, s.max_x, s.max_y = float(g["resolution"]), float(g["max_x"]), float(g["max_y"])
        s.num_y_cells, s.num_x_cells = self.cost.shape
        s.correspondence_cost = _ptr(self.cost, C.c_float)
        self.struct = s


class Window(C.Structure):
    _fields_ = [("root_id", C.c_uint64),
                ("n_poses", C.c_int32), ("pose_ids", _pu64), ("pose_Twr", _pd),
                ("n_links", C.c_int32), ("link_from", _pu64), ("link_to", _pu64), ("link_T", _pd),
                ("n_cameras", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("baseline", C.c_float), ("Trc", C.c_double * 12),
                ("n_points", C.c_int32), ("point_ids", _pu64), ("point_xyz", _pd), ("point_fixed", _pu8),
                ("n_refs", C.c_int32), ("ref_feature", _pu64), ("ref_pose", _pu64),
                ("ref_u", _pf), ("ref_v", _pf), ("ref_depth", _pf),
                ("n_laser_points", C.c_int32), ("laser_xyz", _pd), ("grid", C.POINTER(Grid))]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_poses_out", C.c_int32), ("pose_ids_out", _pu64),
                ("pose_Twr_out", _pd), ("outlier_capacity", C.c_int32), ("n_outliers", C.c_int32),
                ("outlier_feature", _pu64), ("outlier_pose", _pu64),
                ("iterations_run", C.c_int32 * 2), ("chi2_initial", C.c_double),
                ("chi2_phase1", C.c_double), ("chi2_final", C.c_double),
                ("warn_mono_skipped", C.c_int32), ("solver_fallback", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("n_odo", C.c_int32),
                ("pose_tq", _pd), ("pose_fixed", _pu8), ("point_xyz", _pd), ("point_fixed", _pu8),
                ("obs_point", _pi32), ("obs_pose", _pi32), ("obs_uvr", _pd),
                ("odo_from", _pi32), ("odo_to", _pi32), ("odo_tq", _pd),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("n_laser", C.c_int32), ("laser_pose", C.c_int32), ("laser_xyz", _pd), ("grid", C.POINTER(Grid)),
                ("Tcr", C.c_double * 12)]


K_NAMES = ["k_linearize", "k_lin_finalize", "k_schur_partial", "k_schur_finalize", "k_pcg", "k_direct", "k_backsub",
           "k_decide", "k_phase_end", "k_reset", "k_small_optimize"]
K_COUNT = len(K_NAMES)


class GraphInfo(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_free_poses", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32),
                ("n_odo", C.c_int32), ("n_blk", C.c_int32), ("n_pairs", C.c_int64), ("lanes_per_landmark", C.c_int32),
                ("n_schur_chunks", C.c_int32), ("device_bytes", C.c_int64), ("fused_path", C.c_int32), ("solver_kernel", C.c_int32),
                ("band_blocks", C.c_int32), ("graph_replayed", C.c_int32), ("unit_form", C.c_int32),
                ("schur_runs", C.c_int32), ("schur_run_landmarks", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("total_ms", C.c_double * K_COUNT), ("launches", C.c_int64 * K_COUNT),
                ("active_ms", C.c_double * K_COUNT), ("active_launches", C.c_int64 * K_COUNT), ("null_pair_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations_run", C.c_int32 * 2), ("trials_run", C.c_int32 * 2),
                ("pcg_iterations", C.c_int32), ("n_outliers", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_phase1", C.c_double), ("chi2_final", C.c_double),
                ("n_trace", C.c_int32), ("trace_lambda", C.c_double * MAX_TRACE),
                ("trace_chi2", C.c_double * MAX_TRACE),
                ("n_active_edges", C.c_int32 * 2), ("pcg_iterations_phase", C.c_int32 * 2),
                ("solver_fallback", C.c_int32), ("reserved", C.c_int32)]


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _arr(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)


class WindowBuffers:
    """Owns the numpy arrays behind a `Window` struct (keeps them alive)."""

    def __init__(self, w):
        """w: dict as produced by visfs_amd.synth.make_window (robot-frame inputs of localOptimize)."""
        self.pose_ids = _arr(w["pose_ids"], np.uint64)
        self.pose_Twr = _arr(w["pose_Twr"], np.float64).reshape(-1, 12)
        self.link_from = _arr(w.get("link_from", []), np.uint64)
        self.link_to = _arr(w.get("link_to", []), np.uint64)
        self.link_T = _arr(w.get("link_T", np.zeros((0, 12))), np.float64).reshape(-1, 12)
        self.point_ids = _arr(w["point_ids"], np.uint64)
        self.point_xyz = _arr(w["point_xyz"], np.float64).reshape(-1, 3).copy()   # in/out
        self.point_fixed = _arr(w["point_fixed"], np.uint8)
        self.ref_feature = _arr(w["ref_feature"], np.uint64)
        self.ref_pose = _arr(w["ref_pose"], np.uint64)
        self.ref_u = _arr(w["ref_u"], np.float32)
        self.ref_v = _arr(w["ref_v"], np.float32)
        self.ref_depth = _arr(w["ref_depth"], np.float32)
        s = Window()
        s.root_id = int(w["root_id"])
        s.n_poses = len(self.pose_ids); s.pose_ids = _ptr(self.pose_ids, C.c_uint64); s.pose_Twr = _ptr(self.pose_Twr, C.c_double)
        s.n_links = len(self.link_from); s.link_from = _ptr(self.link_from, C.c_uint64)
        s.link_to = _ptr(self.link_to, C.c_uint64); s.link_T = _ptr(self.link_T, C.c_double)
        s.n_cameras = int(w.get("n_cameras", 2))
        s.fx, s.fy, s.cx, s.cy = (float(w[k]) for k in ("fx", "fy", "cx", "cy"))
        s.baseline = float(w["baseline"])
        trc = _arr(w["Trc"], np.float64).reshape(12)
        for i in range(12):
            s.Trc[i] = trc[i]
        s.n_points = len(self.point_ids); s.point_ids = _ptr(self.point_ids, C.c_uint64)
        s.point_xyz = _ptr(self.point_xyz, C.c_double); s.point_fixed = _ptr(self.point_fixed, C.c_uint8)
        s.n_refs = len(self.ref_feature); s.ref_feature = _ptr(self.ref_feature, C.c_uint64)
        s.ref_pose = _ptr(self.ref_pose, C.c_uint64); s.ref_u = _ptr(self.ref_u, C.c_float)
        s.ref_v = _ptr(self.ref_v, C.c_float); s.ref_depth = _ptr(self.ref_depth, C.c_float)
        self.laser_xyz = _arr(w.get("laser_xyz", np.zeros((0, 3))), np.float64).reshape(-1, 3)
        s.n_laser_points = int(w.get("n_laser_points", len(self.laser_xyz)))
        if len(self.laser_xyz):
            s.laser_xyz = _ptr(self.laser_xyz, C.c_double)
        self.grid = GridBuffers(w["grid"]) if w.get("grid") is not None else None
        if self.grid is not None:
            s.grid = C.pointer(self.grid.struct)
        self.struct = s


class ResultBuffers:
    def __init__(self, n_poses, n_refs):
        self.pose_ids_out = np.zeros(max(n_poses, 1), np.uint64)
        self.pose_Twr_out = np.zeros((max(n_poses, 1), 12), np.float64)
        self.outlier_feature = np.zeros(max(n_refs, 1), np.uint64)
        self.outlier_pose = np.zeros(max(n_refs, 1), np.uint64)
        r = Result()
        r.pose_ids_out = _ptr(self.pose_ids_out, C.c_uint64)
        r.pose_Twr_out = _ptr(self.pose_Twr_out, C.c_double)
        r.outlier_capacity = max(n_refs, 1)
        r.outlier_feature = _ptr(self.outlier_feature, C.c_uint64)
        r.outlier_pose = _ptr(self.outlier_pose, C.c_uint64)
        self.struct = r

    def poses(self):
        n = self.struct.n_poses_out
        return {int(self.pose_ids_out[i]): self.pose_Twr_out[i].reshape(3, 4).copy() for i in range(n)}

    def outliers(self):
        n = self.struct.n_outliers
        return [(int(self.outlier_feature[i]), int(self.outlier_pose[i])) for i in range(n)]


class GraphBuffers:
    """Flat factor graph (camera-frame) as numpy arrays + the `Graph` struct over them."""

    def __init__(self, pose_tq, pose_fixed, point_xyz, point_fixed, obs_point, obs_pose, obs_uvr,
                 odo_from, odo_to, odo_tq, fx, fy, cx, cy, bf, laser_xyz=None, laser_pose=0, grid=None, Tcr=None):
        self.pose_tq = _arr(pose_tq, np.float64).reshape(-1, 7)
        self.pose_fixed = _arr(pose_fixed, np.uint8)
        self.point_xyz = _arr(point_xyz, np.float64).reshape(-1, 3)
        self.point_fixed = _arr(point_fixed, np.uint8)
        self.obs_point = _arr(obs_point, np.int32)
        self.obs_pose = _arr(obs_pose, np.int32)
        self.obs_uvr = _arr(obs_uvr, np.float64).reshape(-1, 3)
        self.odo_from = _arr(odo_from, np.int32)
        self.odo_to = _arr(odo_to, np.int32)
        self.odo_tq = _arr(odo_tq, np.float64).reshape(-1, 7)
        g = Graph()
        g.n_poses = len(self.pose_tq); g.n_points = len(self.point_xyz)
        g.n_obs = len(self.obs_point); g.n_odo = len(self.odo_from)
        g.pose_tq = _ptr(self.pose_tq, C.c_double); g.pose_fixed = _ptr(self.pose_fixed, C.c_uint8)
        g.point_xyz = _ptr(self.point_xyz, C.c_double); g.point_fixed = _ptr(self.point_fixed, C.c_uint8)
        g.obs_point = _ptr(self.obs_point, C.c_int32); g.obs_pose = _ptr(self.obs_pose, C.c_int32)
        g.obs_uvr = _ptr(self.obs_uvr, C.c_double)
        g.odo_from = _ptr(self.odo_from, C.c_int32); g.odo_to = _ptr(self.odo_to, C.c_int32)
        g.odo_tq = _ptr(self.odo_tq, C.c_double)
        g.fx, g.fy, g.cx, g.cy, g.bf = float(fx), float(fy), float(cx), float(cy), float(bf)
        # laser occupied-space edges: grid is a GridBuffers (kept alive here) or None
        self.laser_xyz = _arr(laser_xyz if laser_xyz is not None else np.zeros((0, 3)), np.float64).reshape(-1, 3)
        self.grid = grid
        if grid is not None and len(self.laser_xyz):
            g.n_laser = len(self.laser_xyz); g.laser_pose = int(laser_pose)
            g.laser_xyz = _ptr(self.laser_xyz, C.c_double); g.grid = C.pointer(grid.struct)
        tcr = _arr(Tcr if Tcr is not None else [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64).reshape(12)
        for i in range(12):
            g.Tcr[i] = tcr[i]
        self.struct = g

    @property
    def n_poses(self):
        return len(self.pose_tq)

    @property
    def n_points(self):
        return len(self.point_xyz)

    @property
    def n_obs(self):
        return len(self.obs_point)


def pack_window_with(lib_pack, params, wb):
    """Run a `*_pack_window` entry point and return (GraphBuffers, point_used, obs_ref, n_mono)."""
    w = wb.struct
    Np, Nl, Nr, Nk = w.n_poses, w.n_points, w.n_refs, w.n_links
    pose_tq = np.zeros((Np, 7)); pose_fixed = np.zeros(Np, np.uint8); used = np.zeros(max(Nl, 1), np.uint8)
    op = np.zeros(max(Nr, 1), np.int32); oc = np.zeros(max(Nr, 1), np.int32); oref = np.zeros(max(Nr, 1), np.int32)
    uvr = np.zeros((max(Nr, 1), 3))
    of = np.zeros(max(Nk, 1), np.int32); ot = np.zeros(max(Nk, 1), np.int32); otq = np.zeros((max(Nk, 1), 7))
    g = Graph(); mono = C.c_int32(0)
    rc = lib_pack(C.byref(params), C.byref(w), _ptr(pose_tq, C.c_double), _ptr(pose_fixed, C.c_uint8),
                  _ptr(used, C.c_uint8), _ptr(op, C.c_int32), _ptr(oc, C.c_int32), _ptr(uvr, C.c_double),
                  _ptr(oref, C.c_int32), _ptr(of, C.c_int32), _ptr(ot, C.c_int32), _ptr(otq, C.c_double),
                  C.byref(g), C.byref(mono))
    if rc != OK:
        raise RuntimeError(f"pack_window failed: status {rc}")
    no, ne = g.n_obs, g.n_odo
    gb = GraphBuffers(pose_tq, pose_fixed, wb.point_xyz, wb.point_fixed, op[:no], oc[:no], uvr[:no],
                      of[:ne], ot[:ne], otq[:ne], g.fx, g.fy, g.cx, g.cy, g.bf,
                      laser_xyz=wb.laser_xyz if g.n_laser else None, laser_pose=g.laser_pose, grid=wb.grid if g.n_laser else None,
                      Tcr=list(g.Tcr))
    return gb, used[:Nl].copy(), oref[:no].copy(), mono.value


PACK_ARGTYPES = [C.POINTER(Params), C.POINTER(Window), _pd, _pu8, _pu8, _pi32, _pi32, _pd, _pi32,
                 _pi32, _pi32, _pd, C.POINTER(Graph), _pi32]
