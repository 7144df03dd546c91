"""ctypes binding of the HIP backend (visfs_amd/lib/libvisfs_ba_hip.so) — plumbing only.

The product is the C-ABI library; this module exists so that tests and bench.py can
drive it.  It never falls back to a CPU path: a missing library raises ImportError-like
RuntimeError, a missing GPU surfaces as VISFS_BA_ERR_DEVICE from `visfs_ba_create`.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VISFS_BA_LIB") or os.path.join(_HERE, "lib", "libvisfs_ba_hip.so")     # override: A/B of library builds (tools/)

_pd = C.POINTER(C.c_double)
_pu8 = C.POINTER(C.c_uint8)

EXPORTS = [
    "visfs_ba_abi_version", "visfs_ba_default_params", "visfs_ba_create", "visfs_ba_destroy",
    "visfs_ba_last_error", "visfs_ba_solve_window", "visfs_ba_solve_batch", "visfs_ba_pack_window",
    "visfs_ba_unpack_pose", "visfs_ba_graph_upload", "visfs_ba_graph_reset", "visfs_ba_optimize",
    "visfs_ba_graph_download", "visfs_ba_graph_free_poses", "visfs_ba_stage_linearize",
    "visfs_ba_stage_trial", "visfs_ba_stage_fetch", "visfs_ba_graph_describe", "visfs_ba_profile_enable",
    "visfs_ba_profile_read", "visfs_ba_batch_upload", "visfs_ba_batch_reset", "visfs_ba_batch_optimize", "visfs_ba_batch_download",
    "visfs_ba_create_error", "visfs_ba_set_tuning", "visfs_ba_hook_lm_script", "visfs_ba_hook_ceres_script", "visfs_ba_hook_dogleg_script", "visfs_ba_hook_dogleg_combine", "visfs_ba_solve_batch_sharded", "visfs_ba_stage_commit", "visfs_ba_stage_begin_phase", "visfs_ba_stage_mark_outliers",
]

_lib = None


class BackendError(RuntimeError):
    pass


def load_library():
    """Load the HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("VISFS_BA_LIB", LIB_PATH)        # A/B measurement builds (tools/build_variant.sh); the default is the product library
    if not os.path.exists(path):
        raise BackendError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    lib.visfs_ba_abi_version.restype = C.c_int
    lib.visfs_ba_default_params.argtypes = [C.POINTER(abi.Params)]
    lib.visfs_ba_create.argtypes = [C.POINTER(abi.Params), C.c_int, C.POINTER(C.c_void_p)]
    lib.visfs_ba_create.restype = C.c_int
    lib.visfs_ba_destroy.argtypes = [C.c_void_p]
    lib.visfs_ba_last_error.argtypes = [C.c_void_p]
    lib.visfs_ba_last_error.restype = C.c_char_p
    lib.visfs_ba_create_error.restype = C.c_char_p
    lib.visfs_ba_set_tuning.argtypes = [C.c_void_p, C.c_int32]
    lib.visfs_ba_set_tuning.restype = C.c_int
    lib.visfs_ba_solve_window.argtypes = [C.c_void_p, C.POINTER(abi.Window), C.POINTER(abi.Result)]
    lib.visfs_ba_solve_window.restype = C.c_int
    lib.visfs_ba_solve_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(abi.Window)), C.POINTER(C.POINTER(abi.Result))]
    lib.visfs_ba_solve_batch.restype = C.c_int
    lib.visfs_ba_solve_batch_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.POINTER(C.POINTER(abi.Window)), C.POINTER(C.POINTER(abi.Result))]
    lib.visfs_ba_solve_batch_sharded.restype = C.c_int
    lib.visfs_ba_pack_window.argtypes = abi.PACK_ARGTYPES
    lib.visfs_ba_pack_window.restype = C.c_int
    lib.visfs_ba_unpack_pose.argtypes = [_pd, _pd, _pd]
    lib.visfs_ba_graph_upload.argtypes = [C.c_void_p, C.POINTER(abi.Graph)]
    lib.visfs_ba_graph_upload.restype = C.c_int
    lib.visfs_ba_graph_reset.argtypes = [C.c_void_p]
    lib.visfs_ba_graph_reset.restype = C.c_int
    lib.visfs_ba_optimize.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
    lib.visfs_ba_optimize.restype = C.c_int
    lib.visfs_ba_graph_download.argtypes = [C.c_void_p, _pd, _pd, _pu8, _pd]
    lib.visfs_ba_graph_download.restype = C.c_int
    lib.visfs_ba_graph_free_poses.argtypes = [C.c_void_p]
    lib.visfs_ba_graph_free_poses.restype = C.c_int
    lib.visfs_ba_stage_linearize.argtypes = [C.c_void_p, _pd, _pd]
    lib.visfs_ba_stage_linearize.restype = C.c_int
    lib.visfs_ba_stage_trial.argtypes = [C.c_void_p, C.c_double, _pd, _pd, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.visfs_ba_stage_trial.restype = C.c_int
    lib.visfs_ba_stage_fetch.argtypes = [C.c_void_p, C.c_int32, _pd, C.c_size_t]
    lib.visfs_ba_stage_fetch.restype = C.c_int
    lib.visfs_ba_graph_describe.argtypes = [C.c_void_p, C.POINTER(abi.GraphInfo)]
    lib.visfs_ba_graph_describe.restype = C.c_int
    lib.visfs_ba_batch_upload.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(abi.Graph))]
    lib.visfs_ba_batch_reset.argtypes = [C.c_void_p]
    lib.visfs_ba_batch_optimize.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
    lib.visfs_ba_batch_download.argtypes = [C.c_void_p, C.c_int32, _pd, _pd, _pu8, _pd]
    lib.visfs_ba_profile_enable.argtypes = [C.c_void_p, C.c_uint32]
    lib.visfs_ba_profile_enable.restype = C.c_int
    lib.visfs_ba_profile_read.argtypes = [C.c_void_p, C.POINTER(abi.Profile)]
    lib.visfs_ba_profile_read.restype = C.c_int
    for name in ("visfs_ba_stage_commit", "visfs_ba_stage_begin_phase", "visfs_ba_stage_mark_outliers"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.visfs_ba_hook_lm_script.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, _pd, _pd, C.POINTER(C.c_int32), C.POINTER(abi.Stats)]
    lib.visfs_ba_hook_lm_script.restype = C.c_int
    _pi = C.POINTER(C.c_int32)
    lib.visfs_ba_hook_ceres_script.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, _pi, _pd, _pd, _pd, _pd, _pd, C.POINTER(abi.Stats)]
    lib.visfs_ba_hook_dogleg_script.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, _pi, _pd, _pd, _pd, _pd, _pd, _pd, C.POINTER(abi.Stats), _pd]
    lib.visfs_ba_hook_dogleg_script.restype = C.c_int
    lib.visfs_ba_hook_dogleg_combine.argtypes = [C.c_double] * 6 + [_pd]
    lib.visfs_ba_hook_dogleg_combine.restype = C.c_int
    lib.visfs_ba_hook_ceres_script.restype = C.c_int
    if lib.visfs_ba_abi_version() != abi.ABI_VERSION:
        raise BackendError("ABI version mismatch between visfs_amd/abi.py and libvisfs_ba_hip.so")
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(_pd)


class Solver:
    """One `visfs_ba_handle` (one GPU, one stream) — the analogue of a VISFS::Optimizer::Optimizer instance."""

    def __init__(self, params=None, device=0, tuning=None):
        self.lib = load_library()
        self.params = params if params is not None else abi.default_params()
        h = C.c_void_p()
        rc = self.lib.visfs_ba_create(C.byref(self.params), device, C.byref(h))
        if rc != abi.OK:
            raise BackendError(f"visfs_ba_create failed with status {rc}: {self.lib.visfs_ba_create_error().decode()}")
        self.h = h
        self.gb = None
        if tuning is not None:
            self.set_tuning(tuning)

    def set_tuning(self, tuning):
        """abi.TUNE_LATENCY (default) / abi.TUNE_THROUGHPUT (a handle that is mostly given batches): applies to later uploads."""
        self._check(self.lib.visfs_ba_set_tuning(self.h, int(tuning)), "set_tuning")

    def close(self):
        if getattr(self, "h", None):
            self.lib.visfs_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what, allow=()):
        if rc != abi.OK and rc not in allow:
            raise BackendError(f"{what}: status {rc}: {self.lib.visfs_ba_last_error(self.h).decode()}")
        return rc

    # ---- graph layer
    def upload(self, gb):
        self.gb = gb
        self._check(self.lib.visfs_ba_graph_upload(self.h, C.byref(gb.struct)), "graph_upload")
        self.npf = self.lib.visfs_ba_graph_free_poses(self.h)

    def reset(self):
        self._check(self.lib.visfs_ba_graph_reset(self.h), "graph_reset")

    def optimize(self):
        st = abi.Stats()
        rc = self.lib.visfs_ba_optimize(self.h, C.byref(st))
        if rc in (abi.ERR_DEVICE, abi.ERR_NOT_LOADED, abi.ERR_BAD_ARGUMENT):
            self._check(rc, "optimize")
        return rc, st

    # ---- a batch of independent windows resident side by side (BASELINE config 5): one sequence of launches serves all
    def batch_upload(self, gbs):
        self.batch = list(gbs)                    # keep the host arrays alive
        arr = (C.POINTER(abi.Graph) * len(gbs))(*[C.pointer(g.struct) for g in gbs])
        self._check(self.lib.visfs_ba_batch_upload(self.h, len(gbs), arr), "batch_upload")

    def batch_reset(self):
        self._check(self.lib.visfs_ba_batch_reset(self.h), "batch_reset")

    def batch_optimize(self):
        stats = (abi.Stats * len(self.batch))()
        rc = self.lib.visfs_ba_batch_optimize(self.h, stats)
        if rc in (abi.ERR_DEVICE, abi.ERR_NOT_LOADED, abi.ERR_UNSUPPORTED, abi.ERR_BAD_ARGUMENT):
            self._check(rc, "batch_optimize")
        return rc, list(stats)

    def batch_download(self, index):
        gb = self.batch[index]
        pose = np.zeros((gb.n_poses, 7)); pt = np.zeros((max(gb.n_points, 1), 3))
        outl = np.zeros(max(gb.n_obs, 1), np.uint8); chi2 = np.zeros(max(gb.n_obs, 1))
        self._check(self.lib.visfs_ba_batch_download(self.h, index, _p(pose), _p(pt), outl.ctypes.data_as(_pu8), _p(chi2)), "batch_download")
        return pose, pt[:gb.n_points], outl[:gb.n_obs], chi2[:gb.n_obs]

    def download(self):
        g = self.gb
        pose = np.zeros((g.n_poses, 7)); pt = np.zeros((max(g.n_points, 1), 3))
        out = np.zeros(max(g.n_obs, 1), np.uint8); chi = np.zeros(max(g.n_obs, 1))
        self._check(self.lib.visfs_ba_graph_download(self.h, _p(pose), _p(pt), out.ctypes.data_as(_pu8), _p(chi)), "graph_download")
        return pose, pt[:g.n_points], out[:g.n_obs], chi[:g.n_obs]

    # ---- measurement hooks
    def describe(self):
        info = abi.GraphInfo()
        self._check(self.lib.visfs_ba_graph_describe(self.h, C.byref(info)), "graph_describe")
        return {k: getattr(info, k) for k, _ in abi.GraphInfo._fields_}

    def profile_enable(self, which=True):
        """which: True (all kernel classes), False (none) or an iterable of kernel names (abi.K_NAMES)."""
        if which is True:
            mask = (1 << abi.K_COUNT) - 1
        elif not which:
            mask = 0
        else:
            mask = sum(1 << abi.K_NAMES.index(k) for k in which)
        self._check(self.lib.visfs_ba_profile_enable(self.h, mask), "profile_enable")

    def profile_read(self):
        p = abi.Profile()
        self._check(self.lib.visfs_ba_profile_read(self.h, C.byref(p)), "profile_read")
        out = {abi.K_NAMES[k]: dict(total_ms=p.total_ms[k], launches=p.launches[k], active_ms=p.active_ms[k],
                                    active_launches=p.active_launches[k]) for k in range(abi.K_COUNT) if p.launches[k]}
        self.null_pair_ms = p.null_pair_ms        # what an empty event pair measures on this stream
        return out

    # ---- stage hooks (parity tests)
    def linearize(self):
        chi, md = C.c_double(), C.c_double()
        self._check(self.lib.visfs_ba_stage_linearize(self.h, C.byref(chi), C.byref(md)), "stage_linearize")
        return chi.value, md.value

    def trial(self, lam):
        chi, sc, it, ok = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        self._check(self.lib.visfs_ba_stage_trial(self.h, lam, C.byref(chi), C.byref(sc), C.byref(it), C.byref(ok)), "stage_trial")
        return chi.value, sc.value, it.value, ok.value

    def commit(self):
        self._check(self.lib.visfs_ba_stage_commit(self.h), "stage_commit")

    def begin_phase(self):
        self._check(self.lib.visfs_ba_stage_begin_phase(self.h), "stage_begin_phase")

    def mark_outliers(self):
        self._check(self.lib.visfs_ba_stage_mark_outliers(self.h), "stage_mark_outliers")

    def fetch(self, which):
        n6 = 6 * self.npf
        g = self.gb
        size = {abi.BUF_OBS_ERR: g.n_obs * 3, abi.BUF_OBS_CHI2: g.n_obs, abi.BUF_OBS_WEIGHT: g.n_obs,
                abi.BUF_HPL: g.n_obs * 18, abi.BUF_HLL: g.n_points * 6, abi.BUF_BL: g.n_points * 3,
                abi.BUF_HPP: n6 * n6, abi.BUF_BP: n6, abi.BUF_S: n6 * n6, abi.BUF_BS: n6,
                abi.BUF_DX_POSE: n6, abi.BUF_DX_POINT: g.n_points * 3,
                abi.BUF_POSE_TRIAL: g.n_poses * 7, abi.BUF_POINT_TRIAL: g.n_points * 3}[which]
        out = np.zeros(max(size, 1))
        self._check(self.lib.visfs_ba_stage_fetch(self.h, which, _p(out), size), "stage_fetch")
        return out[:size]

    # ---- window layer
    def solve_window(self, wb, rb=None):
        rb = rb if rb is not None else abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs)
        rc = self.lib.visfs_ba_solve_window(self.h, C.byref(wb.struct), C.byref(rb.struct))
        if rc in (abi.ERR_DEVICE, abi.ERR_NOT_LOADED):
            self._check(rc, "solve_window")
        return rc, rb

    @staticmethod
    def solve_batch_sharded(solvers, wbs):
        """visfs_ba_solve_batch_sharded: the windows in contiguous blocks over the solvers' handles (one per GPU), one host thread each."""
        n = len(wbs)
        rbs = [abi.ResultBuffers(w.struct.n_poses, w.struct.n_refs) for w in wbs]
        wa = (C.POINTER(abi.Window) * n)(*[C.pointer(w.struct) for w in wbs])
        ra = (C.POINTER(abi.Result) * n)(*[C.pointer(r.struct) for r in rbs])
        hs = (C.c_void_p * len(solvers))(*[s.h for s in solvers])
        rc = solvers[0].lib.visfs_ba_solve_batch_sharded(hs, len(solvers), n, wa, ra)
        return rc, rbs

    def solve_batch(self, wbs):
        n = len(wbs)
        rbs = [abi.ResultBuffers(w.struct.n_poses, w.struct.n_refs) for w in wbs]
        WP = C.POINTER(abi.Window) * n
        RP = C.POINTER(abi.Result) * n
        wa = WP(*[C.pointer(w.struct) for w in wbs])
        ra = RP(*[C.pointer(r.struct) for r in rbs])
        rc = self.lib.visfs_ba_solve_batch(self.h, n, wa, ra)
        self._check(rc, "solve_batch")
        return rbs
