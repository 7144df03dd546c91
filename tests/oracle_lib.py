"""ctypes binding of the CPU oracle (oracle/libvisfs_ba_oracle*.so).

TEST INFRASTRUCTURE ONLY.  The oracle is the checker for the HIP path; nothing
under visfs_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from visfs_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_pd = C.POINTER(C.c_double)


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def cpu_has_avx512():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return all(x in fl for x in ("avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"))
    except OSError:
        pass
    return False


def load(omp=False, v4=False):
    """v4: the AVX-512 / FMA-contracting build — bench.py's cpu_baseline leg only, never the checker."""
    name = "libvisfs_ba_oracle" + ("_omp" if omp else "") + ("_v4" if v4 else "") + ".so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    lib.oracle_omp_pin.argtypes = [C.POINTER(C.c_int32), C.c_int]
    lib.oracle_omp_pin.restype = C.c_int
    lib.oracle_omp_unpin.argtypes = []
    lib.oracle_omp_unpin.restype = None
    lib.oracle_pose_from_Rt.argtypes = [_pd, _pd, _pd]
    lib.oracle_pose_to_Rt.argtypes = [_pd, _pd, _pd]
    lib.oracle_pose_update.argtypes = [_pd, _pd]
    lib.oracle_stereo_edge.argtypes = [_pd, _pd, _pd, _pd, _pd, _pd, _pd]
    lib.oracle_odo_edge.argtypes = [_pd, _pd, _pd, _pd, _pd, _pd]
    lib.oracle_huber.argtypes = [C.c_double, C.c_double, _pd]
    lib.oracle_pack_window.argtypes = abi.PACK_ARGTYPES
    lib.oracle_pack_window.restype = C.c_int
    lib.oracle_unpack_pose.argtypes = [_pd, _pd, _pd]
    lib.oracle_sys_create.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.Graph), C.c_int]
    lib.oracle_sys_create.restype = C.c_void_p
    lib.oracle_sys_destroy.argtypes = [C.c_void_p]
    lib.oracle_sys_free_poses.argtypes = [C.c_void_p]
    lib.oracle_sys_free_poses.restype = C.c_int
    lib.oracle_sys_linearize.argtypes = [C.c_void_p, _pd, _pd]
    lib.oracle_sys_trial.argtypes = [C.c_void_p, C.c_double, _pd, _pd, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.oracle_sys_fetch.argtypes = [C.c_void_p, C.c_int32, _pd, C.c_size_t]
    lib.oracle_sys_fetch.restype = C.c_int
    lib.oracle_sys_optimize.argtypes = [C.c_void_p, C.POINTER(abi.Stats), _pd]
    lib.oracle_sys_optimize.restype = C.c_int
    lib.oracle_sys_download.argtypes = [C.c_void_p, _pd, _pd, C.POINTER(C.c_uint8), _pd]
    lib.oracle_sys_reset.argtypes = [C.c_void_p]
    for name in ("oracle_sys_commit", "oracle_sys_begin_phase", "oracle_sys_mark_outliers"):
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.oracle_sys_mark_outliers.restype = C.c_int
    lib.oracle_lm_script.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _pd, _pd, C.POINTER(C.c_int32), C.POINTER(abi.Stats)]
    lib.oracle_lm_script.restype = C.c_int
    _pi = C.POINTER(C.c_int32)
    lib.oracle_ceres_script.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _pi, _pd, _pd, _pd, _pd, _pd, C.POINTER(abi.Stats)]
    lib.oracle_ceres_script.restype = C.c_int
    lib.oracle_dogleg_script.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _pi, _pd, _pd, _pd, _pd, _pd, _pd, C.POINTER(abi.Stats), _pd]
    lib.oracle_dogleg_script.restype = C.c_int
    lib.oracle_dogleg_combine.argtypes = [C.c_double] * 6 + [_pd]
    lib.oracle_dogleg_combine.restype = None
    lib.oracle_sys_dogleg_trial.argtypes = [C.c_void_p, C.c_double, C.c_double, _pd]
    lib.oracle_sys_dogleg_trial.restype = C.c_int
    lib.oracle_solve_window.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.Window), C.POINTER(abi.Result), C.c_int]
    lib.oracle_solve_window.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(_pd)


class OracleSystem:
    """RAII wrapper over oracle_sys with the same stage surface as visfs_amd.backend.Solver."""

    def __init__(self, lib, params, gb, threads=1):
        self.lib, self.gb, self.params = lib, gb, params
        self.h = lib.oracle_sys_create(C.byref(params), C.byref(gb.struct), threads)
        self.npf = lib.oracle_sys_free_poses(self.h)

    def close(self):
        if self.h:
            self.lib.oracle_sys_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def linearize(self):
        chi, md = C.c_double(), C.c_double()
        self.lib.oracle_sys_linearize(self.h, C.byref(chi), C.byref(md))
        return chi.value, md.value

    def trial(self, lam):
        chi, sc, it, ok = C.c_double(), C.c_double(), C.c_int32(), C.c_int32()
        self.lib.oracle_sys_trial(self.h, lam, C.byref(chi), C.byref(sc), C.byref(it), C.byref(ok))
        return chi.value, sc.value, it.value, ok.value

    def dogleg_trial(self, radius, mu):
        out = np.zeros(3)
        ok = self.lib.oracle_sys_dogleg_trial(self.h, radius, mu, out.ctypes.data_as(C.POINTER(C.c_double)))
        return ok, out[0], out[1], out[2]

    def commit(self):
        self.lib.oracle_sys_commit(self.h)

    def begin_phase(self):
        self.lib.oracle_sys_begin_phase(self.h)

    def mark_outliers(self):
        return self.lib.oracle_sys_mark_outliers(self.h)

    def fetch(self, which):
        n6 = 6 * self.npf
        g = self.gb
        size = {abi.BUF_OBS_ERR: g.n_obs * 3, abi.BUF_OBS_CHI2: g.n_obs, abi.BUF_OBS_WEIGHT: g.n_obs,
                abi.BUF_HPL: g.n_obs * 18, abi.BUF_HLL: g.n_points * 6, abi.BUF_BL: g.n_points * 3,
                abi.BUF_HPP: n6 * n6, abi.BUF_BP: n6, abi.BUF_S: n6 * n6, abi.BUF_BS: n6,
                abi.BUF_DX_POSE: n6, abi.BUF_DX_POINT: g.n_points * 3,
                abi.BUF_POSE_TRIAL: g.n_poses * 7, abi.BUF_POINT_TRIAL: g.n_points * 3}[which]
        out = np.zeros(max(size, 1))
        rc = self.lib.oracle_sys_fetch(self.h, which, _p(out), size)
        assert rc == 0
        return out[:size]

    def optimize(self):
        st = abi.Stats()
        sec = C.c_double()
        rc = self.lib.oracle_sys_optimize(self.h, C.byref(st), C.byref(sec))
        return rc, st, sec.value

    def reset(self):
        self.lib.oracle_sys_reset(self.h)

    def download(self):
        g = self.gb
        pose = np.zeros((g.n_poses, 7)); pt = np.zeros((max(g.n_points, 1), 3))
        out = np.zeros(max(g.n_obs, 1), np.uint8); chi = np.zeros(max(g.n_obs, 1))
        self.lib.oracle_sys_download(self.h, _p(pose), _p(pt), out.ctypes.data_as(C.POINTER(C.c_uint8)), _p(chi))
        return pose, pt[:g.n_points], out[:g.n_obs], chi[:g.n_obs]
