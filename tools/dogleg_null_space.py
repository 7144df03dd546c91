"""Diagnostic for the DOGLEG soak (tools/soak_ceres.py ... 1): where do the HIP path and the CPU checker differ on windows whose reduced
system is rank-deficient at the start (a pose that sees too few landmarks: DOGLEG's Gauss-Newton solve is regularised by mu = 1e-8 only)?
Per seed: the smallest eigenvalue of the Jacobi-scaled reduced system at x0, then per pose the observation count and the difference
between the two results after 1 iteration and after the full solve, with the small-solve path on and off.

usage: python tools/dogleg_null_space.py 559 576 ..."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib
import test_gpu_random as T
from helpers import graph_of
from test_ceres_flavour import _dense_system
from visfs_amd import abi, backend


def min_eig(olib, w, kw):
    prm = abi.default_params(**dict(kw, framework=1, trust_region=1))
    wb, gb, *_ = graph_of(olib.oracle_pack_window, prm, w)
    o = oracle_lib.OracleSystem(olib, prm, gb)
    o.linearize()
    H, g, free_pt, col = _dense_system(o, gb)
    n6 = 6 * o.npf
    d = np.diag(H); s2 = 1 / (1 + np.sqrt(d)) ** 2; M = np.clip(d * s2, 1e-6, 1e32) / s2
    A = H + 1e-8 * np.diag(M)
    S = A[:n6, :n6] - A[:n6, n6:] @ np.linalg.solve(A[n6:, n6:], A[n6:, :n6]) if len(g) > n6 else A
    dS = np.sqrt(np.diag(S))
    ev, V = np.linalg.eigh(S / np.outer(dS, dS))
    o.close()
    counts = np.bincount(np.asarray(gb.obs_pose), minlength=gb.n_poses)
    weak = np.abs(V[:, 0]).reshape(-1, 6).max(axis=1)             # which free pose carries the weakest direction
    return ev[0], counts, weak


def main():
    olib = oracle_lib.load()
    for i in [int(a) for a in sys.argv[1:]]:
        w, kw = T.random_case(i)
        e0, counts, weak = min_eig(olib, w, kw)
        print(f"seed {i}: {kw}; smallest eigenvalue of the scaled reduced system {e0:.3e}; observations per pose {counts.tolist()}")
        print(f"   weight of the weakest direction per free pose: {np.round(weak, 3).tolist()}")
        for its in (1, kw["iterations"]):
            prm = abi.default_params(**dict(kw, framework=1, trust_region=1, iterations=its))
            wb_o = abi.WindowBuffers(w)
            rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
            olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
            n = rb_o.struct.n_poses_out
            for small in ("1", "0"):
                os.environ["VISFS_BA_SMALL_SOLVE"] = small
                s = backend.Solver(prm); rc, rb = s.solve_window(abi.WindowBuffers(w)); s.close()
                dt = np.abs(rb.pose_Twr_out[:n].reshape(n, 3, 4)[:, :, 3] - rb_o.pose_Twr_out[:n].reshape(n, 3, 4)[:, :, 3]).max(axis=1)
                print(f"   {its:2d} iteration(s), small-solve {small}: |dt| per pose {' '.join('%.1e' % v for v in dt)}   chi2 {rb.struct.chi2_final:.9g} / {rb_o.struct.chi2_final:.9g}")
            os.environ.pop("VISFS_BA_SMALL_SOLVE", None)


if __name__ == "__main__":
    main()
