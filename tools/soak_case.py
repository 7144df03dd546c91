"""One seed of tools/soak_random.py in detail: status codes, iteration counts, pose / chi2 differences, outlier sets — for the default
solver kernels and for the alternatives (VISFS_BA_BAND=0: dense Cholesky; VISFS_BA_SPEC_FUSED=0; VISFS_BA_SPEC=0).
usage: python tools/soak_case.py 3363"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib
import test_gpu_random as T
from visfs_amd import abi, backend, synth


def main():
    olib = oracle_lib.load()
    i = int(sys.argv[1])
    w, kw = T.random_case(i)
    print("seed", i, kw, "poses", len(w["pose_ids"]), "points", len(w["point_ids"]), "refs", len(w["ref_feature"]), "fixed points", int(np.asarray(w["point_fixed"]).sum()))
    prm = abi.default_params(**kw)
    wb_o = abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    print("oracle: rc", rc_o, "iterations", list(rb_o.struct.iterations_run), "chi2", rb_o.struct.chi2_initial, rb_o.struct.chi2_phase1, rb_o.struct.chi2_final, "outliers", rb_o.struct.n_outliers)
    for env in ({}, {"VISFS_BA_BAND": "0"}, {"VISFS_BA_SPEC_FUSED": "0"}, {"VISFS_BA_SPEC": "0"}, {"VISFS_BA_SMALL_SOLVE": "0"}):
        for k, v in env.items():
            os.environ[k] = v
        s = backend.Solver(prm)
        wb_g = abi.WindowBuffers(w)
        rc_g, rb_g = s.solve_window(wb_g)
        info = s.describe()
        s.close()
        for k in env:
            os.environ.pop(k)
        n = min(rb_o.struct.n_poses_out, rb_g.struct.n_poses_out)
        et, er = synth.pose_errors(rb_g.pose_Twr_out[:n], rb_o.pose_Twr_out[:n]) if n else (0, 0)
        print(f"HIP {env}: rc {rc_g} kernel {info.get('solver_kernel')} band {info.get('band_blocks')} iterations {list(rb_g.struct.iterations_run)} chi2 {rb_g.struct.chi2_initial:.9g} {rb_g.struct.chi2_phase1:.9g} {rb_g.struct.chi2_final:.9g} "
              f"outliers {rb_g.struct.n_outliers} same {rb_g.outliers() == rb_o.outliers()} | pose err {et:.2e} {er:.2e}")


if __name__ == "__main__":
    main()
