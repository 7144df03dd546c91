"""Soak of the Ceres branch (Optimizer/Framework=1): the random windows of tests/test_gpu_random.random_case (shape, raggedness, fixed
fractions, laser, robust delta, iteration cap) through visfs_ba_solve_window against the oracle's restatement of the branch.  Classes:
`exact` (same status, iteration counts, outlier lists; poses to 1e-7), `converged` (poses / chi2 agree to 1e-9 but the minimizer, already
at a tolerance, stopped an iteration apart: |cost change| against 1e-6 x cost decides on a 1e-13 difference), FAILED (must stay empty).

usage: python tools/soak_ceres.py 0 300"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib
import test_gpu_random as T
from visfs_amd import abi, backend, synth


def classify(olib, i):
    try:
        w, kw = T.random_case(i)
    except ValueError:
        return "not generated", None
    kw = dict(kw, framework=1, trust_region=0)
    prm = abi.default_params(**kw)
    wb_o, wb_g = abi.WindowBuffers(w), abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    s = backend.Solver(prm); rc_g, rb_g = s.solve_window(wb_g); s.close()
    if rc_o != rc_g or rb_o.struct.n_poses_out != rb_g.struct.n_poses_out:
        return "FAILED", kw
    if rc_o != abi.OK:
        return "exact", kw
    n = rb_o.struct.n_poses_out
    et, er = synth.pose_errors(rb_g.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
    chi_rel = abs(rb_g.struct.chi2_final - rb_o.struct.chi2_final) / max(abs(rb_o.struct.chi2_final), 1e-9)
    same_out = rb_g.outliers() == rb_o.outliers()
    same_it = list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)
    if same_out and same_it and et < 1e-7 and er < 1e-7:
        return "exact", kw
    if same_out and et < 1e-9 and er < 1e-9 and chi_rel <= 1e-9:
        return "converged", kw
    return f"FAILED (et {et:.1e} er {er:.1e} chi {chi_rel:.1e} outliers {same_out} iterations {list(rb_g.struct.iterations_run)} / {list(rb_o.struct.iterations_run)})", kw


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    olib = oracle_lib.load()
    count, failed = {}, []
    for i in range(lo, hi):
        c, kw = classify(olib, i)
        key = c.split(" (")[0]
        count[key] = count.get(key, 0) + 1
        if key not in ("exact", "not generated"):
            print(f"case {i}: {c} {kw}", flush=True)
        if key == "FAILED":
            failed.append(i)
        if (i - lo) % 100 == 99:
            print(f"... {i + 1 - lo} cases: {count}", flush=True)
    print(f"ceres soak {lo}..{hi}: {count}; failed seeds: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
