"""Soak of the Ceres branch (Optimizer/Framework=1): the random windows of tests/test_gpu_random.random_case (shape, raggedness, fixed
fractions, laser, robust delta, iteration cap) through visfs_ba_solve_window against the oracle's restatement of the branch.  Classes:
`exact` (same status, iteration counts, outlier lists; poses to 1e-7), `converged` (poses / chi2 agree to 1e-9 but the minimizer, already
at a tolerance, stopped an iteration apart: |cost change| against 1e-6 x cost decides on a 1e-13 difference), FAILED (must stay empty).

DOGLEG (third argument 1) solves the Gauss-Newton system regularised by mu = 1e-8 only, so two more classes exist there, each decided by a
measurement of the PROBLEM, not by loosening the tolerance:
`rank-deficient`: the Jacobi-scaled reduced system at the start has an eigenvalue below 1e-6 (a pose that sees too few landmarks, or a part
of the window without any fixed vertex: an exact gauge freedom) — the step along that direction is rounding noise / mu in every
implementation (profiles/r03_stage_precision.log: the checker's own solve is 5e-2 away from a dense NumPy solve of its own S there);
`sensitive`: the checker, run again on the same window with the landmark coordinates changed by 1e-14 (relative), moves by more than a
third of the distance between the HIP path and the checker (a long trajectory with rejected steps that amplifies the last bit).

usage: python tools/soak_ceres.py 0 300 [trust_region: 0 = LEVENBERG_MARQUARDT (default), 1 = DOGLEG]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib
import test_gpu_random as T
from visfs_amd import abi, backend, synth


def smallest_scaled_eigenvalue(olib, w, kw):
    """Smallest eigenvalue of diag(S)^-1/2 S diag(S)^-1/2, S = the reduced system of (H + 1e-8 M) at the start (dense NumPy)."""
    from helpers import graph_of
    from test_ceres_flavour import _dense_system
    prm = abi.default_params(**dict(kw, framework=1, trust_region=1))
    wb, gb, *_ = graph_of(olib.oracle_pack_window, prm, w)
    o = oracle_lib.OracleSystem(olib, prm, gb)
    o.linearize()
    H, g, free_pt, col = _dense_system(o, gb)
    n6 = 6 * o.npf
    o.close()
    d = np.diag(H); s2 = 1.0 / (1.0 + np.sqrt(d)) ** 2; M = np.clip(d * s2, 1e-6, 1e32) / s2
    A = H + 1e-8 * np.diag(M)
    S = A[:n6, :n6] - A[:n6, n6:] @ np.linalg.solve(A[n6:, n6:], A[n6:, :n6]) if len(g) > n6 else A
    dS = np.sqrt(np.diag(S))
    return float(np.linalg.eigvalsh(S / np.outer(dS, dS))[0])


def classify(olib, i, trust_region=0):
    try:
        w, kw = T.random_case(i)
    except ValueError:
        return "not generated", None
    kw = dict(kw, framework=1, trust_region=trust_region)
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
    if trust_region == 1:
        e0 = smallest_scaled_eigenvalue(olib, w, kw)
        if e0 < 1e-6:
            return f"rank-deficient (smallest eigenvalue {e0:.1e}, et {et:.1e} er {er:.1e})", kw
        w2 = dict(w)
        w2["point_xyz"] = np.asarray(w["point_xyz"]) * (1.0 + 1e-14 * np.random.default_rng(0).normal(size=np.asarray(w["point_xyz"]).shape))
        wb_p = abi.WindowBuffers(w2)
        rb_p = abi.ResultBuffers(wb_p.struct.n_poses, wb_p.struct.n_refs)
        olib.oracle_solve_window(C.byref(prm), C.byref(wb_p.struct), C.byref(rb_p.struct), 1)
        st, sr = synth.pose_errors(rb_p.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
        if same_out and st > et / 3 and sr > er / 3:
            return f"sensitive (checker against itself {st:.1e} {sr:.1e}, HIP against checker {et:.1e} {er:.1e})", kw
    return f"FAILED (et {et:.1e} er {er:.1e} chi {chi_rel:.1e} outliers {same_out} iterations {list(rb_g.struct.iterations_run)} / {list(rb_o.struct.iterations_run)})", kw


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    tr = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    olib = oracle_lib.load()
    count, failed = {}, []
    for i in range(lo, hi):
        c, kw = classify(olib, i, tr)
        key = c.split(" (")[0]
        count[key] = count.get(key, 0) + 1
        if key not in ("exact", "not generated"):
            print(f"case {i}: {c} {kw}", flush=True)
        if key == "FAILED":
            failed.append(i)
        if (i - lo) % 100 == 99:
            print(f"... {i + 1 - lo} cases: {count}", flush=True)
    print(f"ceres soak {lo}..{hi} (trust region strategy {tr}): {count}; failed seeds: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
