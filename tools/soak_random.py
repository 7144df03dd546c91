"""Soak: the randomised window parity of tests/test_gpu_random.py over many more seeds than the suite runs (one process).
Every seed is classified: exact (the suite's criteria), benign mismatches the reference algorithm itself produces —
`converged` (chi2 agrees to 1e-9 and the outlier sets are equal, but the LM loop, already at machine precision, stopped some
iterations apart: the sign of a 1e-13 chi2 change decides; the poses agree to the suite's own 1e-7 — seed 3363 is the case that
set this bound: 8e-9 between the HIP path and the oracle with EITHER direct solver, the dense one running the oracle's 10 + 10
iterations, the banded one 10 + 4, `tools/soak_case.py 3363`), `pcg tolerance` (Solver=2 stops at a RELATIVE residual of 1e-6: on an ill-conditioned window two
correct implementations agree to about that, not to the suite's 1e-7), `gauss-newton` (trust region 1 is undamped: after the
outlier pass has culled a landmark's observations the system is rank-deficient and either implementation returns rounding
noise; the oracle's own chi2 often RISES) — or FAILED, which must stay empty."""
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
    try:
        T.test_random_window_matches_oracle(olib, i)
        return "exact", kw
    except AssertionError:
        pass
    prm = abi.default_params(**kw)
    wb_o, wb_g = abi.WindowBuffers(w), abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    s = backend.Solver(prm); rc_g, rb_g = s.solve_window(wb_g); s.close()
    if rc_o != rc_g or rb_o.struct.n_poses_out != rb_g.struct.n_poses_out:
        # (undamped Gauss-Newton trajectories that have drifted apart may hit the NaN / 1e12 guards of Optimizer.cpp:262-268 in one
        # implementation only: seeds 756, 781, 1102, 1108 agree to 1e-9 for five iterations, then chi2 rises and the paths split)
        return ("gauss-newton" if kw["trust_region"] == 1 else "FAILED"), kw
    n = rb_o.struct.n_poses_out
    et, er = synth.pose_errors(rb_g.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
    chi_rel = abs(rb_g.struct.chi2_final - rb_o.struct.chi2_final) / max(abs(rb_o.struct.chi2_final), 1e-9)
    same_out = rb_g.outliers() == rb_o.outliers()
    if et < 1e-7 and er < 1e-7 and chi_rel <= 1e-9 and same_out:
        return "converged", kw
    if kw["trust_region"] == 1:
        return "gauss-newton", kw
    if kw["solver"] == 2 and et < 1e-5 and er < 1e-5 and chi_rel < 1e-4 and same_out:
        return "pcg tolerance", kw
    return "FAILED", kw


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    olib = oracle_lib.load()
    count, failed = {}, []
    for i in range(lo, hi):
        c, kw = classify(olib, i)
        count[c] = count.get(c, 0) + 1
        if c not in ("exact", "not generated"):
            print(f"case {i}: {c} {kw}", flush=True)
        if c == "FAILED":
            failed.append(i)
        if (i - lo) % 100 == 99:
            print(f"... {i + 1 - lo} cases: {count}", flush=True)
    print(f"soak {lo}..{hi}: {count}; failed seeds: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
