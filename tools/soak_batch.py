"""Soak: visfs_ba_solve_batch against one visfs_ba_solve_window per window over random windows (tests/test_gpu_random.py's
generator): every output must be bit-identical — batched launches, per-window gates and the one-read schedule change nothing.
(Batches of 12 by default: from 16 windows on, the members of a batch run the single-workgroup PCG kernel, whose sums associate
differently — such batches equal single solves to rounding, tests/test_gpu_workloads.py.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gpu_random as T
from visfs_amd import abi, backend


def main():
    lo, hi, bsz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 12
    by_prm = {}
    for i in range(lo, hi):
        try:
            w, kw = T.random_case(i)
        except ValueError:
            continue
        by_prm.setdefault(tuple(sorted(kw.items())), []).append((i, w))
    n_ok, n_round, bad = 0, 0, []
    for key, cases in by_prm.items():
        prm = abi.default_params(**dict(key))
        s = backend.Solver(prm)
        for c0 in range(0, len(cases), bsz):
            chunk = cases[c0:c0 + bsz]
            singles = []
            for i, w in chunk:
                wb = abi.WindowBuffers(w); rc, rb = s.solve_window(wb); singles.append((rc, rb, wb))
            wbs = [abi.WindowBuffers(w) for _, w in chunk]
            rbs = s.solve_batch(wbs)
            for (i, _), (rc, a, wa), b, wb in zip(chunk, singles, rbs, wbs):
                same = (b.struct.status == rc and b.struct.n_poses_out == a.struct.n_poses_out
                        and np.array_equal(a.pose_Twr_out, b.pose_Twr_out, equal_nan=True) and a.outliers() == b.outliers()
                        and np.array_equal(wa.point_xyz, wb.point_xyz, equal_nan=True)
                        and list(a.struct.iterations_run) == list(b.struct.iterations_run)
                        and (a.struct.chi2_final == b.struct.chi2_final or (a.struct.chi2_final != a.struct.chi2_final and b.struct.chi2_final != b.struct.chi2_final)))
                if same:
                    n_ok += 1
                elif bsz >= 16 and b.struct.status == rc and a.outliers() == b.outliers() and list(a.struct.iterations_run) == list(b.struct.iterations_run) \
                        and np.allclose(a.pose_Twr_out, b.pose_Twr_out, rtol=0, atol=1e-9, equal_nan=True):
                    n_round += 1                       # batches of >= 16 windows run the single-workgroup PCG: equal to rounding
                else:
                    bad.append(i); print(f"case {i} differs: {dict(key)}", flush=True)
        s.close()
    print(f"batch soak {lo}..{hi} (batches of {bsz}): {n_ok} identical, {n_round} equal to rounding (k_pcg_cu in batches of >= 16), differing seeds: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
