"""Generates tests/golden/*.npz — small regression fixtures (inputs + expected outputs) for the BA hot path.

Provenance: the reference holds no golden vectors for Optimizer::localOptimize and cannot be built or imported here
(SURVEY.md §8c), so these vectors are produced by THIS repository's CPU oracle (oracle/, a restatement of the reference's
algorithm) from synthetic inputs, and pinned by tests/test_oracle_pinning.py (hand-derived known answers, finite differences).
They are data only.  Regenerate with:  python tests/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)

import oracle_lib
from helpers import ragged_window
from visfs_amd import abi, synth

INPUT_KEYS = ["root_id", "pose_ids", "pose_Twr", "link_from", "link_to", "link_T", "n_cameras", "fx", "fy", "cx", "cy",
              "baseline", "Trc", "point_ids", "point_xyz", "point_fixed", "ref_feature", "ref_pose", "ref_u", "ref_v", "ref_depth"]

CASES = {
    "prod_pcg": (lambda: synth.make_window("PROD"), dict(iterations=10, solver=2)),
    "c1_direct": (lambda: synth.make_window("C1"), dict(iterations=20, solver=0)),
    "ragged_odo_pcg": (lambda: ragged_window(seed=7), dict(iterations=20, solver=2)),
}


def main():
    olib = oracle_lib.load()
    os.makedirs(os.path.join(HERE, "golden"), exist_ok=True)
    for name, (mk, kw) in CASES.items():
        w = mk()
        prm = abi.default_params(**kw)
        wb = abi.WindowBuffers(w); rb = abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs)
        rc = olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1)
        n = rb.struct.n_poses_out
        out = dict(status=rc, pose_ids_out=rb.pose_ids_out[:n], pose_Twr_out=rb.pose_Twr_out[:n], point_xyz_out=wb.point_xyz,
                   outliers=np.array(rb.outliers(), dtype=np.uint64).reshape(-1, 2), iterations_run=np.array(list(rb.struct.iterations_run)),
                   chi2=np.array([rb.struct.chi2_initial, rb.struct.chi2_phase1, rb.struct.chi2_final]),
                   params=np.array([kw.get("iterations", 10), kw.get("solver", 0)]))
        ins = {"in_" + k: np.asarray(w[k]) for k in INPUT_KEYS}
        np.savez_compressed(os.path.join(HERE, "golden", name + ".npz"), **ins, **{"out_" + k: np.asarray(v) for k, v in out.items()})
        print(name, "status", rc, "iters", out["iterations_run"], "outliers", len(out["outliers"]))


if __name__ == "__main__":
    main()
