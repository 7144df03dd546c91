"""Golden fixtures (tests/golden/*.npz, made by tests/make_golden.py from this repository's oracle):
the oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from helpers import rel_err
from visfs_amd import abi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load_case(path):
    z = np.load(path)
    w = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    for k in ("root_id", "n_cameras"):
        w[k] = int(w[k])
    for k in ("fx", "fy", "cx", "cy", "baseline"):
        w[k] = float(w[k])
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    prm = abi.default_params(iterations=int(out["params"][0]), solver=int(out["params"][1]))
    return w, out, prm


def check(out, rc, wb, rb, tol):
    n = rb.struct.n_poses_out
    assert rc == int(out["status"]) and n == len(out["pose_ids_out"])
    assert np.array_equal(rb.pose_ids_out[:n], out["pose_ids_out"])
    et, er = synth.pose_errors(rb.pose_Twr_out[:n], out["pose_Twr_out"])
    assert et < tol and er < tol
    assert [tuple(x) for x in out["outliers"]] == rb.outliers()
    assert rel_err(wb.point_xyz, out["point_xyz_out"]) < tol
    assert list(rb.struct.iterations_run) == list(out["iterations_run"])
    assert rel_err([rb.struct.chi2_initial, rb.struct.chi2_phase1, rb.struct.chi2_final], out["chi2"]) < 1e-7


def test_fixtures_exist():
    assert len(FILES) >= 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(olib, path):
    w, out, prm = load_case(path)
    wb = abi.WindowBuffers(w); rb = abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs)
    rc = olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1)
    check(out, rc, wb, rb, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_matches_golden(path):
    from visfs_amd import backend
    w, out, prm = load_case(path)
    wb = abi.WindowBuffers(w)
    s = backend.Solver(prm)
    rc, rb = s.solve_window(wb)
    check(out, rc, wb, rb, 1e-6)
    s.close()
