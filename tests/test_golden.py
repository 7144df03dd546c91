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
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(f).startswith(("g2o_", "ref_ceres_")))


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


# ---------------------------------------------------------------- the g2o cross-check hand-off (tools/g2o_crosscheck.cpp)
GRAPHS = sorted(glob.glob(os.path.join(HERE, "golden", "graphs", "*.vbag")))
G2O_FILES = sorted(glob.glob(os.path.join(HERE, "golden", "g2o_*.npz")) + glob.glob(os.path.join(HERE, "golden", "ref_ceres_*.npz")))


def test_graph_dumps_are_what_the_tools_generate(hiplib, tmp_path):
    """tests/golden/graphs/*.vbag (inputs for a machine with the real g2o) are exactly what tools/dump_graphs.py writes from the
    synthetic windows through the product's host graph build, and the format round-trips byte for byte."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import dump_graphs
    from visfs_amd import graphio
    assert len(GRAPHS) == len(dump_graphs.CASES) >= 6
    dump_graphs.main(str(tmp_path))
    for path in GRAPHS:
        name = os.path.basename(path)
        committed = open(path, "rb").read()
        assert committed == open(tmp_path / name, "rb").read(), name
        prm, gb = graphio.load_graph(path)
        graphio.dump_graph(tmp_path / ("again_" + name), prm, gb)
        assert committed == open(tmp_path / ("again_" + name), "rb").read()


@pytest.mark.parametrize("path", GRAPHS, ids=[os.path.basename(f) for f in GRAPHS])
def test_oracle_solves_the_dumped_graphs_and_results_round_trip(olib, path, tmp_path):
    import oracle_lib
    from visfs_amd import graphio
    prm, gb = graphio.load_graph(path)
    o = oracle_lib.OracleSystem(olib, prm, gb)
    rc, st, _ = o.optimize()
    pose, pts, outl, chi = o.download(); o.close()
    assert rc == abi.OK and st.iterations_run[0] >= 1
    out = tmp_path / "r.vbar"
    graphio.dump_result(out, rc, list(st.iterations_run), st.n_outliers, (st.chi2_initial, st.chi2_phase1, st.chi2_final), pose, pts, outl, chi,
                        "oracle (round-trip test)")
    r = graphio.load_result(out)
    assert r["status"] == rc and r["n_outliers"] == st.n_outliers and r["provenance"].startswith("oracle")
    assert np.array_equal(r["pose_tq"], pose) and np.array_equal(r["point_xyz"], pts) and np.array_equal(r["obs_outlier"], outl)


@pytest.mark.skipif(not G2O_FILES, reason="PARITY UNPINNED: no machine with the real g2o / Ceres has run tools/g2o_crosscheck.cpp / ceres_crosscheck.cpp "
                                          "on tests/golden/graphs/*.vbag yet (neither library is in this image or on the GPU box: profiles/r02_probe_box.log)")
@pytest.mark.parametrize("path", G2O_FILES, ids=[os.path.basename(f) for f in G2O_FILES])
def test_oracle_matches_real_g2o_fixture(olib, path):
    """Fixtures imported by tools/g2o_golden_import.py from a run of the real g2o: the oracle must reproduce them."""
    import oracle_lib
    from visfs_amd import graphio
    z = np.load(path)
    prm, gb = graphio.load_graph(os.path.join(HERE, "golden", "graphs", str(z["graph"])))
    o = oracle_lib.OracleSystem(olib, prm, gb)
    rc, st, _ = o.optimize()
    pose, pts, outl, chi = o.download(); o.close()
    assert rc == int(z["status"]) and list(st.iterations_run) == list(z["iterations_run"])
    assert np.array_equal(outl, z["obs_outlier"])
    assert rel_err(pose, z["pose_tq"]) < 1e-6 and rel_err(pts, z["point_xyz"]) < 1e-6
    assert rel_err([st.chi2_initial, st.chi2_phase1, st.chi2_final], z["chi2"]) < 1e-6
