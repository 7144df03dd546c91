"""Optimizer/Framework=1 on the GPU: the Ceres branch of localOptimize (Optimizer.cpp:366-593) through the HIP path against the CPU
oracle's restatement of it (tests/test_ceres_flavour.py says what pins that restatement).  Same kernels as the g2o branch — the
objective differs by the weights (1 / var^2), the damping is LevenbergMarquardtStrategy's per-variable diagonal, the control is
Ceres' trust-region loop (k_ceres_lin_finalize, ceres_decide) — so the bar is the same: identical iteration / step decisions and
radius trace, identical outlier sets, poses and landmarks to rounding."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from helpers import graph_of, hard_window, ragged_window, rel_err
from test_gpu_parity import check_optimize, make_pair, solve_both
from visfs_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["C1", "PROD", "C3s", "LASER", "HARD", "RAGGED", "C2"])
def test_ceres_branch_matches_the_oracle(olib, case):
    kw = dict(framework=1, iterations=20 if case != "C2" else 10)
    if case == "LASER":
        w = synth.make_laser_window(with_visual=True, n_points=400)
    elif case == "HARD":
        w = hard_window()
    elif case == "RAGGED":
        w = ragged_window(seed=7)
    elif case == "C3s":
        w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)       # its odometry links are not part of the Ceres problem
    else:
        w = synth.make_window(case)
    o, s, gb = make_pair(olib, w, **kw)
    st = check_optimize(o, s, pose_tol=1e-6)
    assert st.iterations_run[1] == 0 and st.iterations_run[0] >= 1
    if case == "HARD":
        radius = np.array([st.trace_lambda[i] for i in range(st.n_trace)])
        assert (np.diff(radius) < 0).any()                               # the window really rejects steps
    a = s.download()
    s.reset(); s.optimize()
    assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, s.download()))   # run-to-run bitwise identical
    s.close(); o.close()


def test_ceres_branch_laser_only_window(olib):
    w = synth.make_laser_window(with_visual=False, n_points=1000)
    o, s, gb = make_pair(olib, w, framework=1, iterations=10)
    check_optimize(o, s, pose_tol=1e-6)
    s.close(); o.close()


def test_ceres_branch_window_level_and_batch_entry(olib):
    from visfs_amd import backend
    w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    w["point_ids"] = np.r_[np.asarray(w["point_ids"]), np.uint64(77777)]       # a point without references → NaN on return
    w["point_xyz"] = np.vstack([w["point_xyz"], [[1.0, 2.0, 3.0]]]); w["point_fixed"] = np.r_[w["point_fixed"], np.uint8(0)]
    rc_o, wb_o, rb_o, rc_g, wb_g, rb_g = solve_both(olib, w, framework=1, iterations=10)
    assert rc_o == rc_g == abi.OK and rb_g.struct.n_poses_out == rb_o.struct.n_poses_out == 12
    et, er = synth.pose_errors(rb_g.pose_Twr_out[:12], rb_o.pose_Twr_out[:12])
    assert et < 1e-6 and er < 1e-6
    assert rb_g.outliers() == rb_o.outliers() and len(rb_g.outliers()) > 0
    assert np.isnan(wb_g.point_xyz[-1]).all() and rel_err(wb_g.point_xyz[:-1], wb_o.point_xyz[:-1]) < 1e-6
    assert list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)
    # visfs_ba_solve_batch: Ceres-flavour windows are solved one after another, same results
    prm = abi.default_params(framework=1, iterations=10)
    s = backend.Solver(prm)
    ws = [synth.make_window("custom", n_kf=10, n_lm=200, n_obs=1600, seed=60 + i) for i in range(3)]
    got = s.solve_batch([abi.WindowBuffers(x) for x in ws])
    for x, r in zip(ws, got):
        rc1, r1 = s.solve_window(abi.WindowBuffers(x))
        assert rc1 == abi.OK and r.struct.status == abi.OK
        assert np.array_equal(r.pose_Twr_out, r1.pose_Twr_out) and r.outliers() == r1.outliers()
    s.close()


def test_stage_hooks_are_g2o_only(olib):
    from visfs_amd import backend
    w = synth.make_window("C1")
    prm = abi.default_params(framework=1)
    s = backend.Solver(prm)
    gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s.upload(gb)
    chi, md = C.c_double(), C.c_double()
    assert s.lib.visfs_ba_stage_linearize(s.h, C.byref(chi), C.byref(md)) == abi.ERR_UNSUPPORTED
    s.close()


# ------------------------------------------------------------------ Optimizer/TrustRegion=1 under Framework=1: DOGLEG (Optimizer.cpp:515-519)
@pytest.mark.parametrize("case", ["C1", "PROD", "C3s", "LASER", "HARD", "HARD60", "RAGGED", "C2"])
def test_ceres_dogleg_matches_the_oracle(olib, case):
    """k_backsub<DL=1> -> k_dogleg_mid -> k_backsub<DL=2> against the oracle's ceres_dogleg_step: the same accept / reject decisions, the
    same radius trace (HARD: steps on the segment Cauchy -> Gauss-Newton and rejected steps, see the oracle's VISFS_ORACLE_TRACE)."""
    kw = dict(framework=1, trust_region=1, iterations={"C2": 10, "HARD60": 60}.get(case, 20))
    if case == "LASER":
        w = synth.make_laser_window(with_visual=True, n_points=400)
    elif case.startswith("HARD"):
        w = hard_window()
    elif case == "RAGGED":
        w = ragged_window(seed=7)
    elif case == "C3s":
        w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    else:
        w = synth.make_window(case)
    o, s, gb = make_pair(olib, w, **kw)
    st = check_optimize(o, s, pose_tol=1e-6)
    assert st.iterations_run[1] == 0 and st.iterations_run[0] >= 1
    if case.startswith("HARD"):
        radius = np.array([st.trace_lambda[i] for i in range(st.n_trace)])
        assert (np.diff(radius) < 0).any() and radius[0] > 1e4                  # halved by rejected / poor steps; the first good step grew it to 3 x its length
    a = s.download()
    s.reset(); s.optimize()
    assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, s.download()))   # run-to-run bitwise identical
    s.close(); o.close()


def test_ceres_dogleg_differs_from_levenberg_marquardt_where_it_should(olib):
    """The strategies are different algorithms: on the hard window they take different paths (another radius trace, another final cost),
    so a dogleg flag that silently ran the LM strategy would not pass the test above."""
    from visfs_amd import backend
    w = hard_window()
    out = []
    for tr in (0, 1):
        s = backend.Solver(abi.default_params(framework=1, trust_region=tr, iterations=20))
        rc, rb = s.solve_window(abi.WindowBuffers(w))
        assert rc == abi.OK
        out.append((rb.struct.chi2_final, rb.pose_Twr_out.copy()))
        s.close()
    assert out[0][0] != out[1][0] and not np.array_equal(out[0][1], out[1][1])


def test_ceres_dogleg_laser_only_and_window_entry(olib):
    from visfs_amd import backend
    w = synth.make_laser_window(with_visual=False, n_points=1000)
    o, s, gb = make_pair(olib, w, framework=1, trust_region=1, iterations=10)
    check_optimize(o, s, pose_tol=1e-6)
    s.close(); o.close()
    w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    rc_o, wb_o, rb_o, rc_g, wb_g, rb_g = solve_both(olib, w, framework=1, trust_region=1, iterations=10)
    assert rc_o == rc_g == abi.OK and rb_g.struct.n_poses_out == rb_o.struct.n_poses_out == 12
    et, er = synth.pose_errors(rb_g.pose_Twr_out[:12], rb_o.pose_Twr_out[:12])
    assert et < 1e-6 and er < 1e-6 and rb_g.outliers() == rb_o.outliers()
    assert list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)
    # visfs_ba_solve_batch: the same results as single solves
    prm = abi.default_params(framework=1, trust_region=1, iterations=10)
    s = backend.Solver(prm)
    ws = [synth.make_window("PROD", window_index=i) for i in range(3)]
    got = s.solve_batch([abi.WindowBuffers(x) for x in ws])
    for x, r in zip(ws, got):
        rc1, r1 = s.solve_window(abi.WindowBuffers(x))
        assert rc1 == abi.OK and r.struct.status == abi.OK
        assert np.array_equal(r.pose_Twr_out, r1.pose_Twr_out) and r.outliers() == r1.outliers()
    s.close()


@pytest.mark.parametrize("i", range(16))
def test_random_window_matches_oracle_ceres_dogleg(olib, i):
    from visfs_amd import backend
    import test_gpu_random as T
    try:
        w, kw = T.random_case(100 + i)
    except ValueError:
        pytest.skip("this seed's shape is not generated (tracks longer than the window)")
    prm = abi.default_params(**dict(kw, framework=1, trust_region=1))
    wb_o, wb_g = abi.WindowBuffers(w), abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    s = backend.Solver(prm)
    rc_g, rb_g = s.solve_window(wb_g)
    s.close()
    assert rc_g == rc_o and rb_g.struct.n_poses_out == rb_o.struct.n_poses_out
    assert rb_g.outliers() == rb_o.outliers()
    assert list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)
    if rc_o == abi.OK:
        n = rb_o.struct.n_poses_out
        et, er = synth.pose_errors(rb_g.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
        assert et < 1e-7 and er < 1e-7, (et, er, kw)
        assert rel_err(wb_g.point_xyz, wb_o.point_xyz) < 1e-6


@pytest.mark.parametrize("i", range(16))
def test_random_window_matches_oracle_ceres_branch(olib, i):
    """The random windows of test_gpu_random (ragged tracks, fixed fractions, laser, delta 8 / 2 / 0, iteration caps 2 ... 20) through the
    Ceres branch at the window level."""
    from visfs_amd import backend
    import test_gpu_random as T
    w, kw = T.random_case(i)
    prm = abi.default_params(**dict(kw, framework=1, trust_region=0))
    wb_o, wb_g = abi.WindowBuffers(w), abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    s = backend.Solver(prm)
    rc_g, rb_g = s.solve_window(wb_g)
    s.close()
    assert rc_g == rc_o and rb_g.struct.n_poses_out == rb_o.struct.n_poses_out
    assert rb_g.outliers() == rb_o.outliers()
    assert list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)
    if rc_o == abi.OK:
        n = rb_o.struct.n_poses_out
        et, er = synth.pose_errors(rb_g.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
        assert et < 1e-7 and er < 1e-7, (et, er, kw)
        assert rel_err(wb_g.point_xyz, wb_o.point_xyz) < 1e-6


@pytest.mark.parametrize("trust_region", [0, 1])
def test_ceres_branch_in_batched_launches(olib, trust_region):
    """Optimizer/Framework=1 through visfs_ba_solve_batch: windows whose reduced system is small (k_small_solve) or banded (k_band_chol)
    share every launch — k_ceres_lin_finalize after every linearisation, the per-variable damping in the Schur gather, one pass of
    <= Iterations trust-region iterations, no second phase; with the DOGLEG strategy (trust_region = 1) the two back-substitution passes
    and k_dogleg_mid as well — and come out as the bytes of their single-window solves."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=12, framework=1, trust_region=trust_region)
    ws = [synth.make_window("PROD", window_index=i) for i in range(5)]
    ws += [synth.make_window("custom", n_kf=24, n_lm=500, n_obs=4000, seed=400 + i) for i in range(4)]
    ws.append(hard_window())
    s = backend.Solver(prm)
    singles = []
    for w in ws:
        wb = abi.WindowBuffers(w)
        rc, rb = s.solve_window(wb)
        singles.append((rc, rb, wb))
    wbs = [abi.WindowBuffers(w) for w in ws]
    rbs = s.solve_batch(wbs)
    for (rc, a, wa), b, wb in zip(singles, rbs, wbs):
        assert rc == abi.OK and b.struct.status == rc
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run)
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert np.array_equal(wa.point_xyz, wb.point_xyz, equal_nan=True)
    s.close()
