"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (fp64 everywhere): the device sums in a different (fixed) order than the oracle and the
compiler contracts multiply-adds into FMAs, so stage outputs agree to ~1e-13 relative; bounds are
set to 1e-9 for linearisation products, 1e-7 for solved increments (conditioning of S) and 1e-6
relative on final poses (north_star demands 1e-4).  Outlier sets (a discrete output) must be identical.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from helpers import drop_refs, graph_of, hard_window, ragged_window, rel_err, twr_of
from visfs_amd import abi, synth

pytestmark = pytest.mark.gpu

LIN_BUFS = [("err", abi.BUF_OBS_ERR), ("chi2", abi.BUF_OBS_CHI2), ("weight", abi.BUF_OBS_WEIGHT), ("Hpl", abi.BUF_HPL),
            ("Hll", abi.BUF_HLL), ("bl", abi.BUF_BL), ("Hpp", abi.BUF_HPP), ("bp", abi.BUF_BP)]
TRIAL_BUFS = [("S", abi.BUF_S, 1e-9), ("bs", abi.BUF_BS, 1e-9), ("dx_pose", abi.BUF_DX_POSE, 1e-7), ("dx_point", abi.BUF_DX_POINT, 1e-7),
              ("pose_trial", abi.BUF_POSE_TRIAL, 1e-10), ("point_trial", abi.BUF_POINT_TRIAL, 1e-10)]


def make_pair(olib, w, **prm_kw):
    from visfs_amd import backend
    prm = abi.default_params(**prm_kw)
    wb, gb, used, oref, mono = graph_of(olib.oracle_pack_window, prm, w)
    o = oracle_lib.OracleSystem(olib, prm, gb)
    s = backend.Solver(prm)
    s.upload(gb)
    return o, s, gb


def check_stages(o, s, lambdas=(None, 1.0)):
    oc, omd = o.linearize(); gc, gmd = s.linearize()
    assert abs(oc - gc) <= 1e-10 * abs(oc) and abs(omd - gmd) <= 1e-10 * abs(omd)
    for name, b in LIN_BUFS:
        assert rel_err(s.fetch(b), o.fetch(b)) < 1e-9, name
    for lam in lambdas:
        lam = 1e-5 * omd if lam is None else lam * omd
        ot, gt = o.trial(lam), s.trial(lam)
        assert ot[3] == gt[3] == 1
        assert ot[2] == gt[2], "PCG iteration counts differ"
        assert abs(ot[0] - gt[0]) <= 1e-8 * abs(ot[0]) and abs(ot[1] - gt[1]) <= 1e-7 * abs(ot[1])
        for name, b, tol in TRIAL_BUFS:
            assert rel_err(s.fetch(b), o.fetch(b)) < tol, (name, lam)


def check_optimize(o, s, pose_tol=1e-6):
    o.reset(); s.reset()
    rc_o, st_o, _ = o.optimize()
    rc_g, st_g = s.optimize()
    assert rc_o == rc_g
    assert list(st_o.iterations_run) == list(st_g.iterations_run) and list(st_o.trials_run) == list(st_g.trials_run)
    assert st_o.n_outliers == st_g.n_outliers and st_o.n_trace == st_g.n_trace
    for a, b in ((st_o.chi2_initial, st_g.chi2_initial), (st_o.chi2_phase1, st_g.chi2_phase1), (st_o.chi2_final, st_g.chi2_final)):
        assert abs(a - b) <= 1e-7 * max(abs(a), 1e-12)
    lam_o = np.array([st_o.trace_lambda[i] for i in range(st_o.n_trace)]); lam_g = np.array([st_g.trace_lambda[i] for i in range(st_g.n_trace)])
    assert rel_err(lam_g, lam_o) < 1e-6                         # identical LM trajectory
    po, pto, outo, chio = o.download(); pg, ptg, outg, chig = s.download()
    assert np.array_equal(outo, outg), "outlier sets differ"
    assert rel_err(pg, po) < pose_tol and rel_err(ptg, pto) < pose_tol
    assert rel_err(chig, chio) < 1e-6
    return st_g


# ---------------------------------------------------------------- stage parity
@pytest.mark.parametrize("solver", [2, 0])
def test_stage_parity_c1(olib, solver):
    o, s, gb = make_pair(olib, synth.make_window("C1"), iterations=20, solver=solver)
    check_stages(o, s)
    s.close(); o.close()


def test_stage_parity_ragged_with_odometry(olib):
    """Ragged tracks (0, 1, many observations per landmark), wheel-odometry edges, 12 poses."""
    o, s, gb = make_pair(olib, ragged_window(seed=7), iterations=20, solver=2)
    assert gb.struct.n_odo == 11
    check_stages(o, s, lambdas=(None, 1e-2, 10.0))
    s.close(); o.close()


def test_stage_parity_long_tracks_exceed_group(olib):
    """Track length 40 > lanes per landmark: the landmark-major kernels loop over their tiles."""
    w = synth.make_window("custom", n_kf=40, n_lm=60, n_obs=2400, seed=3)
    keep = np.ones(2400, bool); keep[::7] = False            # mean track ≈ 34 → group 64; make it ragged too
    o, s, gb = make_pair(olib, drop_refs(w, keep), iterations=10, solver=2)
    check_stages(o, s)
    s.close(); o.close()


def test_stage_parity_production_window(olib):
    """6 poses x 300 features, the production shape (Parameters.h:161,148), short tracks → 4/8 lanes per landmark."""
    o, s, gb = make_pair(olib, synth.make_window("PROD"), iterations=10, solver=2)
    check_stages(o, s)
    s.close(); o.close()


# ---------------------------------------------------------------- full optimise parity
@pytest.mark.parametrize("cfg,solver", [("C1", 2), ("C1", 0), ("PROD", 2), ("C3", 2), ("C2", 2), ("C2", 0)])
def test_optimize_parity(olib, cfg, solver):
    o, s, gb = make_pair(olib, synth.make_window(cfg), iterations=20, solver=solver)
    st = check_optimize(o, s)
    assert st.chi2_final < 0.02 * st.chi2_initial
    s.close(); o.close()


@pytest.mark.parametrize("n_kf,n_lm,n_obs", [(100, 1000, 8000), (257, 1500, 9000)])
def test_pcg_register_block_variants(olib, n_kf, n_lm, n_obs):
    """The persistent PCG keeps ceil(Npf/64) 6-blocks per lane in registers: 1 (C1..C3), 2 (99 free poses) and
    4 (256 free poses, the maximum; also C4) blocks per lane are separate template instances."""
    w = synth.make_window("custom", n_kf=n_kf, n_lm=n_lm, n_obs=n_obs, seed=17)
    o, s, gb = make_pair(olib, w, iterations=6, solver=2)
    check_stages(o, s)
    check_optimize(o, s, pose_tol=1e-5)
    s.close(); o.close()


@pytest.mark.parametrize("n_kf,solver", [(300, 2), (700, 2), (700, 0), (900, 2)])
def test_windows_beyond_the_former_size_limits(olib, n_kf, solver):
    """The reference API takes maps of any size (Optimizer.h:46-56).  More than 256 free poses: a PCG workgroup owns several
    block rows of S (2 / 3 / 4 here) and an owner thread 2 or 4 six-blocks, the grid stays within one workgroup per CU.  More
    than 840 poses (900): the landmark kernels read the poses from HBM instead of staging all of them as R|t in LDS.  Same
    parity bar as every other window: stage buffers, LM trajectory, PCG iteration counts, outliers, final states."""
    w = synth.make_window("custom", n_kf=n_kf, n_lm=5 * n_kf, n_obs=30 * n_kf, seed=17)
    o, s, gb = make_pair(olib, w, iterations=4, solver=solver)
    assert s.describe()["n_free_poses"] == n_kf - 1
    check_stages(o, s)
    check_optimize(o, s, pose_tol=1e-5)
    s.close(); o.close()


def test_pcg_refuses_only_what_it_cannot_hold(olib):
    """Above 1024 free poses the persistent PCG is refused (never hung); the direct solver takes such windows."""
    from visfs_amd import backend
    w = synth.make_window("custom", n_kf=1030, n_lm=3000, n_obs=12000, seed=17)
    prm = abi.default_params(iterations=2, solver=2)
    wb, gb, *_ = graph_of(olib.oracle_pack_window, prm, w)
    s = backend.Solver(prm)
    with pytest.raises(backend.BackendError, match="1024 free poses"):
        s.upload(gb)
    s.close()


def test_optimize_parity_default_iterations(olib):
    """Reference defaults: Iterations=10 → 5+5 (Parameters.h:187)."""
    o, s, gb = make_pair(olib, synth.make_window("C1"))
    st = check_optimize(o, s)
    assert list(st.iterations_run) == [5, 5]
    s.close(); o.close()


def test_optimize_parity_with_rejected_steps(olib):
    """The lambda *= ni / pop path of the LM loop: trials > iterations, identical trajectory on both sides."""
    o, s, gb = make_pair(olib, hard_window(), iterations=20, solver=2)
    st = check_optimize(o, s, pose_tol=1e-5)
    assert st.trials_run[0] > st.iterations_run[0]
    s.close(); o.close()


def test_optimize_parity_ragged_odometry_direct(olib):
    o, s, gb = make_pair(olib, ragged_window(seed=21), iterations=20, solver=0)
    check_optimize(o, s)
    s.close(); o.close()


def test_optimize_parity_gauss_newton(olib):
    w = synth.make_window("C1", pose_noise_t=0.01, pose_noise_r=0.002, point_noise=0.01)
    o, s, gb = make_pair(olib, w, iterations=10, trust_region=1, solver=2)
    check_optimize(o, s)
    s.close(); o.close()


def test_optimize_parity_no_robust_kernel(olib):
    """RobustKernelDelta <= 0: no Huber, no second phase (Optimizer.cpp:212,283)."""
    o, s, gb = make_pair(olib, synth.make_window("C1", outlier_frac=0.0), iterations=10, robust_kernel_delta=0.0, solver=2)
    st = check_optimize(o, s)
    assert list(st.iterations_run) == [5, 0] and st.n_outliers == 0
    s.close(); o.close()


def test_no_fixed_pose_and_all_landmarks_fixed(olib):
    """rootId outside the window → no fixed pose; gauge held by LM damping and fixed landmarks."""
    w = synth.make_window("C1")
    w["root_id"] = 10 ** 6
    o, s, gb = make_pair(olib, w, iterations=10, solver=2)
    assert gb.pose_fixed.sum() == 0
    check_stages(o, s)
    check_optimize(o, s, pose_tol=1e-5)
    s.close(); o.close()
    w2 = synth.make_window("C1", fixed_frac=1.1)                 # every landmark STABLE → pose-only problem, empty Schur lists
    o, s, gb = make_pair(olib, w2, iterations=10, solver=2)
    assert gb.point_fixed.all()
    check_stages(o, s)
    check_optimize(o, s)
    s.close(); o.close()


def test_two_pose_window(olib):
    w = synth.make_window("custom", n_kf=2, n_lm=40, n_obs=80, seed=4)
    o, s, gb = make_pair(olib, w, iterations=10, solver=2)
    check_stages(o, s)
    check_optimize(o, s)
    s.close(); o.close()


def test_zero_noise_fixed_point_at_c2(olib):
    """Full-size property test: at the true state chi2 ~ 0, no outliers, nothing moves."""
    from visfs_amd import backend
    w = synth.make_window("C2", noise_px=0.0, outlier_frac=0.0, pose_noise_t=0.0, pose_noise_r=0.0, point_noise=0.0)
    prm = abi.default_params(iterations=10, solver=2)
    wb, gb, *_ = graph_of(olib.oracle_pack_window, prm, w)
    s = backend.Solver(prm); s.upload(gb)
    rc, st = s.optimize()
    pose, pt, out, chi = s.download()
    assert rc == abi.OK and st.n_outliers == 0 and st.chi2_initial < 0.05 and st.chi2_final <= st.chi2_initial
    assert np.abs(pose - gb.pose_tq).max() < 1e-6 and np.abs(pt - gb.point_xyz).max() < 1e-4
    s.close()


def test_c4_parity_and_determinism(olib):
    """200 KF / 30k landmarks / 300k observations: parity with the oracle and bitwise run-to-run reproducibility
    (every device reduction has a fixed order; no floating-point atomics)."""
    o, s, gb = make_pair(olib, synth.make_window("C4"), iterations=10, solver=2)
    check_optimize(o, s, pose_tol=1e-5)
    a = s.download()
    s.reset(); s.optimize()
    b = s.download()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    s.close(); o.close()


@pytest.mark.parametrize("solver", [0, 2])
def test_repeated_solves_are_bitwise_identical(olib, solver):
    """Regression: the dense assembly of the direct solver once wrote every entry of a diagonal block twice (from (r,c) and
    (c,r), which can differ in the last bit) — a write race that made `Optimizer/Solver=0` non-repeatable."""
    from visfs_amd import backend
    w = synth.make_window("C1")
    s = backend.Solver(abi.default_params(iterations=10, solver=solver))
    outs = [s.solve_window(abi.WindowBuffers(w))[1].pose_Twr_out.copy() for _ in range(4)]
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    s.close()


# ---------------------------------------------------------------- window layer (localOptimize contract)
def solve_both(olib, w, **prm_kw):
    from visfs_amd import backend
    prm = abi.default_params(**prm_kw)
    wb_o, wb_g = abi.WindowBuffers(w), abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    s = backend.Solver(prm)
    rc_g, rb_g = s.solve_window(wb_g)
    s.close()
    return rc_o, wb_o, rb_o, rc_g, wb_g, rb_g


def test_solve_window_matches_oracle(olib):
    w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    w["point_ids"] = np.r_[np.asarray(w["point_ids"]), np.uint64(77777)]       # a point without references → NaN on return
    w["point_xyz"] = np.vstack([w["point_xyz"], [[1.0, 2.0, 3.0]]]); w["point_fixed"] = np.r_[w["point_fixed"], np.uint8(0)]
    rc_o, wb_o, rb_o, rc_g, wb_g, rb_g = solve_both(olib, w, iterations=20, solver=2)
    assert rc_o == rc_g == abi.OK
    assert rb_g.struct.n_poses_out == rb_o.struct.n_poses_out == 12
    assert np.array_equal(rb_g.pose_ids_out[:12], rb_o.pose_ids_out[:12])
    et, er = synth.pose_errors(rb_g.pose_Twr_out[:12], rb_o.pose_Twr_out[:12])
    assert et < 1e-6 and er < 1e-6                                             # north_star: <= 1e-4
    assert rb_g.outliers() == rb_o.outliers()                                  # same pairs, same (reference) order
    assert np.isnan(wb_g.point_xyz[-1]).all() and np.isnan(wb_o.point_xyz[-1]).all()
    assert rel_err(wb_g.point_xyz[:-1], wb_o.point_xyz[:-1]) < 1e-6
    assert list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)


def test_solve_window_error_convention(olib):
    from visfs_amd import backend
    w = synth.make_window("C1")
    s = backend.Solver(abi.default_params(iterations=10))
    # single pose → passthrough (Optimizer.cpp:360-361)
    w1 = dict(w); w1["pose_ids"] = w["pose_ids"][:1]; w1["pose_Twr"] = w["pose_Twr"][:1]
    rc, rb = s.solve_window(abi.WindowBuffers(w1))
    assert rc == abi.PASSTHROUGH and rb.struct.n_poses_out == 1 and np.array_equal(rb.pose_Twr_out[0], np.asarray(w["pose_Twr"])[0])
    # first id == 0 → error, empty map (Optimizer.cpp:74, 362-364)
    w0 = dict(w); w0["pose_ids"] = np.arange(0, 10, dtype=np.uint64)
    rc, rb = s.solve_window(abi.WindowBuffers(w0))
    assert rc == abi.ERR_TOO_FEW_POSES and rb.struct.n_poses_out == 0
    # NaN pose → "Optimization generated NANs" → empty map (Optimizer.cpp:272-275)
    wn = dict(w); T = np.asarray(w["pose_Twr"]).copy(); T[3, 3] = np.nan; wn["pose_Twr"] = T
    rc, rb = s.solve_window(abi.WindowBuffers(wn))
    assert rc == abi.ERR_NAN_CHI2 and rb.struct.n_poses_out == 0 and rb.struct.n_outliers == 0
    # laser points without a sub-map: the reference builds no factor (Optimizer.cpp:225) → same result as without them
    wl = dict(w); wl["laser_xyz"] = np.ones((5, 3)); wl["n_laser_points"] = 5
    rc, rb = s.solve_window(abi.WindowBuffers(wl))
    rc_ref, rb_ref = s.solve_window(abi.WindowBuffers(w))
    assert rc == rc_ref == abi.OK and np.array_equal(rb.pose_Twr_out, rb_ref.pose_Twr_out)
    # iterations <= 0 → passthrough
    s0 = backend.Solver(abi.default_params(iterations=0))
    rc, rb = s0.solve_window(abi.WindowBuffers(w))
    assert rc == abi.PASSTHROUGH and rb.struct.n_poses_out == 10
    s.close(); s0.close()


# ---------------------------------------------------------------- size-independent properties at the full BASELINE size
def _rigid(yaw, pitch, t):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    return R, np.asarray(t, float)


def test_c2_gauge_covariance():
    """Rotating the whole world about its origin (poses G*Twr, landmarks G*p, measurements and odometry untouched) must
    rotate the solution by G.  No oracle involved — a property of the path itself at the BASELINE size (50 KF / 5k
    landmarks / 50k observations + 49 odometry edges).  NOT true for a translation of the world, and that is the
    reference's doing: its pose Jacobian (OptimizeTypeDefine.h:157-176) is the SE(3)-left form in Pc while the update
    leaves t unrotated, an approximation whose error grows with |t_cw| — shifting C1 by 12 m changes the outlier count
    from 92 to 2206 in the CPU restatement as well (quirk reproduced)."""
    from visfs_amd import backend
    w = synth.make_window("C3")
    R, t = _rigid(0.7, -0.3, [0.0, 0.0, 0.0])
    w2 = dict(w)
    T = np.asarray(w["pose_Twr"]).reshape(-1, 3, 4)
    w2["pose_Twr"] = np.concatenate([R @ T[:, :, :3], (R @ T[:, :, 3:]) + t[:, None]], axis=2).reshape(-1, 12)
    w2["point_xyz"] = np.asarray(w["point_xyz"]) @ R.T + t
    s = backend.Solver(abi.default_params(iterations=20, solver=2))
    wa, wb = abi.WindowBuffers(w), abi.WindowBuffers(w2)
    rca, ra = s.solve_window(wa)
    rcb, rb = s.solve_window(wb)
    s.close()
    assert rca == rcb == abi.OK and ra.outliers() == rb.outliers()
    Ta = ra.pose_Twr_out.reshape(-1, 3, 4)
    expect = np.concatenate([R @ Ta[:, :, :3], (R @ Ta[:, :, 3:]) + t[:, None]], axis=2).reshape(-1, 12)
    et, er = synth.pose_errors(rb.pose_Twr_out, expect)
    assert et < 1e-8 and er < 1e-8
    ok = ~np.isnan(wa.point_xyz).any(axis=1)
    assert rel_err(wb.point_xyz[ok], (wa.point_xyz[ok] @ R.T + t)) < 1e-7


def test_c2_converged_solution_is_a_fixed_point():
    """Feeding a solution back in (same measurements, culled references removed) leaves it where it is: the second solve starts
    at the minimum of the same robust objective, so no pose moves by more than the first solve's own convergence tolerance."""
    from visfs_amd import backend
    w = synth.make_window("C2")
    s = backend.Solver(abi.default_params(iterations=40, solver=2))
    wa = abi.WindowBuffers(w)
    rc, ra = s.solve_window(wa)
    assert rc == abi.OK
    out = set(ra.outliers())
    keep = np.array([(int(f), int(p)) not in out for f, p in zip(w["ref_feature"], w["ref_pose"])])
    w2 = drop_refs(w, keep)
    w2["pose_Twr"] = ra.pose_Twr_out[:len(w["pose_ids"])].copy()
    pts = wa.point_xyz.copy(); bad = np.isnan(pts).any(axis=1); pts[bad] = np.asarray(w["point_xyz"])[bad]
    w2["point_xyz"] = pts
    wb = abi.WindowBuffers(w2)
    rc2, rb = s.solve_window(wb)
    s.close()
    assert rc2 == abi.OK
    et, er = synth.pose_errors(rb.pose_Twr_out, ra.pose_Twr_out)
    assert et < 1e-5 and er < 1e-5
    assert rb.struct.chi2_final <= rb.struct.chi2_initial * (1 + 1e-9)


# ---------------------------------------------------------------- small windows: k_small_solve / fused single-workgroup kernel
def _solve_in_mode(monkeypatch, w, env, **prm_kw):
    from visfs_amd import backend
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    prm = abi.default_params(**prm_kw)
    s = backend.Solver(prm)
    gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s.upload(gb)
    info = s.describe()
    rc, st = s.optimize()
    out = s.download()
    s.close()
    return info, rc, st, out


@pytest.mark.parametrize("solver", [0, 2])
@pytest.mark.parametrize("cfg", ["PROD", "C1", "LASER"])
def test_small_window_paths_agree(olib, monkeypatch, cfg, solver):
    """<= 10 free poses: the reduced system is finalised and solved by one workgroup (k_small_solve, default) instead of
    k_schur_finalize + k_pcg / the blocked Cholesky; VISFS_BA_FUSED=1 runs the whole optimisation in one launch of one
    workgroup.  All three orders of summation must give the same LM trajectory, outliers and poses (to rounding)."""
    w = synth.make_laser_window(with_visual=True, n_points=300) if cfg == "LASER" else synth.make_window(cfg)
    kw = dict(iterations=10, solver=solver)
    gi, grc, gst, gout = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SMALL_SOLVE="0", VISFS_BA_FUSED="0"), **kw)
    assert gi["fused_path"] == 0 and grc == abi.OK
    for env in (dict(VISFS_BA_SMALL_SOLVE="1", VISFS_BA_FUSED="0"), dict(VISFS_BA_SMALL_SOLVE="1", VISFS_BA_FUSED="1")):
        info, rc, st, out = _solve_in_mode(monkeypatch, w, env, **kw)
        assert info["fused_path"] == int(env["VISFS_BA_FUSED"]) and rc == abi.OK
        assert list(st.iterations_run) == list(gst.iterations_run) and list(st.trials_run) == list(gst.trials_run)
        assert st.pcg_iterations == gst.pcg_iterations and st.n_outliers == gst.n_outliers
        assert abs(st.chi2_final - gst.chi2_final) <= 1e-10 * gst.chi2_final
        assert np.abs(out[0] - gout[0]).max() < 1e-12 and rel_err(out[1], gout[1]) < 1e-11
        assert np.array_equal(out[2], gout[2])


def test_fused_path_matches_oracle_with_rejected_steps(olib, monkeypatch):
    # a hard start on a 9-free-pose window: rejected trials, lambda growth — through the fused kernel's own LM loop
    monkeypatch.setenv("VISFS_BA_FUSED", "1")
    w = synth.make_window("custom", n_kf=10, n_lm=300, n_obs=2400, seed=5, point_noise=1.0, fixed_frac=0.0)
    o, s, gb = make_pair(olib, w, iterations=20, solver=0)
    assert s.describe()["fused_path"] == 1
    check_optimize(o, s, pose_tol=1e-6)
    s.close(); o.close()


# ---------------------------------------------------------------- laser occupied-space factor
@pytest.mark.parametrize("visual,solver", [(False, 2), (False, 0), (True, 2)])
def test_laser_factor_stages_and_optimize(olib, visual, solver):
    """EdgeOccupiedObservation (TypeOccupiedSpace2D.h:75-185) on the newest pose: sensor strategy 4/5 windows have no
    landmarks at all (Estimator.cpp:243-250); the mixed case keeps the stereo part."""
    w = synth.make_laser_window(with_visual=visual, n_points=1000)
    o, s, gb = make_pair(olib, w, iterations=10, solver=solver)
    assert gb.struct.n_laser == 1000 and gb.struct.laser_pose == 5
    check_stages(o, s)
    check_optimize(o, s, pose_tol=1e-7)
    s.close(); o.close()


def test_laser_factor_window_level(olib):
    w = synth.make_laser_window(with_visual=True, n_points=500, seed=3)
    rc_o, wb_o, rb_o, rc_g, wb_g, rb_g = solve_both(olib, w, iterations=10, solver=2)
    assert rc_o == rc_g == abi.OK and rb_g.struct.n_poses_out == rb_o.struct.n_poses_out == 6
    et, er = synth.pose_errors(rb_g.pose_Twr_out[:6], rb_o.pose_Twr_out[:6])
    assert et < 1e-7 and er < 1e-7
    assert rb_g.outliers() == rb_o.outliers()
    assert abs(rb_g.struct.chi2_final - rb_o.struct.chi2_final) <= 1e-7 * rb_o.struct.chi2_final
    # the factor is really in the objective: without it the final chi2 is different
    w2 = dict(w); w2["grid"] = None
    _, _, rb_o2, _, _, rb_g2 = solve_both(olib, w2, iterations=10, solver=2)
    assert abs(rb_g2.struct.chi2_final - rb_g.struct.chi2_final) > 1.0
    assert abs(rb_g2.struct.chi2_final - rb_o2.struct.chi2_final) <= 1e-7 * rb_o2.struct.chi2_final


def test_laser_edges_are_inactive_when_their_pose_is_fixed(olib):
    # allVerticesFixed (the range points are fixed vertices): the edges leave the active set, chi2 included
    w = synth.make_laser_window(with_visual=True, n_points=300, seed=1)
    w["root_id"] = int(w["pose_ids"][-1])
    _, _, rb_o, _, _, rb_g = solve_both(olib, w, iterations=10, solver=2)
    w2 = dict(w); w2["grid"] = None
    _, _, rb_o2, _, _, rb_g2 = solve_both(olib, w2, iterations=10, solver=2)
    assert rb_g.struct.chi2_final == rb_g2.struct.chi2_final and np.array_equal(rb_g.pose_Twr_out, rb_g2.pose_Twr_out)
    assert rb_o.struct.chi2_final == rb_o2.struct.chi2_final


def test_solve_batch_equals_individual_solves(olib):
    """BASELINE config 5 in miniature: independent windows solved concurrently give the single-window results bit for bit."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("C1", window_index=i) for i in range(5)]
    s = backend.Solver(prm)
    singles = []
    for w in ws:
        rc, rb = s.solve_window(abi.WindowBuffers(w))
        assert rc == abi.OK
        singles.append(rb)
    wbs = [abi.WindowBuffers(w) for w in ws]
    rbs = s.solve_batch(wbs)
    for a, b in zip(singles, rbs):
        assert b.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
    s.close()


def test_batched_launches_match_single_window_solves(olib):
    """independent windows as ONE sequence of launches (blockIdx.y = window, each window gated by its own LM
    state).  Mixed shapes: two launch-geometry groups, a window that needs more damped solves than the others, one that is
    refused — every result must equal the single-window solve bit for bit."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("C1", window_index=i) for i in range(3)]
    ws += [synth.make_window("PROD", window_index=i) for i in range(3)]
    ws.append(hard_window())                                                   # rejected trials: runs longer than its neighbours
    ws.append(synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11))   # > 10 free poses: general PCG path
    bad = dict(ws[0]); bad["pose_ids"] = np.arange(0, 10, dtype=np.uint64)     # first id 0: refused (Optimizer.cpp:74)
    ws.append(bad)
    s = backend.Solver(prm)
    singles = []
    for w in ws:
        wb = abi.WindowBuffers(w)
        rc, rb = s.solve_window(wb)
        singles.append((rc, rb, wb))
    wbs = [abi.WindowBuffers(w) for w in ws]
    rbs = s.solve_batch(wbs)
    for (rc, a, wa), b, wb in zip(singles, rbs, wbs):
        assert b.struct.status == rc
        assert b.struct.n_poses_out == a.struct.n_poses_out
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert np.array_equal(wa.point_xyz, wb.point_xyz, equal_nan=True)
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run) and a.struct.chi2_final == b.struct.chi2_final
    s.close()


def test_batch_graph_layer_resident_windows(olib):
    # the bench's config-5 path: graphs resident side by side, reset + one batched optimise per step
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    gbs, refs = [], []
    for i in range(4):
        w = synth.make_window("C1", window_index=i)
        wb, gb, *_ = graph_of(olib.oracle_pack_window, prm, w)
        gbs.append(gb)
        s1 = backend.Solver(prm); s1.upload(gb); s1.optimize(); refs.append(s1.download()); s1.close()
    s = backend.Solver(prm)
    s.batch_upload(gbs)
    for _ in range(2):                                     # twice: reset restores the uploaded estimates
        s.batch_reset()
        rc, stats = s.batch_optimize()
        assert rc == abi.OK and all(st.status == abi.OK for st in stats)
        for i in range(4):
            got = s.batch_download(i)
            assert all(np.array_equal(a, b) for a, b in zip(got, refs[i]))
    s.close()


def test_fused_kernel_batch(olib, monkeypatch):
    # many small windows, one CU each: the fused kernel as a batched launch
    from visfs_amd import backend
    monkeypatch.setenv("VISFS_BA_FUSED", "1")
    prm = abi.default_params(iterations=10, solver=0)
    ws = [synth.make_window("PROD", window_index=i) for i in range(24)]
    s = backend.Solver(prm)
    rbs = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    for w, b in zip(ws, rbs):
        rc, a = s.solve_window(abi.WindowBuffers(w))
        assert rc == b.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
    s.close()


def test_handle_reuse_across_shapes(olib):
    """One handle, windows of different sizes back to back (the per-frame usage pattern of Estimator::process)."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    s = backend.Solver(prm)
    for cfg in ("PROD", "C1", "PROD"):
        w = synth.make_window(cfg)
        rc, rb = s.solve_window(abi.WindowBuffers(w))
        wb_o = abi.WindowBuffers(w); rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
        assert olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1) == abi.OK == rc
        n = rb.struct.n_poses_out
        et, er = synth.pose_errors(rb.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
        assert et < 1e-6 and er < 1e-6 and rb.outliers() == rb_o.outliers()
    s.close()


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


# ---------------------------------------------------------------- the speculative unit, chunk passes, the WIDE PCG kernel
def _stats_tuple(st):
    return (list(st.iterations_run), list(st.trials_run), st.pcg_iterations, st.n_outliers, st.chi2_initial, st.chi2_phase1, st.chi2_final,
            [st.trace_lambda[i] for i in range(st.n_trace)], [st.trace_chi2[i] for i in range(st.n_trace)])


@pytest.mark.parametrize("case", ["C1", "PROD", "LASER", "HARD", "GN"])
def test_speculative_unit_equals_gated_unit(olib, monkeypatch, case):
    """the unit that linearises the trial state beside the LM decision (and, with odometry / laser edges, inside
    k_backsub) performs the arithmetic of the gated unit — every output, counter and trace entry must be bit-identical,
    rejected trials (HARD) and Gauss-Newton included."""
    kw = dict(iterations=20, solver=2)
    if case == "LASER":
        w = synth.make_laser_window(with_visual=True, n_points=400)
    elif case == "HARD":
        w = hard_window()
    elif case == "GN":
        w = synth.make_window("C1"); kw["trust_region"] = 1
    else:
        w = synth.make_window(case)
    _, rc0, st0, out0 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="0"), **kw)
    _, rc1, st1, out1 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="1"), **kw)
    assert rc0 == rc1 == abi.OK
    assert _stats_tuple(st0) == _stats_tuple(st1)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out0, out1))
    if case == "HARD":
        assert sum(st0.trials_run) > sum(st0.iterations_run)          # the window really rejects trials


@pytest.mark.parametrize("case", ["C1", "PROD", "LASER", "HARD", "GN", "C3", "C2", "RAGGED", "S0"])
def test_fused_speculative_unit_equals_the_two_launch_form_and_the_gated_unit(olib, monkeypatch, case):
    """Round 3: the speculative unit's tail in ONE launch (k_backsub<LINA>: back-substitution, LM decision, role A of the trial's
    linearisation; its pose-major role rides behind the next Schur gather) against the two-launch form it replaces and the gated
    unit: every output, counter and trace entry bit-identical — rejected trials (HARD), Gauss-Newton, odometry (C3), laser, the direct
    solver (S0) and ragged tracks included."""
    kw = dict(iterations=20, solver=2)
    if case == "LASER":
        w = synth.make_laser_window(with_visual=True, n_points=400)
    elif case == "HARD":
        w = hard_window()
    elif case == "RAGGED":
        w = ragged_window(seed=7)
    elif case == "GN":
        w = synth.make_window("C1"); kw["trust_region"] = 1
    elif case == "S0":
        w = synth.make_window("C1"); kw["solver"] = 0
    elif case == "C3":
        w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    else:
        w = synth.make_window(case)
    _, rc0, st0, out0 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="0"), **kw)
    _, rc1, st1, out1 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="1", VISFS_BA_SPEC_FUSED="0"), **kw)
    _, rc2, st2, out2 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="1", VISFS_BA_SPEC_FUSED="1"), **kw)
    assert rc0 == rc1 == rc2 == abi.OK
    assert _stats_tuple(st0) == _stats_tuple(st1) == _stats_tuple(st2)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out0, out2))
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out1, out2))
    if case == "HARD":
        assert sum(st0.trials_run) > sum(st0.iterations_run)


@pytest.mark.parametrize("case", ["C1", "LASER", "HARD", "GN", "C3"])
def test_decision_on_board_backsub_equals_separate_decide_launch(olib, monkeypatch, case):
    """The gated unit (batched windows, large windows, VISFS_BA_SPEC=0): one workgroup of k_backsub waits for the partial sums of all
    the others (hand-off words, the data is the flag) and takes the LM decision; round 1 launched k_decide for it.  Same sums in
    the same order: every output, counter and trace entry must be bit-identical, rejected trials and failed solves included."""
    kw = dict(iterations=20, solver=2)
    if case == "LASER":
        w = synth.make_laser_window(with_visual=True, n_points=400)
    elif case == "HARD":
        w = hard_window()
    elif case == "GN":
        w = synth.make_window("C1"); kw["trust_region"] = 1
    else:
        w = synth.make_window(case)
    _, rc0, st0, out0 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="0", VISFS_BA_DECIDE_FUSED="0"), **kw)
    _, rc1, st1, out1 = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SPEC="0", VISFS_BA_DECIDE_FUSED="1"), **kw)
    assert rc0 == rc1 == abi.OK
    assert _stats_tuple(st0) == _stats_tuple(st1)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out0, out1))
    # twice on the same resident graph: the launch tags keep counting across solves (stale hand-off words never match)
    from visfs_amd import backend
    prm = abi.default_params(**kw)
    s = backend.Solver(prm)
    gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s.upload(gb)
    for _ in range(3):
        s.reset(); rc, st = s.optimize()
        assert rc == abi.OK and _stats_tuple(st) == _stats_tuple(st0)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(s.download(), out0))
    s.close()


def test_schur_chunk_passes_agree_to_rounding(olib, monkeypatch):
    # 64 / 128 / 192 pairs per chunk: same sums in a different association
    w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11)
    ref = None
    for passes in ("1", "2", "3"):
        info, rc, st, out = _solve_in_mode(monkeypatch, w, dict(VISFS_BA_SCH_PASSES=passes, VISFS_BA_SCHUR_RUNS="0"), iterations=10, solver=2)   # (the pair-list gather)
        assert rc == abi.OK
        if ref is None:
            ref = (info, st, out)
            continue
        assert info["n_schur_chunks"] < ref[0]["n_schur_chunks"]
        assert list(st.iterations_run) == list(ref[1].iterations_run)
        assert rel_err(out[0], ref[2][0]) < 1e-11 and rel_err(out[1], ref[2][1]) < 1e-11
        assert abs(st.chi2_final - ref[1].chi2_final) <= 1e-11 * ref[1].chi2_final


def test_wide_pcg_kernel_matches_oracle(olib):
    """More than 64 free poses: the vector recurrences of the persistent PCG run on all four waves (k_pcg WIDE)."""
    w = synth.make_window("custom", n_kf=100, n_lm=3000, n_obs=24000, seed=21)
    o, s, gb = make_pair(olib, w, iterations=10, solver=2)
    assert s.describe()["n_free_poses"] > 64
    check_optimize(o, s, pose_tol=1e-9)
    s.close(); o.close()
