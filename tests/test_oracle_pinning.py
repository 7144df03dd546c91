"""Pins the CPU oracle (the checker of every GPU parity test).

The reference holds no test or golden vector for Optimizer::localOptimize,
so the oracle is pinned by: hand-derived known-answer vectors of the cited formulas,
finite differences under the reference's own oplus, fixed points, solver cross-checks and
the LM invariants of the g2o algorithm.  All CPU, a few seconds.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from helpers import graph_of, hard_window, ragged_window, rel_err, twr_of
from visfs_amd import abi, synth

_pd = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(_pd)


def stereo(olib, tq, pw, uvr, intr, jac=True):
    tq = np.ascontiguousarray(tq, np.float64); pw = np.ascontiguousarray(pw, np.float64)
    uvr = np.ascontiguousarray(uvr, np.float64); intr = np.ascontiguousarray(intr, np.float64)
    e = np.zeros(3); Jp = np.zeros(9); Jx = np.zeros(18)
    olib.oracle_stereo_edge(P(tq), P(pw), P(uvr), P(intr), P(e), P(Jp) if jac else None, P(Jx) if jac else None)
    return e, Jp.reshape(3, 3), Jx.reshape(3, 6)


def oplus(olib, tq, d):
    out = np.ascontiguousarray(tq, np.float64).copy()
    d = np.ascontiguousarray(d, np.float64)
    olib.oracle_pose_update(P(out), P(d))
    return out


def odo(olib, tq1, tq2, m):
    e = np.zeros(6); Ji = np.zeros(36); Jj = np.zeros(36)
    a, b, c = (np.ascontiguousarray(x, np.float64) for x in (tq1, tq2, m))
    olib.oracle_odo_edge(P(a), P(b), P(c), P(e), P(Ji), P(Jj))
    return e, Ji.reshape(6, 6), Jj.reshape(6, 6)


def rand_pose(rng, t_scale=1.0, r_scale=0.5):
    q = np.r_[r_scale * rng.normal(size=3), 1.0]
    q /= np.linalg.norm(q)
    return np.r_[t_scale * rng.normal(size=3), q]


# ---------------------------------------------------------------- known-answer vectors
def test_stereo_edge_known_answer(olib):
    """identity pose, Pw=(0.5,-0.2,4), fx=fy=400, cx=320, cy=240, bf=48 → pi=(370,220,358);
    Jacobian rows follow OptimizeTypeDefine.h:145-176 literally (hand-evaluated)."""
    tq = [0, 0, 0, 0, 0, 0, 1]
    e, Jp, Jx = stereo(olib, tq, [0.5, -0.2, 4.0], [371.0, 219.0, 360.0], [400, 400, 320, 240, 48])
    assert np.allclose(e, [1.0, -1.0, 2.0], atol=1e-12)
    assert np.allclose(Jp, [[-100, 0, 12.5], [0, -100, -5.0], [-100, 0, 9.5]], atol=1e-12)
    assert np.allclose(Jx, [[-100, 0, 12.5, -2.5, -406.25, -20.0],
                            [0, -100, -5.0, 401.0, 2.5, -50.0],
                            [-100, 0, 9.5, -1.9, -404.75, -20.0]], atol=1e-12)


def test_pose_update_known_answer(olib):
    """CameraPose::update: t += dt; q = normalize((1, dtheta/2) * q) (OptimizeTypeDefine.cpp:7-14, Math.h:277-287)."""
    out = oplus(olib, [1, 2, 3, 0, 0, 0, 1], [0.1, 0.2, 0.3, 0.02, 0.0, 0.0])
    n = np.sqrt(1 + 0.01 ** 2)
    assert np.allclose(out, [1.1, 2.2, 3.3, 0.01 / n, 0, 0, 1 / n], atol=1e-15)
    # left multiplication: dq * q with q = 90 deg about z
    s = np.sqrt(0.5)
    out = oplus(olib, [0, 0, 0, 0, 0, s, s], [0, 0, 0, 0.2, 0, 0])
    expect = np.array([0.1 * s, 0.1 * s, s, s])        # (0.1,0,0,1)*(0,0,s,s) = (0.1s, 0.1s... Hamilton product
    expect = np.array([1 * 0 + 0.1 * s + 0 * s - 0 * 0, 1 * 0 + 0 * s + 0 * 0 - 0.1 * s, 1 * s + 0 * s + 0.1 * 0 - 0 * 0, 1 * s - 0.1 * 0 - 0 - 0])
    expect /= np.linalg.norm(expect)
    assert np.allclose(out[3:], expect, atol=1e-15)


def test_huber_known_answer(olib):
    """[g2o-upstream] RobustKernelHuber on chi2 with delta SQUARED as the threshold."""
    rho = np.zeros(2)
    olib.oracle_huber(63.9, 8.0, P(rho)); assert np.allclose(rho, [63.9, 1.0])
    olib.oracle_huber(100.0, 8.0, P(rho)); assert np.allclose(rho, [2 * 10 * 8 - 64, 0.8])


def test_pose_from_Rt_matches_eigen_conventions(olib):
    """CameraPose(R,t): quaternion has w >= 0 and unit norm; image→robot rotation of GeometricCamera.h:15-19."""
    R = np.array([[0., 0, 1], [-1, 0, 0], [0, -1, 0]])
    tq = np.zeros(7)
    olib.oracle_pose_from_Rt(P(np.ascontiguousarray(R.reshape(9))), P(np.zeros(3)), P(tq))
    assert np.allclose(tq[3:], [-0.5, 0.5, -0.5, 0.5], atol=1e-15)      # (x,y,z,w)
    R2 = np.zeros(9); t2 = np.zeros(3)
    olib.oracle_pose_to_Rt(P(tq), P(R2), P(t2))
    assert np.allclose(R2.reshape(3, 3), R, atol=1e-15)
    # trace <= 0 branches
    for Rm in (np.diag([1., -1, -1]), np.diag([-1., 1, -1]), np.diag([-1., -1, 1])):
        olib.oracle_pose_from_Rt(P(np.ascontiguousarray(Rm.reshape(9))), P(np.zeros(3)), P(tq))
        olib.oracle_pose_to_Rt(P(tq), P(R2), P(t2))
        assert np.allclose(R2.reshape(3, 3), Rm, atol=1e-15) and tq[6] >= 0 and abs(np.linalg.norm(tq[3:]) - 1) < 1e-15


# ---------------------------------------------------------------- finite differences
def test_stereo_jacobians_finite_differences(olib):
    rng = np.random.default_rng(1)
    intr = [435.2, 435.2, 367.2, 252.2, 0.11 * 435.2]
    for _ in range(20):
        tq = rand_pose(rng)
        pc = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 8)])
        R = np.zeros(9); t = np.zeros(3)
        olib.oracle_pose_to_Rt(P(np.ascontiguousarray(tq)), P(R), P(t))
        R = R.reshape(3, 3)
        pw = R.T @ (pc - t)
        uvr = rng.uniform(0, 400, 3)
        e0, Jp, Jx = stereo(olib, tq, pw, uvr, intr)
        h = 1e-6
        # point block: exact derivative
        for c in range(3):
            d = np.zeros(3); d[c] = h
            num = (stereo(olib, tq, pw + d, uvr, intr, False)[0] - stereo(olib, tq, pw - d, uvr, intr, False)[0]) / (2 * h)
            assert np.allclose(num, Jp[:, c], rtol=1e-6, atol=1e-5)
        # translation block: exact under the reference oplus (t += dt)
        for c in range(3):
            d = np.zeros(6); d[c] = h
            num = (stereo(olib, oplus(olib, tq, d), pw, uvr, intr, False)[0] - stereo(olib, oplus(olib, tq, -d), pw, uvr, intr, False)[0]) / (2 * h)
            assert np.allclose(num, Jx[:, c], rtol=1e-6, atol=1e-5)
        # rotation block: the reference writes the SE(3) left-perturbation form -dpi/dPc [Pc]x while its
        # oplus leaves t unrotated, for which the exact block uses R Pw = Pc - t.  The oracle reproduces the formula as
        # written; the gap to the true derivative is exactly  -dpi/dPc [t]x.
        x, y, z = pc
        dpi = np.array([[intr[0] / z, 0, -intr[0] * x / z ** 2], [0, intr[1] / z, -intr[1] * y / z ** 2],
                        [intr[0] / z, 0, -intr[0] * x / z ** 2 + intr[4] / z ** 2]])
        def sk(v):
            return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        assert np.allclose(Jx[:, 3:], -dpi @ (-sk(pc)), rtol=1e-9, atol=1e-9)
        num = np.zeros((3, 3))
        for c in range(3):
            d = np.zeros(6); d[3 + c] = h
            num[:, c] = (stereo(olib, oplus(olib, tq, d), pw, uvr, intr, False)[0] - stereo(olib, oplus(olib, tq, -d), pw, uvr, intr, False)[0]) / (2 * h)
        assert np.allclose(num, -dpi @ (-sk(pc - t)), rtol=1e-5, atol=1e-4)          # true derivative under the reference oplus
        assert np.allclose(Jx[:, 3:] - num, -dpi @ (-sk(t)), rtol=1e-5, atol=1e-4)   # the documented gap


def test_odometry_edge_zero_residual_and_jacobians(olib):
    """EdgePoseConstraint (OptimizeTypeDefine.cpp:35-88): zero error for a consistent measurement; the "Left update"
    Jacobians equal central differences under the reference oplus at (near-)zero residual."""
    rng = np.random.default_rng(2)
    for _ in range(10):
        tq1, tq2 = rand_pose(rng), rand_pose(rng)
        # measurement T_c1c2 = T1 * T2^-1 expressed as the edge expects: mP = Q1 Q2^-1 (-P2) + P1, mQ = Q1 Q2^-1
        e, _, _ = odo(olib, tq1, tq2, [0, 0, 0, 0, 0, 0, 1])
        mP = e[:3]
        def qmul(a, b):
            ax, ay, az, aw = a; bx, by, bz, bw = b
            return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                             aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])
        q2i = np.r_[-tq2[3:6], tq2[6]]
        mQ = qmul(tq1[3:], q2i)
        if mQ[3] < 0:
            mQ = -mQ
        m = np.r_[mP, mQ]
        e0, Ji, Jj = odo(olib, tq1, tq2, m)
        assert np.abs(e0).max() < 1e-12
        h = 1e-6
        for c in range(6):
            d = np.zeros(6); d[c] = h
            ni = (odo(olib, oplus(olib, tq1, d), tq2, m)[0] - odo(olib, oplus(olib, tq1, -d), tq2, m)[0]) / (2 * h)
            nj = (odo(olib, tq1, oplus(olib, tq2, d), m)[0] - odo(olib, tq1, oplus(olib, tq2, -d), m)[0]) / (2 * h)
            assert np.allclose(ni, Ji[:, c], atol=2e-6), (c, ni, Ji[:, c])
            assert np.allclose(nj, Jj[:, c], atol=2e-6), (c, nj, Jj[:, c])


# ---------------------------------------------------------------- graph build
# ---------------------------------------------------------------- laser occupied-space factor
PAD = 2147483647 // 4


def _bicubic(olib, gb, r, c):
    f, dr, dc = C.c_double(), C.c_double(), C.c_double()
    olib.oracle_bicubic(C.byref(gb.struct), C.c_double(r), C.c_double(c), C.byref(f), C.byref(dr), C.byref(dc))
    return f.value, dr.value, dc.value


def _laser_coords(tq, w, Tcr, P, g, pad=PAD):
    """numpy restatement of the functor's geometry (TypeOccupiedSpace2D.h:97-118) for quaternion (x, y, z, w) used raw."""
    x, y, z = tq[3:6]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    Tcr = np.asarray(Tcr).reshape(3, 4)
    Po = R.T @ (Tcr[:, :3] @ P + Tcr[:, 3] - tq[:3])
    return (g["max_x"] - Po[0]) / g["resolution"] - 0.5 + pad, (g["max_y"] - Po[1]) / g["resolution"] - 0.5 + pad


def _laser_setup(olib):
    olib.oracle_bicubic.argtypes = [C.POINTER(abi.Grid), C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 3
    olib.oracle_bicubic.restype = None
    olib.oracle_laser_edge.argtypes = [C.POINTER(C.c_double)] * 3 + [C.POINTER(abi.Grid)] + [C.POINTER(C.c_double)] * 2
    olib.oracle_laser_edge.restype = None


def test_bicubic_interpolator_properties(olib):
    """[ceres-upstream] BiCubicInterpolator over GridArrayAdapter: exact at cell centres, C1 across cell borders, the
    analytic derivatives match central differences, 0.9 outside the padded grid (TypeOccupiedSpace2D.h:28-37)."""
    _laser_setup(olib)
    g = synth.make_grid(); gb = abi.GridBuffers(g)
    for row, col in ((14, 70), (100, 120), (0, 0), (199, 239)):
        assert _bicubic(olib, gb, PAD + float(row), PAD + float(col))[0] == float(g["cost"][row, col])
    assert _bicubic(olib, gb, PAD - 10.0, PAD + 5.0) == (0.9, 0.0, 0.0)
    r, c, h = PAD + 14.3, PAD + 70.6, 2.0 ** -12       # coordinates ~5e8: steps must be exactly representable
    f, dr, dc = _bicubic(olib, gb, r, c)
    assert abs(dr - (_bicubic(olib, gb, r + h, c)[0] - _bicubic(olib, gb, r - h, c)[0]) / (2 * h)) < 1e-5
    assert abs(dc - (_bicubic(olib, gb, r, c + h)[0] - _bicubic(olib, gb, r, c - h)[0]) / (2 * h)) < 1e-5
    e = 2.0 ** -20
    lo, hi = _bicubic(olib, gb, PAD + 15.0 - e, c), _bicubic(olib, gb, PAD + 15.0 + e, c)
    assert abs(lo[0] - hi[0]) < 1e-6 and abs(lo[1] - hi[1]) < 1e-4 and abs(lo[2] - hi[2]) < 1e-4


def test_laser_grid_addressing_matches_reference_unit_tests(olib):
    """PINNED against the reference's own test vectors (tests/golden/ref_map2d_cell_index.json, transcribed from
    tests/Map/2d/UT4ProbabilityGrid/UT4ProbabilityGrid.cpp:58-103): the cell the occupied-space functor samples for a world
    point (axis swap, flip about `max`, half-cell offset, row-major flat index: TypeOccupiedSpace2D.h:28-37,115-118 over
    MapLimits.h:43-64 / Grid2d.h:93-95) is the cell MapLimits::getCellIndex names for that point.  Identity pose and
    identity robot->camera transform make the functor's world point the range point itself; a grid that is 0 except for ONE
    hot cell then answers > 0.5 exactly when the hot cell is the nearest one (Catmull-Rom weight of the nearest sample at an
    offset <= 0.25 cell is >= 0.86 per axis), and exactly 1 at a cell centre."""
    import json, os
    _laser_setup(olib)
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_map2d_cell_index.json")))
    tq = np.array([0, 0, 0, 0, 0, 0, 1.0]); Tcr = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.0])
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))

    def sample(gb, pt):
        P = np.array([pt[0], pt[1], 0.3]); e = C.c_double()
        olib.oracle_laser_edge(p(tq), p(Tcr), p(P), C.byref(gb.struct), C.byref(e), None)
        return e.value

    checked = 0
    for case in fx["cases"]:
        nx, ny = case["num_x_cells"], case["num_y_cells"]
        base = dict(resolution=case["resolution"], max_x=case["max"][0], max_y=case["max"][1])
        hit = set()
        for v in case["points"]:
            centre = all(((m - q) / case["resolution"] - 0.5) % 1.0 == 0.0 for m, q in zip(case["max"], v["point"]))
            found = []
            for y in range(ny):
                for x in range(nx):
                    cost = np.zeros((ny, nx), np.float32); cost[y, x] = 1.0
                    e = sample(abi.GridBuffers(dict(base, cost=cost)), v["point"])
                    if e > 0.5:
                        found.append(([x, y], e))
            assert len(found) == 1, (v, found)                      # inside the limits, one nearest cell
            if v["cell"] is not None:
                assert found[0][0] == v["cell"], (v, found)
            if centre:
                assert found[0][1] == 1.0
            hit.add(tuple(found[0][0])); checked += 1
        if all(v["cell"] is None for v in case["points"]):
            assert len(hit) == len(case["points"])                  # four points, four distinct cells, all contained
        # outside the limits the adapter answers kMaxCorrespondenceCost whatever the grid holds
        far = [case["max"][0] + 50 * case["resolution"], case["max"][1] + 50 * case["resolution"]]
        assert sample(abi.GridBuffers(dict(base, cost=np.zeros((ny, nx), np.float32))), far) == 0.9
    assert checked == 13


def test_laser_edge_error_and_aliased_jacobian(olib):
    """computeError uses the true pose; linearizeOplus differentiates the functor with q.w aliased to the range point's x
    (ceres autodiff over StaticParameterDims<6, 3>, TypeOccupiedSpace2D.h:145-179): the restated Jacobian must equal
    central differences of THAT function, and in general differ from the derivative of the error itself."""
    _laser_setup(olib)
    w = synth.make_laser_window()
    wb, gb, _, _, _ = graph_of(olib.oracle_pack_window, abi.default_params(), w)
    g = w["grid"]; Tcr = np.array(list(gb.struct.Tcr)); tq = gb.pose_tq[-1].copy()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    checked = 0
    for k in range(0, len(wb.laser_xyz), 7):
        P = wb.laser_xyz[k].copy(); e = C.c_double(); J = np.zeros(6)
        olib.oracle_laser_edge(p(tq), p(Tcr), p(P), C.byref(wb.grid.struct), C.byref(e), p(J))
        r, c = _laser_coords(tq, tq[6], Tcr, P, g)
        assert abs(e.value - _bicubic(olib, wb.grid, r, c)[0]) < 1e-6
        ra, ca = _laser_coords(tq, P[0], Tcr, P, g)
        if not (PAD + 3 < ra < PAD + g["cost"].shape[0] - 3 and PAD + 3 < ca < PAD + g["cost"].shape[1] - 3):
            continue                                       # aliased point falls outside the grid: constant cost, zero Jacobian
        f, dfr, dfc = _bicubic(olib, wb.grid, ra, ca)
        Jfd = np.zeros(6)
        for i in range(6):
            h = 1e-6
            tp, tm = tq.copy(), tq.copy(); tp[i] += h; tm[i] -= h
            # differences taken without the 5e8 padding offset, which would quantise them to 6e-8
            rp, cp = _laser_coords(tp, P[0], Tcr, P, g, 0.0); rm, cm = _laser_coords(tm, P[0], Tcr, P, g, 0.0)
            Jfd[i] = dfr * (rp - rm) / (2 * h) + dfc * (cp - cm) / (2 * h)
        assert np.abs(J - Jfd).max() <= 1e-5 * max(1.0, np.abs(Jfd).max())
        checked += 1
    assert checked >= 5


def test_laser_factor_enters_the_objective(olib):
    w = synth.make_laser_window(with_visual=True)
    prm = abi.default_params(iterations=10, solver=0)
    wb = abi.WindowBuffers(w); rb = abi.ResultBuffers(6, wb.struct.n_refs)
    assert olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1) == abi.OK
    w2 = dict(w); w2["grid"] = None
    wb2 = abi.WindowBuffers(w2); rb2 = abi.ResultBuffers(6, wb2.struct.n_refs)
    assert olib.oracle_solve_window(C.byref(prm), C.byref(wb2.struct), C.byref(rb2.struct), 1) == abi.OK
    # ~720 points with cost ~0.13 each at information 1 / 0.1
    assert 50.0 < rb.struct.chi2_final - rb2.struct.chi2_final < 400.0


def test_pack_window_matches_hand_computation(olib):
    w = synth.make_window("C1")
    prm = abi.default_params()
    wb, gb, used, oref, mono = graph_of(olib.oracle_pack_window, prm, w)
    assert mono == 0 and gb.n_obs == 3000 and gb.n_poses == 10 and used.all()
    # Twr -> Tcw (Optimizer.cpp:104-109) recomputed with numpy
    Twr = np.asarray(w["pose_Twr"]).reshape(-1, 3, 4)
    Twc = synth.iso_mul(Twr, synth.TRC[None])
    Tcw = synth.iso_inv(Twc)
    R = np.zeros(9); t = np.zeros(3)
    for i in range(gb.n_poses):
        olib.oracle_pose_to_Rt(P(np.ascontiguousarray(gb.pose_tq[i])), P(R), P(t))
        assert np.allclose(R.reshape(3, 3), Tcw[i, :, :3], atol=1e-14) and np.allclose(t, Tcw[i, :, 3], atol=1e-14)
        assert gb.pose_tq[i, 6] >= 0
    # fixed pose = rootId = newest id - 1 (Estimator.cpp:252)
    assert list(np.nonzero(gb.pose_fixed)[0]) == [8]
    # float disparity and float subtraction (Optimizer.cpp:187-188), bit-exact
    u = np.asarray(w["ref_u"], np.float32); depth = np.asarray(w["ref_depth"], np.float32)
    disp = (np.float64(np.float32(w["baseline"])) * np.float64(w["fx"]) / depth.astype(np.float64)).astype(np.float32)
    assert np.array_equal(gb.obs_uvr[:, 0], u.astype(np.float64))
    assert np.array_equal(gb.obs_uvr[:, 2], (u - disp).astype(np.float64))
    assert gb.struct.bf == np.float64(np.float32(0.11)) * 435.2


def test_pack_window_filters(olib):
    """Unknown features / poses are skipped (Optimizer.cpp:158,172); bad depth takes the (skipped) mono branch (:184,:197)."""
    w = synth.make_window("C1")
    w["ref_depth"] = np.asarray(w["ref_depth"]).copy()
    w["ref_depth"][0] = np.nan; w["ref_depth"][1] = -1.0; w["ref_depth"][2] = np.inf
    w["ref_pose"] = np.asarray(w["ref_pose"]).copy(); w["ref_pose"][10] = 999      # pose not in window
    w["point_ids"] = np.asarray(w["point_ids"])[1:]                                # feature 0 not in points3D
    w["point_xyz"] = np.asarray(w["point_xyz"])[1:]; w["point_fixed"] = np.asarray(w["point_fixed"])[1:]
    n_feat0 = int((np.asarray(w["ref_feature"]) == 0).sum())
    prm = abi.default_params()
    wb, gb, used, oref, mono = graph_of(olib.oracle_pack_window, prm, w)
    assert mono == 0 if n_feat0 >= 3 else True     # the three bad depths belong to feature 0, which is filtered first
    assert gb.n_obs == 3000 - n_feat0 - 1
    w2 = synth.make_window("C1"); w2["n_cameras"] = 1                               # baseline unused → every edge is "mono"
    wb, gb, used, oref, mono = graph_of(olib.oracle_pack_window, prm, w2)
    assert gb.n_obs == 0 and mono == 3000


# ---------------------------------------------------------------- solver invariants
def run_oracle(olib, w, threads=1, lib=None, **prm_kw):
    prm = abi.default_params(**prm_kw)
    wb, gb, used, oref, mono = graph_of(olib.oracle_pack_window, prm, w)
    s = oracle_lib.OracleSystem(lib or olib, prm, gb, threads)
    rc, st, sec = s.optimize()
    pose, pt, out, chi = s.download()
    s.close()
    return rc, st, pose, pt, out, chi, gb


def test_zero_noise_window_is_a_fixed_point(olib):
    w = synth.make_window("C1", noise_px=0.0, outlier_frac=0.0, pose_noise_t=0.0, pose_noise_r=0.0, point_noise=0.0)
    rc, st, pose, pt, out, chi, gb = run_oracle(olib, w, iterations=10)
    assert rc == abi.OK and st.n_outliers == 0
    assert st.chi2_initial < 1e-3                       # only float32 key-point / depth rounding is left
    assert np.abs(pose - gb.pose_tq).max() < 1e-6 and np.abs(pt - gb.point_xyz).max() < 1e-4


def test_convergence_and_lm_invariants(olib):
    w = synth.make_window("C1")
    rc, st, pose, pt, out, chi, gb = run_oracle(olib, w, iterations=20)
    assert rc == abi.OK and list(st.iterations_run) == [10, 10]
    tr = np.array([st.trace_chi2[i] for i in range(st.n_trace)])
    assert np.all(np.diff(tr[:10]) <= 1e-9) and np.all(np.diff(tr[10:]) <= 1e-9)   # LM never accepts an increase
    assert st.chi2_final < 0.01 * st.chi2_initial
    Twr = twr_of(olib.oracle_unpack_pose, pose, w["Trc"])
    et, er = synth.pose_errors(Twr, w["truth_Twr"])
    e0t, e0r = synth.pose_errors(w["pose_Twr"], w["truth_Twr"])
    assert et < 0.1 * e0t and er < 0.1 * e0r
    # every gross outlier of the generator is culled by the chi2 > delta rule (Optimizer.cpp:285)
    gross = np.asarray(w["gross"]); ok = ~(gb.pose_fixed[gb.obs_pose].astype(bool) & gb.point_fixed[gb.obs_point].astype(bool))
    assert out[gross & ok].all()


def test_direct_and_pcg_agree(olib):
    w = synth.make_window("C1")
    a = run_oracle(olib, w, iterations=20, solver=0)
    b = run_oracle(olib, w, iterations=20, solver=2)
    assert a[1].pcg_iterations == 0 and b[1].pcg_iterations > 0
    assert rel_err(b[2], a[2]) < 1e-5 and rel_err(b[3], a[3]) < 1e-4 and (a[4] == b[4]).mean() > 0.999


def test_lm_rejects_steps_on_a_hard_start(olib):
    w = hard_window()
    rc, st, *_ = run_oracle(olib, w, iterations=20)
    assert st.trials_run[0] > st.iterations_run[0]      # at least one damped solve was rejected (lambda *= ni path)
    tr = np.array([st.trace_chi2[i] for i in range(st.iterations_run[0])])
    assert np.all(np.diff(tr) <= 1e-9)


def test_gauss_newton_mode(olib):
    w = synth.make_window("C1", pose_noise_t=0.01, pose_noise_r=0.002, point_noise=0.01)
    rc, st, pose, pt, out, chi, gb = run_oracle(olib, w, iterations=10, trust_region=1)
    assert rc == abi.OK and list(st.iterations_run) == [5, 5]
    assert st.chi2_final < 0.05 * st.chi2_initial


def test_no_robust_kernel_means_single_phase(olib):
    w = synth.make_window("C1", outlier_frac=0.0)
    rc, st, *_ = run_oracle(olib, w, iterations=10, robust_kernel_delta=0.0)
    assert rc == abi.OK and list(st.iterations_run) == [5, 0] and st.n_outliers == 0


def test_odometry_edges_tighten_the_solution(olib):
    w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    rc, st, pose, pt, out, chi, gb = run_oracle(olib, w, iterations=20)
    assert rc == abi.OK and gb.struct.n_odo == 11
    Twr = twr_of(olib.oracle_unpack_pose, pose, w["Trc"])
    et, er = synth.pose_errors(Twr, w["truth_Twr"])
    assert et < 0.03 and er < 0.01


def test_openmp_oracle_matches_scalar(olib):
    omp = oracle_lib.load(omp=True)
    w = ragged_window(seed=3)
    a = run_oracle(olib, w, iterations=10)
    b = run_oracle(olib, w, threads=4, lib=omp, iterations=10)
    assert rel_err(b[2], a[2]) < 1e-9 and rel_err(b[3], a[3]) < 1e-9 and np.array_equal(a[4], b[4])


def test_window_level_contract(olib):
    """Error convention of localOptimize (Optimizer.cpp:74, 360-364, 343-358)."""
    prm = abi.default_params(iterations=10)
    w = synth.make_window("C1")
    # (a) single pose → input poses returned
    w1 = dict(w); w1["pose_ids"] = w["pose_ids"][:1]; w1["pose_Twr"] = w["pose_Twr"][:1]
    wb = abi.WindowBuffers(w1); rb = abi.ResultBuffers(1, wb.struct.n_refs)
    rc = olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1)
    assert rc == abi.PASSTHROUGH and rb.struct.n_poses_out == 1 and np.array_equal(rb.pose_Twr_out[0], np.asarray(w["pose_Twr"])[0])
    # (b) first pose id == 0 → error, empty map
    w0 = dict(w); w0["pose_ids"] = np.arange(0, 10, dtype=np.uint64)
    wb = abi.WindowBuffers(w0); rb = abi.ResultBuffers(10, wb.struct.n_refs)
    rc = olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1)
    assert rc == abi.ERR_TOO_FEW_POSES and rb.struct.n_poses_out == 0
    # (c) normal solve: points without references become NaN, others move < 5 m
    w2 = dict(w)
    w2["point_ids"] = np.r_[np.asarray(w["point_ids"]), np.uint64(100000)]
    w2["point_xyz"] = np.vstack([w["point_xyz"], [[1.0, 2.0, 3.0]]]); w2["point_fixed"] = np.r_[w["point_fixed"], np.uint8(0)]
    wb = abi.WindowBuffers(w2); rb = abi.ResultBuffers(10, wb.struct.n_refs)
    rc = olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1)
    assert rc == abi.OK and rb.struct.n_poses_out == 10
    assert np.isnan(wb.point_xyz[-1]).all() and np.isfinite(wb.point_xyz[:-1]).all()
    assert np.linalg.norm(wb.point_xyz[:-1] - np.asarray(w["point_xyz"]), axis=1).max() < 5.0
    fixed = np.asarray(w["point_fixed"]).astype(bool)
    assert np.array_equal(wb.point_xyz[:-1][fixed], np.asarray(w["point_xyz"])[fixed])     # fixed landmarks never move
    assert rb.struct.n_outliers > 0 and len(set(rb.outliers())) == rb.struct.n_outliers
