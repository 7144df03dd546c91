"""Optimizer/Framework=1 — the control of the Ceres branch (Optimizer.cpp:504-527: ceres::Solve with default options but
max_num_iterations): [ceres-upstream] TrustRegionMinimizer + LevenbergMarquardtStrategy on SCRIPTED outcomes, no GPU needed.

Product side: `visfs_ba_hook_ceres_script` steps the device-side state machine's own functions (`ceres_lin_update`, `ceres_decide` in
ba_kernels.hip, compiled for the host as well).  Checker: `oracle_ceres_script`, the loop the CPU oracle's solver runs.  The
known-answer cases are worked out by hand from the published rules (radius / max(1/3, 1 - (2 rho - 1)^3) on an accepted step,
radius / decrease_factor with the factor doubling on a rejected one, radius / 2 on an invalid one, five invalid steps in a row end
the solve, the three tolerances), and random scripts must give identical traces on both sides."""
import ctypes as C

import numpy as np
import pytest

from visfs_amd import abi

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
REASON = dict(max_iter=1, gradient=2, parameter=3, function=4, min_radius=5, invalid=6)


def _run(fn, max_iter, cost0, x0, g0, ok, mcc, cand, step, gmax, xn):
    st = abi.Stats()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (mcc, cand, step, gmax, xn)]
    okv = np.ascontiguousarray(ok, dtype=np.int32)
    reason = fn(max_iter, cost0, x0, g0, len(okv), okv.ctypes.data_as(_pi), *[v.ctypes.data_as(_pd) for v in a], C.byref(st))
    n = st.n_trace
    return dict(reason=reason, iters=st.iterations_run[0], trials=st.trials_run[0], radius=np.array(st.trace_lambda[:n]),
                cost2=np.array(st.trace_chi2[:n]), final=st.chi2_final)


def _both(hiplib, olib):
    return (hiplib.visfs_ba_hook_ceres_script, olib.oracle_ceres_script)


def test_accepted_steps_grow_the_radius_by_three(hiplib, olib):
    # rho = cost_change / model_cost_change = 1 every time: radius / max(1/3, 1 - 1) = 3 x radius, from 1e4
    cand = [50.0, 25.0, 12.5]
    mcc = [50.0, 25.0, 12.5]
    for fn in _both(hiplib, olib):
        r = _run(fn, 3, 100.0, 10.0, 1.0, [1, 1, 1], mcc, cand, [1.0] * 3, [1.0] * 3, [10.0] * 3)
        assert r["reason"] == REASON["max_iter"] and r["iters"] == 3
        assert np.allclose(r["radius"], [3e4, 9e4, 2.7e5], rtol=1e-15)
        assert list(r["cost2"]) == [100.0, 50.0, 25.0] and r["final"] == 25.0


def test_rejected_steps_shrink_by_a_doubling_factor_and_an_accepted_step_resets_it(hiplib, olib):
    # three rejections (cost goes up): radius / 2, / 4, / 8; then rho = 0.5 -> radius / max(1/3, 1 - 0) = radius; factor back to 2
    ok = [1, 1, 1, 1, 1]
    cand = [120.0, 120.0, 120.0, 90.0, 120.0]
    mcc = [10.0, 10.0, 10.0, 20.0, 10.0]
    for fn in _both(hiplib, olib):
        r = _run(fn, 5, 100.0, 10.0, 1.0, ok, mcc, cand, [1.0] * 5, [1.0] * 5, [10.0] * 5)
        assert r["iters"] == 5 and r["reason"] == REASON["max_iter"]
        assert np.allclose(r["radius"], [1e4 / 2, 1e4 / 8, 1e4 / 64, 1e4 / 64, 1e4 / 128], rtol=1e-15)
        assert list(r["cost2"]) == [200.0, 200.0, 200.0, 180.0, 180.0]


def test_a_decrease_below_min_relative_decrease_is_rejected(hiplib, olib):
    # rho = 1e-3 exactly is NOT > min_relative_decrease
    for fn in _both(hiplib, olib):
        r = _run(fn, 1, 100.0, 10.0, 1.0, [1], [1000.0], [99.0], [1.0], [1.0], [10.0])
        assert r["cost2"][0] == 200.0 and r["radius"][0] == 5e3
        r = _run(fn, 1, 100.0, 10.0, 1.0, [1], [999.0], [99.0], [1.0], [1.0], [10.0])
        assert r["cost2"][0] == 198.0


def test_five_invalid_steps_in_a_row_end_the_solve(hiplib, olib):
    # a failed linear solve or a non-positive model cost change is an invalid step: radius / 2, and the fifth one is fatal
    for ok, mcc in (([0], [1.0]), ([1], [0.0]), ([1], [-3.0]), ([1], [float("nan")])):
        for fn in _both(hiplib, olib):
            r = _run(fn, 50, 100.0, 10.0, 1.0, ok, mcc, [50.0], [1.0], [1.0], [10.0])
            assert r["reason"] == REASON["invalid"] and r["iters"] == 5
            assert np.allclose(r["radius"], [5e3, 2.5e3, 1.25e3, 625.0, 625.0], rtol=1e-15)
            assert r["final"] == 200.0
    # a valid step in between restarts the count
    for fn in _both(hiplib, olib):
        r = _run(fn, 9, 100.0, 10.0, 1.0, [0, 0, 0, 0, 1, 0, 0, 0, 0], [10.0] * 9, [120.0] * 9, [1.0] * 9, [1.0] * 9, [10.0] * 9)
        assert r["reason"] == REASON["max_iter"] and r["iters"] == 9


def test_the_three_tolerances(hiplib, olib):
    for fn in _both(hiplib, olib):
        # gradient tolerance at iteration zero: nothing runs
        r = _run(fn, 10, 100.0, 10.0, 1e-10, [1], [1.0], [50.0], [1.0], [1.0], [10.0])
        assert r["reason"] == REASON["gradient"] and r["iters"] == 0 and r["final"] == 200.0
        # ... and after an accepted step
        r = _run(fn, 10, 100.0, 10.0, 1.0, [1], [50.0], [50.0], [1.0], [5e-11], [10.0])
        assert r["reason"] == REASON["gradient"] and r["iters"] == 1 and r["final"] == 100.0
        # parameter tolerance: ||step|| <= 1e-8 (||x|| + 1e-8); the step is not taken
        r = _run(fn, 10, 100.0, 10.0, 1.0, [1], [50.0], [50.0], [1e-7], [1.0], [10.0])
        assert r["reason"] == REASON["parameter"] and r["iters"] == 1 and r["final"] == 200.0
        r = _run(fn, 1, 100.0, 10.0, 1.0, [1], [50.0], [50.0], [1.0001e-7], [1.0], [10.0])
        assert r["reason"] == REASON["max_iter"] and r["final"] == 100.0
        # function tolerance: |cost change| <= 1e-6 cost; the step is not taken either (trust_region_minimizer.cc returns before
        # the step-quality test)
        r = _run(fn, 10, 100.0, 10.0, 1.0, [1], [1e-4], [100.0 - 5e-5], [1.0], [1.0], [10.0])
        assert r["reason"] == REASON["function"] and r["iters"] == 1 and r["final"] == 200.0
        # a cost that went UP by less than the tolerance also ends the solve
        r = _run(fn, 10, 100.0, 10.0, 1.0, [1], [1e-4], [100.0 + 5e-5], [1.0], [1.0], [10.0])
        assert r["reason"] == REASON["function"]


def test_a_candidate_that_cannot_be_evaluated_is_rejected(hiplib, olib):
    for bad in (float("nan"), float("inf")):
        for fn in _both(hiplib, olib):
            r = _run(fn, 2, 100.0, 10.0, 1.0, [1, 1], [10.0, 10.0], [bad, 80.0], [1.0, 1.0], [1.0, 1.0], [10.0, 10.0])
            assert r["iters"] == 2 and list(r["cost2"]) == [200.0, 160.0] and r["radius"][0] == 5e3


def test_the_radius_is_capped_and_a_vanishing_radius_ends_the_solve(hiplib, olib):
    for fn in _both(hiplib, olib):
        n = 40
        cand = 100.0 * 0.5 ** np.arange(1, n + 1)
        mcc = np.concatenate([[50.0], cand[:-1] - cand[1:]])            # rho = 1 each time
        r = _run(fn, n, 100.0, 10.0, 1.0, [1] * n, mcc, cand, [1.0] * n, [1.0] * n, [10.0] * n)
        assert r["radius"].max() == 1e16 and r["radius"][-1] == 1e16
        # rejections: 1e4 / 2^(k(k+1)/2) drops below 1e-32 at k = 15 (2^120 > 1e36)
        r = _run(fn, 64, 100.0, 10.0, 1.0, [1], [10.0], [150.0], [1.0], [1.0], [10.0])
        assert r["reason"] == REASON["min_radius"] and r["iters"] == 15


def test_random_scripts_match_the_checker(hiplib, olib):
    rng = np.random.default_rng(20261004)
    for case in range(3000):
        n = int(rng.integers(1, 24))
        max_iter = int(rng.integers(0, 30))
        cost0 = float(10.0 ** rng.uniform(-3, 6))
        ok = (rng.random(n) > 0.15).astype(np.int32)
        cand = cost0 * np.exp(rng.normal(-0.1, 0.4, n).cumsum() * (rng.random() < 0.7) + rng.normal(0, 0.3, n) * (rng.random() < 0.5))
        mcc = np.abs(rng.normal(0, cost0 * 0.2, n)) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
        if rng.random() < 0.2:
            cand[rng.integers(0, n)] = rng.choice([np.nan, np.inf, cost0, cost0 * (1 + 1e-7)])
        step = 10.0 ** rng.uniform(-9, 1, n)
        gmax = 10.0 ** rng.uniform(-11, 3, n)
        xn = 10.0 ** rng.uniform(-2, 3, n)
        g0 = float(10.0 ** rng.uniform(-11, 3))
        args = (max_iter, cost0, float(xn[0]), g0, ok, mcc, cand, step, gmax, xn)
        a = _run(hiplib.visfs_ba_hook_ceres_script, *args)
        b = _run(olib.oracle_ceres_script, *args)
        assert a["reason"] == b["reason"] and a["iters"] == b["iters"], (case, a, b)
        assert np.array_equal(a["radius"], b["radius"]) and np.array_equal(a["cost2"], b["cost2"], equal_nan=True), (case, a, b)
        assert a["final"] == b["final"], case


# ------------------------------------------------------------------ Optimizer/TrustRegion=1: [ceres-upstream] DoglegStrategy
def _run_dl(fn, max_iter, cost0, x0, g0, ok, mcc, cand, step, dlnorm, gmax, xn):
    st = abi.Stats()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (mcc, cand, step, dlnorm, gmax, xn)]
    okv = np.ascontiguousarray(ok, dtype=np.int32)
    mu = np.full(abi.MAX_TRACE, np.nan)
    reason = fn(max_iter, cost0, x0, g0, len(okv), okv.ctypes.data_as(_pi), *[v.ctypes.data_as(_pd) for v in a], C.byref(st), mu.ctypes.data_as(_pd))
    n = st.n_trace
    return dict(reason=reason, iters=st.iterations_run[0], radius=np.array(st.trace_lambda[:n]), cost2=np.array(st.trace_chi2[:n]),
                final=st.chi2_final, mu=mu[:n].copy())


def _both_dl(hiplib, olib):
    return (hiplib.visfs_ba_hook_dogleg_script, olib.oracle_dogleg_script)


def test_dogleg_radius_rules(hiplib, olib):
    for fn in _both_dl(hiplib, olib):
        # rho = 1 > 0.75: radius = max(radius, 3 x the step's scaled length); a short step leaves the radius alone
        r = _run_dl(fn, 2, 100.0, 10.0, 1.0, [1, 1], [50.0, 25.0], [50.0, 25.0], [1.0, 1.0], [5e3, 10.0], [1.0, 1.0], [10.0, 10.0])
        assert list(r["radius"]) == [1.5e4, 1.5e4] and list(r["cost2"]) == [100.0, 50.0]
        # 1e-3 < rho < 0.25: the step is taken and the radius halves; 0.25 <= rho <= 0.75: unchanged
        r = _run_dl(fn, 2, 100.0, 10.0, 1.0, [1, 1], [100.0, 40.0], [90.0, 70.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0], [10.0, 10.0])
        assert list(r["radius"]) == [5e3, 5e3] and list(r["cost2"]) == [180.0, 140.0]
        # a rejected step halves the radius every time (no doubling factor)
        r = _run_dl(fn, 3, 100.0, 10.0, 1.0, [1], [10.0], [120.0], [1.0], [1.0], [1.0], [10.0])
        assert list(r["radius"]) == [5e3, 2.5e3, 1.25e3] and r["final"] == 200.0


def test_dogleg_mu_rules(hiplib, olib):
    for fn in _both_dl(hiplib, olib):
        # an invalid step (failed factorisation / non-positive model change) multiplies mu by 10 and keeps the radius
        r = _run_dl(fn, 4, 100.0, 10.0, 1.0, [0, 1, 1, 1], [1.0, -1.0, 50.0, 10.0], [50.0] * 4, [1.0] * 4, [1.0] * 4, [1.0] * 4, [10.0] * 4)
        assert np.allclose(r["mu"][:2], [1e-7, 1e-6], rtol=1e-15) and list(r["radius"][:2]) == [1e4, 1e4]
        # an accepted step: mu = max(1e-8, 2 mu / 10)
        assert np.isclose(r["mu"][2], 2e-7, rtol=1e-15) and r["cost2"][2] == 100.0
        # a rejected one leaves mu alone
        assert r["mu"][3] == r["mu"][2]
        # five invalid steps in a row are fatal here too
        r = _run_dl(fn, 50, 100.0, 10.0, 1.0, [0], [1.0], [50.0], [1.0], [1.0], [1.0], [10.0])
        assert r["reason"] == REASON["invalid"] and r["iters"] == 5 and np.allclose(r["mu"], [1e-7, 1e-6, 1e-5, 1e-4, 1e-4], rtol=1e-15)


def test_random_dogleg_scripts_match_the_checker(hiplib, olib):
    rng = np.random.default_rng(20261005)
    for case in range(3000):
        n = int(rng.integers(1, 24))
        max_iter = int(rng.integers(0, 30))
        cost0 = float(10.0 ** rng.uniform(-3, 6))
        ok = (rng.random(n) > 0.15).astype(np.int32)
        cand = cost0 * np.exp(rng.normal(-0.1, 0.4, n).cumsum() * (rng.random() < 0.7) + rng.normal(0, 0.3, n) * (rng.random() < 0.5))
        mcc = np.abs(rng.normal(0, cost0 * 0.2, n)) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
        if rng.random() < 0.2:
            cand[rng.integers(0, n)] = rng.choice([np.nan, np.inf, cost0, cost0 * (1 + 1e-7)])
        step = 10.0 ** rng.uniform(-9, 1, n)
        dln = 10.0 ** rng.uniform(-2, 5, n)
        gmax = 10.0 ** rng.uniform(-11, 3, n)
        xn = 10.0 ** rng.uniform(-2, 3, n)
        g0 = float(10.0 ** rng.uniform(-11, 3))
        args = (max_iter, cost0, float(xn[0]), g0, ok, mcc, cand, step, dln, gmax, xn)
        a = _run_dl(hiplib.visfs_ba_hook_dogleg_script, *args)
        b = _run_dl(olib.oracle_dogleg_script, *args)
        assert a["reason"] == b["reason"] and a["iters"] == b["iters"], (case, a, b)
        assert np.array_equal(a["radius"], b["radius"]) and np.array_equal(a["cost2"], b["cost2"], equal_nan=True), (case, a, b)
        assert np.array_equal(a["mu"], b["mu"]) and a["final"] == b["final"], (case, a, b)


def _combine(fn, *args):
    out = np.zeros(4)
    fn(*[float(a) for a in args], out.ctypes.data_as(_pd))
    return out


def test_the_point_on_the_dogleg_path(hiplib, olib):
    """dogleg_combine (what k_dogleg_mid calls) against the checker and against the geometry it restates: a small dense problem, the
    scaled Gauss-Newton and Cauchy points built with NumPy, the three cases of ComputeTraditionalDoglegStep."""
    rng = np.random.default_rng(7)
    seen = set()
    for case in range(600):
        n, m_ = int(rng.integers(2, 9)), int(rng.integers(9, 20))
        J = rng.normal(size=(m_, n)) * 10.0 ** rng.uniform(-1, 1, n)
        f = rng.normal(size=m_)
        H, g = J.T @ J, J.T @ f
        M = np.clip(np.diag(H), 1e-6, 1e32)
        mu = float(10.0 ** rng.uniform(-8, -2))
        dn = -np.linalg.solve(H + mu * np.diag(M), g)
        v = g / M
        S1, S2, S3, JV2 = float(g @ v), float(dn @ (M * dn)), float(g @ dn), float((J @ v) @ (J @ v))
        radius = float(np.sqrt(S2) * 10.0 ** rng.uniform(-2.5, 0.5))
        a = _combine(hiplib.visfs_ba_hook_dogleg_combine, S1, S2, S3, JV2, radius, mu)
        b = _combine(olib.oracle_dogleg_combine, S1, S2, S3, JV2, radius, mu)
        assert np.array_equal(a, b), (case, a, b)
        A, B, norm, mcc = a
        # the geometry, in the scaled space y = sqrt(M) x
        gs, gns = g / np.sqrt(M), np.sqrt(M) * dn
        cauchy = -(S1 / JV2) * gs
        step_s = A * gs + B * gns
        assert np.isclose(np.linalg.norm(step_s), norm, rtol=1e-10)
        if np.linalg.norm(gns) <= radius:
            seen.add(1); assert A == 0.0 and B == 1.0
        elif np.linalg.norm(cauchy) >= radius:
            seen.add(2); assert B == 0.0 and np.isclose(norm, radius, rtol=1e-12) and np.allclose(step_s, -radius * gs / np.linalg.norm(gs), rtol=1e-12)
        else:
            seen.add(3)
            assert np.isclose(norm, radius, rtol=1e-9)
            beta = B
            assert 0.0 <= beta <= 1.0 and np.allclose(step_s, cauchy + beta * (gns - cauchy), rtol=1e-9, atol=1e-12 * radius)
        # the model cost change of the unregularised model at the unscaled step
        step = step_s / np.sqrt(M)
        want = -(g @ step + 0.5 * step @ (H @ step))
        assert np.isclose(mcc, want, rtol=1e-8, atol=1e-12 * abs(g @ step)), (case, mcc, want)
    assert seen == {1, 2, 3}
