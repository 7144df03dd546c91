"""K9 — the Levenberg-Marquardt / Gauss-Newton control ([g2o-upstream] OptimizationAlgorithmLevenberg::solve inside
SparseOptimizer::optimize; call sites Optimizer.cpp:93-97, :265, :311) on SCRIPTED trial outcomes, no GPU needed.

Product side: `visfs_ba_hook_lm_script` steps the device-side state machine's own functions (`lin_finalize_update`, `lm_decide`
in ba_kernels.hip, compiled for the host as well) — the same code the kernels run.  Checker: `oracle_lm_script`, the loop the
CPU oracle's solver runs.  Both must produce identical lambda / chi2 traces, iteration and trial counts on every script,
including the failure paths (solver failure, NaN / inf chi2, rho == 0, ten rejected trials, a lambda that overflows)."""
import ctypes as C

import numpy as np
import pytest

from visfs_amd import abi

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


def _run(fn, gn, n_iter, chi0, md0, chi, sc, ok):
    st = abi.Stats()
    chi = np.ascontiguousarray(chi, dtype=np.float64)
    sc = np.ascontiguousarray(sc, dtype=np.float64)
    ok = np.ascontiguousarray(ok, dtype=np.int32)
    used = fn(gn, n_iter, chi0, md0, len(chi), chi.ctypes.data_as(_pd), sc.ctypes.data_as(_pd), ok.ctypes.data_as(_pi), C.byref(st))
    n = st.n_trace
    return dict(used=used, iters=st.iterations_run[0], trials=st.trials_run[0], lam=np.array(st.trace_lambda[:n]),
                chi=np.array(st.trace_chi2[:n]), final=st.chi2_final)


def _same(a, b):
    return (a["used"] == b["used"] and a["iters"] == b["iters"] and a["trials"] == b["trials"]
            and np.array_equal(a["lam"], b["lam"], equal_nan=True) and np.array_equal(a["chi"], b["chi"], equal_nan=True)
            and (a["final"] == b["final"] or (np.isnan(a["final"]) and np.isnan(b["final"]))))


def test_accepted_steps_follow_the_published_schedule(hiplib, olib):
    # every trial accepted with rho = 1: alpha = 1 - (2 rho - 1)^3 = 0 -> scaleFactor = max(1/3, min(alpha, 2/3)) = 1/3
    chi = [90.0, 80.0, 70.0]
    sc = [10.0 - 1e-3] * 3                      # rho = (chi_prev - chi) / (scale + 1e-3) = 1
    for fn in (hiplib.visfs_ba_hook_lm_script, olib.oracle_lm_script):
        r = _run(fn, 0, 3, 100.0, 2.0e5, chi, sc, [1, 1, 1])
        assert r["iters"] == 3 and r["trials"] == 3
        lam0 = 1e-5 * 2.0e5                      # computeLambdaInit, tau = 1e-5
        assert np.allclose(r["lam"], [lam0 / 3, lam0 / 9, lam0 / 27], rtol=1e-15)
        assert list(r["chi"]) == chi and r["final"] == 70.0


def test_ten_rejections_terminate_and_lambda_doubles_its_factor(hiplib, olib):
    # every trial worse than the estimate: lambda *= ni, ni *= 2, ten times, then Terminate (qmax == 10)
    for fn in (hiplib.visfs_ba_hook_lm_script, olib.oracle_lm_script):
        r = _run(fn, 0, 5, 100.0, 1.0e5, [150.0], [1.0], [1])
        assert r["iters"] == 1 and r["trials"] == 10
        assert r["lam"][0] == 1.0 * 2.0 ** (sum(range(1, 11)))          # 2 * 4 * ... * 1024 = 2^55
        assert r["final"] == 100.0


def test_rho_zero_terminates(hiplib, olib):
    for fn in (hiplib.visfs_ba_hook_lm_script, olib.oracle_lm_script):
        r = _run(fn, 0, 5, 100.0, 1.0e5, [100.0], [1.0], [1])          # tempChi == currentChi -> rho == 0: pop, leave the loop, Terminate
        assert r["iters"] == 1 and r["trials"] == 1 and r["final"] == 100.0


def test_non_finite_lambda_leaves_the_loop_and_terminates(hiplib, olib):
    # lambda_init = 1e-5 * 1e307 = 1e302; rejected trials multiply it by 2, 4, 8, ...: after the 5th rejection it is 1e302 * 2^15
    # = 3.3e306, after the 6th 2^21 * 1e302 = inf -> `break` before qmax++ and solve() returns Terminate
    # (optimization_algorithm_levenberg.cpp: `|| !g2o_isfinite(_currentLambda)`), so the phase ends after ONE outer iteration.
    for fn in (hiplib.visfs_ba_hook_lm_script, olib.oracle_lm_script):
        r = _run(fn, 0, 6, 100.0, 1.0e307, [150.0], [1.0], [1])
        assert r["iters"] == 1, r
        assert r["trials"] == 6 and np.isinf(r["lam"][0]) and r["final"] == 100.0


def test_solver_failure_is_a_rejected_trial(hiplib, olib):
    # ok = 0 -> tempChi = max double, rho < 0 -> rejected; the next (successful, better) trial is accepted
    for fn in (hiplib.visfs_ba_hook_lm_script, olib.oracle_lm_script):
        r = _run(fn, 0, 1, 100.0, 1.0e5, [50.0, 60.0], [5.0, 5.0], [0, 1])
        assert r["iters"] == 1 and r["trials"] == 2 and r["final"] == 60.0


def test_gauss_newton_takes_every_step_and_stops_on_failure(hiplib, olib):
    for fn in (hiplib.visfs_ba_hook_lm_script, olib.oracle_lm_script):
        r = _run(fn, 1, 4, 100.0, 1.0e5, [150.0, 120.0, 90.0, 95.0], [1.0] * 4, [1, 1, 0, 1])
        assert r["iters"] == 3 and r["trials"] == 3            # the failed solve ends the phase (Fail)
        assert list(r["lam"]) == [0.0, 0.0, 0.0] and list(r["chi"]) == [100.0, 150.0, 120.0] and r["final"] == 120.0


@pytest.mark.parametrize("seed", range(4))
def test_random_scripts_device_functions_equal_the_checker(hiplib, olib, seed):
    rng = np.random.default_rng(20261004 + seed)
    for _ in range(3000):
        n = int(rng.integers(1, 40))
        chi0 = float(rng.uniform(1, 1000))
        chi = chi0 * rng.uniform(0.2, 1.6, size=n)
        sc = rng.uniform(-1, 50, size=n)
        md0 = float(10 ** rng.uniform(-3, 6))
        kind = int(rng.integers(0, 8))
        if kind == 0: chi[rng.integers(0, n)] = np.nan
        if kind == 1: chi[rng.integers(0, n)] = np.inf
        if kind == 2: chi[:] = chi0
        if kind == 3: sc[rng.integers(0, n)] = -1e-3               # scale + 1e-3 == 0 -> rho = +-inf or NaN
        if kind == 4: md0 = 1e300
        if kind == 5: md0 = float(10 ** rng.uniform(290, 308)); chi = chi0 * rng.uniform(1.0, 1.6, size=n)
        ok = (rng.uniform(size=n) > 0.1).astype(np.int32)
        gn, n_iter = int(kind == 6), int(rng.integers(0, 12))
        a = _run(hiplib.visfs_ba_hook_lm_script, gn, n_iter, chi0, md0, chi, sc, ok)
        b = _run(olib.oracle_lm_script, gn, n_iter, chi0, md0, chi, sc, ok)
        assert _same(a, b), (kind, gn, n_iter, md0, a, b)
