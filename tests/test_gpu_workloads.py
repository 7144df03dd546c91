"""GPU parity tests of the BASELINE.json workloads that round 1 left without a `-m gpu` test, and of the residency rules of the
persistent PCG (one grid at a time per device; a batched launch never carries more block rows than the device holds)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle_lib
from helpers import graph_of, rel_err
from test_gpu_parity import check_optimize, make_pair
from visfs_amd import abi, synth

pytestmark = pytest.mark.gpu


def test_c5_per_gpu_workload_eight_resident_c2_windows(olib):
    """BASELINE config 5, one GPU's share: 8 independent C2-size windows resident side by side, solved by ONE sequence of batched
    launches (visfs_ba_batch_upload / optimize, blockIdx.y = window).  Every window must come out bit-identical to its own solve
    as a batch of one (same launch geometry class, same chunking), and one of them is checked against the CPU oracle."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=20, solver=2)
    gbs = []
    for i in range(8):
        w = synth.make_window("C5", window_index=i)
        wb, gb, *_ = graph_of(olib.oracle_pack_window, prm, w)
        gbs.append(gb)
    s = backend.Solver(prm)
    s.batch_upload(gbs)
    s.batch_reset()
    rc, stats = s.batch_optimize()
    assert rc == abi.OK and all(st.status == abi.OK for st in stats)
    assert all(st.iterations_run[0] + st.iterations_run[1] == 20 for st in stats)
    got = [s.batch_download(i) for i in range(8)]
    # (a) each window alone, as a batch of one
    for i in (0, 3, 7):
        s1 = backend.Solver(prm)
        s1.batch_upload([gbs[i]]); s1.batch_reset()
        rc1, st1 = s1.batch_optimize()
        ref = s1.batch_download(0)
        s1.close()
        assert rc1 == abi.OK
        assert all(np.array_equal(a, b) for a, b in zip(got[i], ref)), f"window {i} differs from its single solve"
        assert list(st1[0].iterations_run) == list(stats[i].iterations_run) and st1[0].pcg_iterations == stats[i].pcg_iterations
    # (b) the oracle on one of them: same LM trajectory, outliers, poses
    o = oracle_lib.OracleSystem(olib, prm, gbs[5])
    rco, sto, _ = o.optimize()
    po, pto, outo, chio = o.download(); o.close()
    assert rco == abi.OK and list(sto.iterations_run) == list(stats[5].iterations_run) and list(sto.trials_run) == list(stats[5].trials_run)
    assert sto.pcg_iterations == stats[5].pcg_iterations
    assert np.array_equal(outo, got[5][2])
    assert rel_err(got[5][0], po) < 1e-6 and rel_err(got[5][1], pto) < 1e-6
    # (c) a second batched run reproduces the first bit for bit
    s.batch_reset(); s.batch_optimize()
    assert all(all(np.array_equal(a, b) for a, b in zip(s.batch_download(i), got[i])) for i in range(8))
    s.close()


def test_c4_direct_solver_parity(olib):
    """BASELINE config 4 with the reference-default linear solver (Optimizer/Solver=0, Parameters.h:185): the 1194 x 1194 reduced
    camera system (padded to 1216: 38 panels) through the blocked Cholesky with the fp64-MFMA trailing update."""
    o, s, gb = make_pair(olib, synth.make_window("C4"), iterations=10, solver=0)
    assert s.describe()["n_free_poses"] == 199
    check_optimize(o, s, pose_tol=1e-5)
    a = s.download()
    s.reset(); s.optimize()
    assert all(np.array_equal(x, y) for x, y in zip(a, s.download()))          # run-to-run bitwise identical
    s.close(); o.close()


def test_two_handles_with_persistent_pcg_on_two_threads(olib):
    """Two Optimizer instances on one device, both with Optimizer/Solver=2 on C4-size windows (199 block rows each), driven from
    two threads at once: their persistent-PCG grids must not be interleaved on the device (each needs all its workgroups
    resident).  Both must finish OK with the results of a sequential run."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("C4", window_index=i) for i in range(2)]
    solvers, refs = [], []
    for w in ws:
        s = backend.Solver(prm)
        gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
        s.upload(gb)
        rc, st = s.optimize()
        assert rc == abi.OK
        refs.append(s.download())
        solvers.append(s)
    results = [None, None]

    def run(k):
        out = []
        for _ in range(3):
            solvers[k].reset()
            rc, st = solvers[k].optimize()
            out.append((rc, solvers[k].download()))
        results[k] = out

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(2):
        for rc, out in results[k]:
            assert rc == abi.OK
            assert all(np.array_equal(a, b) for a, b in zip(out, refs[k]))
        solvers[k].close()


def test_one_wave_pcg_grids_of_two_handles_run_side_by_side(olib):
    """Grids of the one-wave PCG kernel (k_pcg1: C2-size windows, 49 block rows each) are homogeneous — 1024 of their waves are
    resident together wherever the dispatcher puts them — so the per-device budget admits them concurrently (two handles, two host
    threads, two streams: their kernels overlap on the device).  Results must be those of sequential runs, bit for bit, every time."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=20, solver=2)
    solvers, refs = [], []
    for i in range(2):
        s = backend.Solver(prm)
        gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(synth.make_window("C5", window_index=i)))
        s.upload(gb)
        assert s.describe()["n_free_poses"] == 49
        rc, st = s.optimize()
        assert rc == abi.OK
        refs.append(s.download()); solvers.append(s)
    results = [None, None]

    def run(k):
        out = []
        for _ in range(8):
            solvers[k].reset()
            rc, st = solvers[k].optimize()
            out.append((rc, solvers[k].download()))
        results[k] = out

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(2):
        for rc, out in results[k]:
            assert rc == abi.OK and all(np.array_equal(a, b) for a, b in zip(out, refs[k]))
        solvers[k].close()


def test_batch_above_the_residency_cap_splits_and_stays_bit_identical(olib, monkeypatch):
    """A batched launch sequence may carry only as many PCG block rows as the device holds at once (occupancy query x CUs); a
    larger batch is cut into several sequences.  VISFS_BA_PCG_CAPACITY forces a tiny cap: six 29-row windows then run as batches
    of two, and every result equals the uncapped run."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=40 + i) for i in range(6)]
    s = backend.Solver(prm)
    ref = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    s.close()
    monkeypatch.setenv("VISFS_BA_PCG_CAPACITY", "64")          # 64 // 29 = 2 windows per launch sequence
    s = backend.Solver(prm)
    got = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    s.close()
    for a, b in zip(ref, got):
        assert a.struct.status == b.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert a.struct.chi2_final == b.struct.chi2_final


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_the_three_pcg_kernels_agree(olib, monkeypatch, cfg):
    """49 free poses, block rows of <= 21 blocks: k_pcg_cu (the whole solve in ONE workgroup, S in registers, no cross-workgroup
    hand-off — opt-in, VISFS_BA_PCG_CU=1), k_pcg1 (one wavefront per block row, granule hand-offs: the default up to 64 free poses) and
    k_pcg (four waves per block row: the default beyond) run the same recurrences with different associations of the mat-vec sums:
    same iteration counts, results equal to rounding; the suite's oracle parity tests run on the default."""
    from test_gpu_parity import _solve_in_mode
    w = synth.make_window(cfg)
    runs = {}
    for name, env in (("cu", dict(VISFS_BA_PCG_CU="1")), ("one_wave", dict(VISFS_BA_PCG_CU="0", VISFS_BA_PCG1="1")),
                      ("four_wave", dict(VISFS_BA_PCG_CU="0", VISFS_BA_PCG1="0"))):
        info, rc, st, out = _solve_in_mode(monkeypatch, w, env, iterations=20, solver=2)
        assert rc == abi.OK
        runs[name] = (st, out)
    st0, out0 = runs["four_wave"]
    for name in ("cu", "one_wave"):
        st1, out1 = runs[name]
        assert list(st0.iterations_run) == list(st1.iterations_run) and list(st0.trials_run) == list(st1.trials_run), name
        assert st0.pcg_iterations == st1.pcg_iterations, name
        assert np.array_equal(out0[2], out1[2]), name
        assert rel_err(out1[0], out0[0]) < 1e-10 and rel_err(out1[1], out0[1]) < 1e-10, name


@pytest.mark.parametrize("n_kf", [12, 33, 57])
def test_single_workgroup_pcg_at_the_edges_of_its_range(olib, monkeypatch, n_kf):
    """k_pcg_cu, a thread per scalar row (round 4): 11 free poses (two wavefronts, every block in registers), 32 (exactly three full
    wavefronts: no idle lane to hide a slip), 56 (the most it takes: six wavefronts, block rows beyond the register part in LDS slices).
    Against the four-wave kernel: same iteration counts, results equal to rounding."""
    from test_gpu_parity import _solve_in_mode
    w = synth.make_window("custom", n_kf=n_kf, n_lm=40 * n_kf, n_obs=400 * n_kf, seed=400 + n_kf)
    runs = {}
    for name, env in (("cu", dict(VISFS_BA_PCG_CU="1")), ("four_wave", dict(VISFS_BA_PCG_CU="0", VISFS_BA_PCG1="0"))):
        info, rc, st, out = _solve_in_mode(monkeypatch, w, env, iterations=20, solver=2)
        assert rc == abi.OK
        assert info["solver_kernel"] == (4 if name == "cu" else 2), (name, info["solver_kernel"])
        runs[name] = (st, out)
    (st0, out0), (st1, out1) = runs["four_wave"], runs["cu"]
    assert list(st0.iterations_run) == list(st1.iterations_run) and list(st0.trials_run) == list(st1.trials_run)
    assert st0.pcg_iterations == st1.pcg_iterations
    assert np.array_equal(out0[2], out1[2])
    assert rel_err(out1[0], out0[0]) < 1e-10 and rel_err(out1[1], out0[1]) < 1e-10


def test_single_workgroup_pcg_in_a_batch_of_unequal_windows(olib, monkeypatch):
    """One batched k_pcg_cu launch is as wide as its widest member and carries the LDS of its hungriest: members of 13, 29, 49 and 56 free
    poses (different wavefront counts, with and without LDS slices) each get the bytes of their own solve as a batch of one."""
    from visfs_amd import backend
    monkeypatch.setenv("VISFS_BA_PCG_CU", "1")
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("custom", n_kf=k, n_lm=40 * k, n_obs=400 * k, seed=500 + k) for k in (14, 57, 30, 50, 14, 57)]
    s = backend.Solver(prm)
    singles = [s.solve_batch([abi.WindowBuffers(w)])[0] for w in ws]
    got = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    s.close()
    for a, b in zip(singles, got):
        assert a.struct.status == b.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run)


@pytest.mark.parametrize("solver", [2, 0])
def test_a_lone_mid_size_window_is_chunked_in_two_passes_and_says_so(olib, monkeypatch, solver):
    """The one way a window's bits depend on how it was submitted (ba_api.cpp, sch_passes): a window of 1 024 .. 6 144 Schur chunks solved
    through the single-window entry points adds two co-observation pairs per lane before the wave's reduction (+4-5 % for a lone C2, -4..-7 %
    inside a batched launch: profiles/r01_v8_schur_passes.log), a batch member one.  The two solves agree to rounding with the same LM
    trajectory; under VISFS_BA_SCH_PASSES=1 they are the same bytes.  (Batches of any size, cut or sharded any way, always agree with the
    batch of one: test_a_window_s_result_never_depends_on_the_size_of_its_batch, test_c5_*.)"""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=solver)
    w = synth.make_window("custom", n_kf=50, n_lm=2000, n_obs=20000, seed=550)
    def both():
        s = backend.Solver(prm)
        rc, a = s.solve_window(abi.WindowBuffers(w))
        b = s.solve_batch([abi.WindowBuffers(w)])[0]
        s.close()
        assert rc == b.struct.status == abi.OK
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run) and a.outliers() == b.outliers()
        return a, b
    a, b = both()
    et, er = synth.pose_errors(b.pose_Twr_out[:50], a.pose_Twr_out[:50])
    assert et < 1e-10 and er < 1e-10
    monkeypatch.setenv("VISFS_BA_SCH_PASSES", "1")
    a, b = both()
    assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out)


def test_a_handle_tuned_for_throughput_runs_its_windows_on_the_single_workgroup_pcg(olib, monkeypatch):
    """ABI 8, visfs_ba_set_tuning: the PCG kernel follows the HANDLE (THROUGHPUT: k_pcg_cu for every window of <= 56 free poses it uploads
    afterwards), never the size of a batch — through one handle a window is the same bytes alone, as a batch of one and among others; the
    two tunings agree to rounding with identical iteration counts; VISFS_BA_PCG_CU=0|1 overrides every handle."""
    from visfs_amd import backend
    monkeypatch.delenv("VISFS_BA_PCG_CU", raising=False)
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("custom", n_kf=14, n_lm=300, n_obs=2400, seed=620 + i) for i in range(6)]
    st = backend.Solver(prm, tuning=abi.TUNE_THROUGHPUT)
    alone = [st.solve_window(abi.WindowBuffers(w)) for w in ws]
    ones = [st.solve_batch([abi.WindowBuffers(w)])[0] for w in ws]
    among = st.solve_batch([abi.WindowBuffers(w) for w in ws])
    gb, *_ = abi.pack_window_with(st.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(ws[0]))
    st.upload(gb); assert st.describe()["solver_kernel"] == 4                      # k_pcg_cu
    st.set_tuning(abi.TUNE_LATENCY); st.upload(gb); assert st.describe()["solver_kernel"] == 1     # later uploads follow the new tuning
    st.close()
    sl = backend.Solver(prm)
    lat = [sl.solve_window(abi.WindowBuffers(w)) for w in ws]
    sl.upload(gb); assert sl.describe()["solver_kernel"] == 1                      # k_pcg1: the default
    sl.close()
    differ = 0
    for (rc, a), b, c, (rcl, l) in zip(alone, ones, among, lat):
        assert rc == rcl == b.struct.status == c.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and np.array_equal(b.pose_Twr_out, c.pose_Twr_out) and a.outliers() == c.outliers()
        assert list(a.struct.iterations_run) == list(l.struct.iterations_run) and a.outliers() == l.outliers()
        et, er = synth.pose_errors(a.pose_Twr_out[:14], l.pose_Twr_out[:14])
        assert et < 1e-9 and er < 1e-9
        differ += int(not np.array_equal(a.pose_Twr_out, l.pose_Twr_out))
    assert differ > 0                                                               # a different kernel really ran
    monkeypatch.setenv("VISFS_BA_PCG_CU", "0")
    so = backend.Solver(prm, tuning=abi.TUNE_THROUGHPUT)
    so.upload(gb); assert so.describe()["solver_kernel"] == 1                      # the environment overrides the handle
    so.close()


def test_batch_sharded_over_handles_equals_one_handle(olib):
    """visfs_ba_solve_batch_sharded: config 5 inside one process — the windows in contiguous blocks over several handles (one per GPU
    on a node; two on this one-GPU box), each block solved by visfs_ba_solve_batch on its own host thread.  Results are those of one
    handle solving all of them, bit for bit; a handle listed twice is refused."""
    from visfs_amd import backend
    import ctypes as C
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("custom", n_kf=20, n_lm=500, n_obs=4000, seed=80 + i) for i in range(7)]
    one = backend.Solver(prm)
    ref = one.solve_batch([abi.WindowBuffers(w) for w in ws])
    a, b, c = backend.Solver(prm), backend.Solver(prm), backend.Solver(prm)
    rc, got = backend.Solver.solve_batch_sharded([a, b, c], [abi.WindowBuffers(w) for w in ws])          # blocks of 3, 3, 1
    assert rc == abi.OK
    for x, y in zip(ref, got):
        assert x.struct.status == y.struct.status == abi.OK
        assert np.array_equal(x.pose_Twr_out, y.pose_Twr_out) and x.outliers() == y.outliers() and x.struct.chi2_final == y.struct.chi2_final
    rc, _ = backend.Solver.solve_batch_sharded([a, a], [abi.WindowBuffers(w) for w in ws[:2]])
    assert rc == abi.ERR_BAD_ARGUMENT
    rc, got = backend.Solver.solve_batch_sharded([a, b, c], [abi.WindowBuffers(ws[0])])                    # fewer windows than handles
    assert rc == abi.OK and np.array_equal(got[0].pose_Twr_out, ref[0].pose_Twr_out)
    for s in (one, a, b, c): s.close()


def test_a_window_s_result_never_depends_on_the_size_of_its_batch(olib, monkeypatch):
    """the PCG kernel of a window must not follow the number of OTHER windows submitted with it.  A 16-window batch, its two
    sharded halves and sixteen single solves give the same bytes.  k_pcg_cu (the whole PCG in one workgroup: +7 % throughput
    at 8 C2 windows, +16 % at 16 — profiles/r04_pcg_cu_ab.log —, mat-vec sums in a different order) is opt-in (VISFS_BA_PCG_CU=1): it then equals single-window solves to
    rounding, with identical iteration counts and outlier sets."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    ws = [synth.make_window("custom", n_kf=14, n_lm=300, n_obs=2400, seed=120 + i) for i in range(16)]
    s = backend.Solver(prm)
    singles = [s.solve_window(abi.WindowBuffers(w)) for w in ws]
    got = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    halves = s.solve_batch([abi.WindowBuffers(w) for w in ws[:8]]) + s.solve_batch([abi.WindowBuffers(w) for w in ws[8:]])
    s.close()
    for (rc, a), b, c in zip(singles, got, halves):
        assert rc == b.struct.status == c.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert np.array_equal(b.pose_Twr_out, c.pose_Twr_out) and b.outliers() == c.outliers()
    monkeypatch.setenv("VISFS_BA_PCG_CU", "1")
    s = backend.Solver(prm)
    got1 = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    s.close()
    exact = 0
    for (rc, a), b in zip(singles, got1):
        assert b.struct.status == abi.OK
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run) and a.outliers() == b.outliers()
        et, er = synth.pose_errors(b.pose_Twr_out[:14], a.pose_Twr_out[:14])
        assert et < 1e-9 and er < 1e-9
        exact += int(np.array_equal(a.pose_Twr_out, b.pose_Twr_out))
    assert exact < 16                                            # a different kernel really ran

def test_describe_names_the_kernel_that_solves_the_reduced_system(olib):
    """visfs_ba_graph_info::solver_kernel: the symbol a kernel trace shows for the solver class (bench.py's roofline.kernel_symbol)."""
    from visfs_amd import backend
    # (unit_form: 2 = the fused speculative unit — windows of <= 150 k observations on their own —, 0 = the gated unit: large windows, Ceres)
    cases = [("PROD", dict(solver=2), 5, 2), ("PROD", dict(solver=0), 5, 2), ("C2", dict(solver=2), 1, 2), ("C2", dict(solver=0), 7, 2),
             ("C2", dict(solver=2, framework=1), 7, 0), ("C4", dict(solver=2), 2, 0)]
    for cfg, kw, want, form in cases:
        prm = abi.default_params(iterations=2, **kw)
        s = backend.Solver(prm)
        gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(synth.make_window(cfg)))
        s.upload(gb)
        info = s.describe()
        assert info["solver_kernel"] == want and info["unit_form"] == form, (cfg, kw, info["solver_kernel"], info["unit_form"])
        # lanes per landmark: at most 8 once the window has a thousand landmarks
        assert info["lanes_per_landmark"] <= 8 or info["n_points"] < 1024 or "VISFS_BA_GROUP" in os.environ     # (the tuning override)
        s.close()


@pytest.mark.timeout(300)
def test_shared_and_exclusive_pcg_leases_interleave_without_deadlock(olib):
    """One thread keeps solving a 12-window batch (two concurrent parts of one-wave PCG grids: shared leases on the device budget), another
    a C4-size window (four-wave PCG: exclusive lease), on two handles of the same device.  Everything must finish and every result must
    equal its reference — the budget admits the small grids together and makes the big one wait for all of them (and vice versa)."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    small = [synth.make_window("custom", n_kf=20, n_lm=400, n_obs=3200, seed=200 + i) for i in range(12)]
    sa, sb = backend.Solver(prm), backend.Solver(prm)
    ref_small = sa.solve_batch([abi.WindowBuffers(w) for w in small])
    big = synth.make_window("C4")
    gb, *_ = abi.pack_window_with(sb.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(big))
    sb.upload(gb); rc, _ = sb.optimize(); assert rc == abi.OK
    ref_big = sb.download()
    out = {}

    def run_small():
        res = []
        for _ in range(4):
            res.append(sa.solve_batch([abi.WindowBuffers(w) for w in small]))
        out["small"] = res

    def run_big():
        res = []
        for _ in range(4):
            sb.reset(); rc, _ = sb.optimize(); res.append((rc, sb.download()))
        out["big"] = res

    th = [threading.Thread(target=run_small), threading.Thread(target=run_big)]
    for t in th: t.start()
    for t in th: t.join()
    for res in out["small"]:
        for a, b in zip(ref_small, res):
            assert a.struct.status == b.struct.status == abi.OK and np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
    for rc, d in out["big"]:
        assert rc == abi.OK and all(np.array_equal(x, y) for x, y in zip(d, ref_big))
    sa.close(); sb.close()


# ---------------------------------------------------------------- banded direct solver (k_band_chol)
def _direct_trial(olib, w, env, lam_scale=1e-3, **prm_kw):
    """One damped direct solve of window `w` under the environment overrides `env`: (oracle dx, GPU dx, GPU trial chi2, solver kernel)."""
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        o, s, gb = make_pair(olib, w, **prm_kw)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    chi, md = o.linearize(); s.linearize()
    lam = lam_scale * md
    ot, gt = o.trial(lam), s.trial(lam)
    assert ot[3] == gt[3] == 1
    out = (o.fetch(abi.BUF_DX_POSE), s.fetch(abi.BUF_DX_POSE), s.fetch(abi.BUF_POSE_TRIAL), gt[0])
    s.close(); o.close()
    return out


def test_banded_cholesky_matches_dense_fallback_and_oracle(olib):
    """Optimizer/Solver=0 at C2: S is block-banded (49 block rows, half-bandwidth 9).  The banded Cholesky in one workgroup
    (whole band in LDS), its streaming form (a 12-row window, the factor through HBM — what C4 runs) and the dense blocked
    Cholesky (VISFS_BA_BAND=0) solve the same system: the two banded forms bit for bit, all three with the oracle to 1e-9."""
    w = synth.make_window("C2")
    ox, band, pose_b, chi_b = _direct_trial(olib, w, {"VISFS_BA_BAND": "1"}, iterations=20, solver=0)
    _, stream, pose_s, chi_s = _direct_trial(olib, w, {"VISFS_BA_BAND": "1", "VISFS_BA_BAND_ROWS": "12"}, iterations=20, solver=0)
    _, dense, pose_d, chi_d = _direct_trial(olib, w, {"VISFS_BA_BAND": "0"}, iterations=20, solver=0)
    assert np.array_equal(band, stream) and np.array_equal(pose_b, pose_s) and chi_b == chi_s
    assert rel_err(band, ox) < 1e-9 and rel_err(dense, ox) < 1e-9 and rel_err(band, dense) < 1e-9
    assert abs(chi_b - chi_d) <= 1e-9 * abs(chi_d)


def test_banded_cholesky_on_odd_shapes(olib):
    """Ragged tracks with odometry (12 poses: the band is almost the whole matrix), a hard start, a window whose newest poses see
    no landmark (pinned identity blocks inside the band), and the tail handling of a band wider than the rows that are left."""
    from helpers import hard_window, ragged_window
    for w, kw in ((ragged_window(seed=7), {}), (hard_window(seed=5), {}), (synth.make_window("C3", n_kf=24, n_lm=400, n_obs=3200), {})):
        ox, gx, _, _ = _direct_trial(olib, w, {"VISFS_BA_BAND": "1", "VISFS_BA_SMALL_SOLVE": "0"}, iterations=10, solver=0, **kw)
        assert rel_err(gx, ox) < 1e-9
    # tracks of 16 key-frames: block half-bandwidth 15 — the trailing update runs several rounds of tiles per thread and the backward
    # substitution takes the LDS form (more than ten blocks per row do not fit one wavefront's register window); resident and streaming
    w = synth.make_window("custom", n_kf=30, n_lm=200, n_obs=3200, seed=12)
    for env in ({"VISFS_BA_BAND": "1"}, {"VISFS_BA_BAND": "1", "VISFS_BA_BAND_ROWS": "19"}):
        ox, gx, _, _ = _direct_trial(olib, w, env, iterations=10, solver=0)
        assert rel_err(gx, ox) < 1e-9
    # a streaming window that is shorter than the band allows at the bottom of the matrix
    w = synth.make_window("custom", n_kf=40, n_lm=600, n_obs=6000, seed=11)
    ox, gx, _, _ = _direct_trial(olib, w, {"VISFS_BA_BAND": "1", "VISFS_BA_BAND_ROWS": "12"}, iterations=10, solver=0)
    assert rel_err(gx, ox) < 1e-9


def test_banded_cholesky_reports_a_failed_factorisation(olib):
    """A non-positive pivot (undamped Gauss-Newton on a window whose gauge is free: no fixed landmark, lambda = 0) must reject the trial
    (g2o: solver returns false), not produce numbers: solver_ok = 0 on both sides or an identical finite solve."""
    w = synth.make_window("custom", n_kf=16, n_lm=200, n_obs=1600, seed=3, fixed_frac=0.0)
    o, s, gb = make_pair(olib, w, iterations=10, solver=0)
    o.linearize(); s.linearize()
    ot, gt = o.trial(-1e12), s.trial(-1e12)             # H - 1e12 I: indefinite by construction
    assert ot[3] == 0 and gt[3] == 0
    s.close(); o.close()


def test_c4r_parity_with_the_converged_class_spelled_out(olib):
    """C4R = BASELINE config 4 (200 KF / 30k landmarks / 300k observations) started from 1e-3 rad of rotation error: phase 1 converges and
    the kernels work on all 300k edges.  VERDICT r02: this is the workload of the "converged" class — at machine precision the LM
    accept / reject decision depends on the summation order, so GPU and oracle may stop ONE iteration apart.  Asserted explicitly:
    every stage buffer of the first linearisation and of two damped solves agrees to 1e-9; the LM traces agree trial for trial up
    to the point where the runs split (if they do); a split is only allowed where chi2 has converged to 1e-9 relative and the
    iteration counts differ by at most one; final poses agree to 1e-6, outlier sets to a handful of borderline edges."""
    from test_gpu_parity import check_stages
    o, s, gb = make_pair(olib, synth.make_window("C4R"), iterations=20, solver=2)
    check_stages(o, s, lambdas=(None, 1.0))
    o.reset(); s.reset()
    rc_o, st_o, _ = o.optimize()
    rc_g, st_g = s.optimize()
    assert rc_o == rc_g == abi.OK
    lam_o = np.array([st_o.trace_lambda[i] for i in range(st_o.n_trace)]); lam_g = np.array([st_g.trace_lambda[i] for i in range(st_g.n_trace)])
    chi_o = np.array([st_o.trace_chi2[i] for i in range(st_o.n_trace)]); chi_g = np.array([st_g.trace_chi2[i] for i in range(st_g.n_trace)])
    n = min(len(lam_o), len(lam_g))
    split = next((i for i in range(n) if abs(lam_o[i] - lam_g[i]) > 1e-6 * abs(lam_o[i]) or abs(chi_o[i] - chi_g[i]) > 1e-7 * abs(chi_o[i])), n)
    assert list(st_o.iterations_run)[0] == list(st_g.iterations_run)[0], "phase 1 must run identically"
    assert abs(st_o.chi2_phase1 - st_g.chi2_phase1) <= 1e-7 * abs(st_o.chi2_phase1) and st_o.n_outliers == st_g.n_outliers
    if split < n or len(lam_o) != len(lam_g):
        # the converged class: the traces part only where the iteration has stalled at machine precision
        assert abs(st_o.iterations_run[1] - st_g.iterations_run[1]) <= 1
        k = max(split - 1, 0)
        assert abs(chi_o[k] - chi_g[k]) <= 1e-9 * abs(chi_o[k])
        assert abs(st_o.chi2_final - st_g.chi2_final) <= 1e-9 * abs(st_o.chi2_final)
    else:
        assert list(st_o.iterations_run) == list(st_g.iterations_run) and list(st_o.trials_run) == list(st_g.trials_run)
    po, pto, outo, _ = o.download(); pg, ptg, outg, _ = s.download()
    assert np.array_equal(outo, outg)
    assert rel_err(pg, po) < 1e-6 and rel_err(ptg, pto) < 1e-6
    s.close(); o.close()


def test_reference_default_solver_in_batched_launches(olib):
    """Optimizer/Solver=0 (the reference default, Parameters.h:185) through visfs_ba_solve_batch: windows whose reduced system is banded
    share every launch (k_band_chol<Many>: one workgroup per window) instead of being solved one after another — each result equal
    to the single-window solve bit for bit; a window with a wide band (every pose sees every landmark: dense S) is solved on its own."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=0)
    ws = [synth.make_window("custom", n_kf=24, n_lm=500, n_obs=4000, seed=300 + i) for i in range(6)]
    ws.append(synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11))
    ws += [synth.make_window("PROD", window_index=i) for i in range(2)]                   # k_small_solve group
    s = backend.Solver(prm)
    singles = []
    for w in ws:
        wb = abi.WindowBuffers(w)
        rc, rb = s.solve_window(wb)
        singles.append((rc, rb, wb))
        if len(w["pose_ids"]) > 11:
            assert s.describe()["solver_kernel"] == 7
    wbs = [abi.WindowBuffers(w) for w in ws]
    rbs = s.solve_batch(wbs)
    for (rc, a, wa), b, wb in zip(singles, rbs, wbs):
        assert rc == abi.OK and b.struct.status == rc
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert np.array_equal(wa.point_xyz, wb.point_xyz, equal_nan=True)
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run)
    s.close()
    # and against the oracle for one of them
    o, g, gb = make_pair(olib, ws[0], iterations=10, solver=0)
    check_optimize(o, g)
    g.close(); o.close()


@pytest.mark.parametrize("solver", [2, 0])
def test_batch_members_on_the_fused_speculative_unit_equal_the_gated_unit(olib, monkeypatch, solver):
    """Round 4: the windows of a batched launch can run the fused speculative unit of a lone window (k_linearize only
    at the first iteration of a phase, the pose-major role behind the Schur gather, back-substitution + LM decision + landmark-major role
    in one launch: 4 launches per iteration instead of 6).  It performs the arithmetic of the gated unit: every window of a mixed batch
    — a window that rejects trials, odometry, k_small_solve members, a refused window — must come out bit for bit the same in both forms
    (the gated unit stays the default for batch members: profiles/r04_batch_fused_unit_ab.log)."""
    from visfs_amd import backend
    from helpers import hard_window
    prm = abi.default_params(iterations=10, solver=solver)
    ws = [synth.make_window("C1", window_index=i) for i in range(3)]
    ws += [synth.make_window("PROD", window_index=i) for i in range(3)]
    ws.append(hard_window())
    ws.append(synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11))
    ws.append(synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400))
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VISFS_BA_BATCH_SPEC", mode)
        s = backend.Solver(prm)
        wbs = [abi.WindowBuffers(w) for w in ws]
        rbs = s.solve_batch(wbs)
        got[mode] = (rbs, wbs)
        s.close()
    for a, b, wa, wb in zip(got["0"][0], got["1"][0], got["0"][1], got["1"][1]):
        assert a.struct.status == b.struct.status == abi.OK
        assert np.array_equal(a.pose_Twr_out, b.pose_Twr_out) and a.outliers() == b.outliers()
        assert np.array_equal(wa.point_xyz, wb.point_xyz, equal_nan=True)
        assert list(a.struct.iterations_run) == list(b.struct.iterations_run) and a.struct.chi2_final == b.struct.chi2_final


def test_c5_share_on_the_fused_unit_is_bit_identical_to_the_gated_unit(olib, monkeypatch):
    # the per-GPU workload of BASELINE config 5 (resident C2-size windows, GRAPH layer) in both unit forms
    from visfs_amd import backend
    prm = abi.default_params(iterations=20, solver=2)
    gbs = [abi.pack_window_with(backend.load_library().visfs_ba_pack_window, prm, abi.WindowBuffers(synth.make_window("C5", window_index=i)))[0] for i in range(4)]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("VISFS_BA_BATCH_SPEC", mode)
        s = backend.Solver(prm)
        s.batch_upload(gbs); s.batch_reset()
        rc, stats = s.batch_optimize()
        assert rc == abi.OK
        outs[mode] = ([s.batch_download(i) for i in range(4)], [(list(st.iterations_run), list(st.trials_run), st.pcg_iterations, st.chi2_final) for st in stats])
        s.close()
    assert outs["0"][1] == outs["1"][1]
    for a, b in zip(outs["0"][0], outs["1"][0]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_a_failed_upload_leaves_no_graph_resident(olib):
    """ws_upload overwrites the primary device arrays before the graph is validated — after a REJECTED upload the handle must
    not keep the previous graph marked resident (its index structures over the rejected graph's arrays would be walked by the kernels)."""
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=2)
    lib = backend.load_library()
    s = backend.Solver(prm)
    gb = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(synth.make_window("C1")))[0]
    s.upload(gb)
    rc, _ = s.optimize()
    assert rc == abi.OK
    bad = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(synth.make_window("C1", window_index=1)))[0]
    bad.obs_pose[[0, 1]] = bad.obs_pose[[1, 0]]                      # observations no longer sorted by (point, pose)
    rc_bad = s.lib.visfs_ba_graph_upload(s.h, C.byref(bad.struct))
    assert rc_bad == abi.ERR_BAD_ARGUMENT
    rc2 = s.lib.visfs_ba_optimize(s.h, None)
    assert rc2 == abi.ERR_NOT_LOADED
    s.upload(gb)                                                     # the handle stays usable
    rc3, _ = s.optimize()
    assert rc3 == abi.OK
    s.close()


@pytest.mark.parametrize("seed", [559, 764, 873, 1136, 1242])
def test_banded_solver_keeps_the_checker_s_accuracy_on_ill_conditioned_systems(olib, seed):
    """VERDICT r03 item 3.  Windows without enough landmarks on some pose (random cases whose reduced system has cond(S) of 1e8 at
    lambda = 1e-2 and 1e11 at 1e-5): the reference-default solver is a backward-stable Cholesky (Parameters.h:185), and production windows
    have no fixed pose (Estimator.cpp:252), so S is routinely this badly conditioned.  k_band_chol must solve ITS system as accurately as
    the checker's scalar Cholesky solves its own — measured against numpy.linalg.solve on each side's S and b_s; round 3's closed-form
    pivot inverses were 1e2 (lambda = 1e-2) to 4e4 (1e-5) times worse (profiles/r03_stage_precision.log, profiles/r04_band_pivot_study.log)."""
    import test_gpu_random as T
    w, kw = T.random_case(seed)
    o, s, gb = make_pair(olib, w, iterations=10, solver=0, robust_kernel_delta=kw["robust_kernel_delta"])
    assert s.describe()["solver_kernel"] == 7                                       # k_band_chol
    o.linearize(); s.linearize()
    n6 = 6 * o.npf
    for lam in (1e-2, 1e-5):
        ot, gt = o.trial(lam), s.trial(lam)
        assert ot[3] == gt[3] == 1                                                  # both factorisations succeed
        So = o.fetch(abi.BUF_S).reshape(n6, n6); Sg = s.fetch(abi.BUF_S).reshape(n6, n6)
        xo, xg = o.fetch(abi.BUF_DX_POSE), s.fetch(abi.BUF_DX_POSE)
        eo = rel_err(xo, np.linalg.solve(So, o.fetch(abi.BUF_BS)))
        eg = rel_err(xg, np.linalg.solve(Sg, s.fetch(abi.BUF_BS)))
        assert eg <= 10.0 * max(eo, 1e-13), (seed, lam, eo, eg)
    s.close(); o.close()


def _fault_free_direct_reference(w, iterations=10):
    from visfs_amd import backend
    s0 = backend.Solver(abi.default_params(iterations=iterations, solver=0))
    wb0 = abi.WindowBuffers(w)
    rc0, rb0 = s0.solve_window(wb0)
    s0.close()
    return rc0, rb0, wb0


@pytest.mark.parametrize("shape", ["one_wave", "four_wave"])
def test_a_timed_out_persistent_pcg_is_solved_again_on_the_direct_solver(olib, monkeypatch, shape):
    """the reference's linear solver cannot fail for lack of residency (Optimizer.cpp:76-91); Optimizer/Solver=2's
    persistent PCG can, when another process keeps part of the GPU busy.  The hand-off time-out is forced (VISFS_BA_FAULT_PCG_TIMEOUT=1:
    block row 0 withholds its first hand-off, every wait gives up after 2^10 polls): the call must still return a solution — the window
    re-run from its uploaded estimates on the direct solver, i.e. exactly what an Optimizer/Solver=0 handle computes — and say so."""
    from visfs_amd import backend
    w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11) if shape == "one_wave" else synth.make_window("custom", n_kf=100, n_lm=3000, n_obs=24000, seed=21)
    rc0, rb0, wb0 = _fault_free_direct_reference(w)
    assert rc0 == abi.OK and rb0.struct.solver_fallback == 0
    monkeypatch.setenv("VISFS_BA_FAULT_PCG_TIMEOUT", "1")
    prm = abi.default_params(iterations=10, solver=2)
    s = backend.Solver(prm)
    wb = abi.WindowBuffers(w)
    rc, rb = s.solve_window(wb)
    assert rc == abi.OK and rb.struct.solver_fallback == 1
    assert np.array_equal(rb.pose_Twr_out, rb0.pose_Twr_out) and rb.outliers() == rb0.outliers()
    assert np.array_equal(wb.point_xyz, wb0.point_xyz, equal_nan=True)
    # GRAPH layer: the first optimise after an upload (or a reset) can be re-run; one that continues from earlier results cannot
    gb = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))[0]
    s.upload(gb)
    rc1, st1 = s.optimize()
    assert rc1 == abi.OK and st1.solver_fallback == 1 and st1.pcg_iterations == 0
    st2 = abi.Stats()
    rc2 = s.lib.visfs_ba_optimize(s.h, C.byref(st2))
    assert rc2 == abi.ERR_DEVICE                                     # not from the uploaded estimates: nothing to re-run from
    s.reset()
    rc3, st3 = s.optimize()
    assert rc3 == abi.OK and st3.solver_fallback == 1
    s.close()


def test_timed_out_members_of_a_batch_fall_back_one_by_one(olib, monkeypatch):
    from visfs_amd import backend
    ws = [synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11 + i) for i in range(3)]
    refs = [_fault_free_direct_reference(w) for w in ws]
    monkeypatch.setenv("VISFS_BA_FAULT_PCG_TIMEOUT", "1")
    s = backend.Solver(abi.default_params(iterations=10, solver=2))
    rbs = s.solve_batch([abi.WindowBuffers(w) for w in ws])
    for rb, (rc0, rb0, _) in zip(rbs, refs):
        assert rb.struct.status == rc0 == abi.OK and rb.struct.solver_fallback == 1
        assert np.array_equal(rb.pose_Twr_out, rb0.pose_Twr_out) and rb.outliers() == rb0.outliers()
    s.close()


def _solve_graph(monkeypatch, w, env, **prm_kw):
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


@pytest.mark.parametrize("case", ["K30", "C2", "C3", "RAGGED30", "HARD", "S0", "CERES", "LASER"])
def test_schur_by_runs_of_landmarks_equals_the_pair_gather_to_rounding(olib, monkeypatch, case):
    """Round 4: k_schur_runs stages every tile core, Q = N D and camera-frame point in LDS once per damped solve and
    accumulates the blocks of a run of landmarks in registers; the pair-list gather rebuilds both tiles and the landmark inverse per pair.
    Per pair the operands are the same values — only the association of the sums differs: identical LM trajectories (iteration and PCG
    counts, outlier sets), results equal to rounding."""
    from helpers import hard_window, ragged_window
    kw = dict(iterations=10, solver=2)
    if case == "K30":
        w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11)
    elif case == "RAGGED30":
        w = ragged_window(seed=9, n_kf=30, n_lm=900, n_obs=7200, odo=True, drop=0.3)
    elif case == "HARD":
        w = hard_window()
    elif case == "S0":
        w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11); kw["solver"] = 0
    elif case == "CERES":
        w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11); kw["framework"] = 1
    elif case == "LASER":
        w = synth.make_laser_window(n_kf=16, with_visual=True, n_points=400)
    else:
        w = synth.make_window(case); kw["iterations"] = 20
    i0, rc0, st0, out0 = _solve_graph(monkeypatch, w, dict(VISFS_BA_SCHUR_RUNS="0"), **kw)
    i1, rc1, st1, out1 = _solve_graph(monkeypatch, w, dict(VISFS_BA_SCHUR_RUNS="1"), **kw)
    assert i0["schur_runs"] == 0 and i0["n_schur_chunks"] > 0
    assert i1["schur_runs"] > 0 and i1["n_schur_chunks"] == 0
    assert rc0 == rc1 == abi.OK
    assert list(st0.iterations_run) == list(st1.iterations_run) and list(st0.trials_run) == list(st1.trials_run)
    assert st0.pcg_iterations == st1.pcg_iterations and st0.n_outliers == st1.n_outliers
    assert abs(st0.chi2_final - st1.chi2_final) <= 1e-10 * st0.chi2_final
    assert np.abs(out0[0] - out1[0]).max() < 1e-10 and rel_err(out1[1], out0[1]) < 1e-9
    assert np.array_equal(out0[2], out1[2])


def test_windows_the_run_kernel_does_not_fit_keep_the_gather(olib, monkeypatch):
    """Landmark ids that do not grow with time (here: reversed against the tracks' first key-frames AND interleaved) give runs whose
    pose span exceeds the kernel's 64 poses; small reduced systems are k_small_solve's.  Both keep the pair-list gather; parity unchanged."""
    w = synth.make_window("custom", n_kf=160, n_lm=3000, n_obs=24000, seed=21)
    # interleave the landmarks of the two halves of the trajectory: every run of consecutive landmarks now spans the whole window
    ids = np.asarray(w["point_ids"]).copy()
    n = len(ids)
    perm = np.empty(n, np.int64); perm[0::2] = np.arange((n + 1) // 2); perm[1::2] = np.arange((n + 1) // 2, n)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)                      # old landmark index -> new position
    w2 = dict(w)
    w2["point_xyz"] = np.asarray(w["point_xyz"])[perm].copy(); w2["point_fixed"] = np.asarray(w["point_fixed"])[perm].copy()
    feat_old = np.searchsorted(ids, np.asarray(w["ref_feature"]))
    new_feat = ids[inv[feat_old]]
    order = np.lexsort((np.asarray(w["ref_pose"]), new_feat))
    for k in ("ref_pose", "ref_u", "ref_v", "ref_depth"):
        w2[k] = np.asarray(w[k])[order]
    w2["ref_feature"] = new_feat[order]
    if "gross" in w2:
        w2["gross"] = np.asarray(w["gross"])[order]
    monkeypatch.setenv("VISFS_BA_SCHUR_RUNS", "1")                              # (opt-in: the gather is the default, profiles/r04_schur_lds_tiles.log)
    o, s, gb = make_pair(olib, w2, iterations=10, solver=2)
    d = s.describe()
    assert d["schur_runs"] == 0 and d["n_schur_chunks"] > 0
    check_optimize(o, s, pose_tol=1e-9)
    s.close(); o.close()
    o, s, gb = make_pair(olib, synth.make_window("PROD"), iterations=10, solver=2)
    assert s.describe()["schur_runs"] == 0
    s.close(); o.close()
    o, s, gb = make_pair(olib, w, iterations=10, solver=2)                       # the same window with its ids in time order: runs
    assert s.describe()["schur_runs"] > 0
    check_optimize(o, s, pose_tol=1e-9)
    s.close(); o.close()


@pytest.mark.parametrize("case", ["C1", "PROD", "K30", "K30_S0", "K30_CERES", "C2"])
def test_per_frame_launch_sequence_is_replayed_across_uploads_and_changes_no_byte(olib, monkeypatch, case):
    """Round 4: visfs_ba_solve_window runs its launches through the kernels that read the window from a fixed device
    address, on grids rounded up to size classes; the sequence captured at the second frame of a class is REPLAYED for the following frames
    although every frame uploads a new window (other measurements, a few references more or less).  Every frame's result must be the bytes
    of the by-value path (VISFS_BA_FRAME_GRAPH=0), and the later frames must really have been replays."""
    from visfs_amd import backend
    from helpers import drop_refs
    kw = dict(iterations=10, solver=2)
    if case == "K30_S0":
        kw["solver"] = 0
    if case == "K30_CERES":
        kw["framework"] = 1
    if case == "C2":
        kw["iterations"] = 20

    def frame(i):
        if case in ("C1", "PROD", "C2"):
            w = synth.make_window(case, window_index=i)
        else:
            w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=100 + i)
        if i % 3 == 1:                                       # a few references fewer: other sizes, (mostly) the same class
            rng = np.random.default_rng(i)
            w = drop_refs(w, rng.random(len(w["ref_feature"])) > 0.01)
        return w
    n_frames = 7
    outs = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("VISFS_BA_FRAME_GRAPH", mode)
        s = backend.Solver(abi.default_params(**kw))
        res = []
        for i in range(n_frames):
            wb = abi.WindowBuffers(frame(i))
            rc, rb = s.solve_window(wb)
            res.append((rc, rb.pose_Twr_out.copy(), rb.outliers(), wb.point_xyz.copy(), list(rb.struct.iterations_run), rb.struct.chi2_final, s.describe()["graph_replayed"]))
        outs[mode] = res
        s.close()
    for a, b in zip(outs["0"], outs["2"]):
        assert a[0] == b[0] == abi.OK
        assert np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3], equal_nan=True) and a[4] == b[4] and a[5] == b[5]
    assert not any(r[6] for r in outs["0"])                  # the by-value path never replays a per-frame call
    assert sum(r[6] for r in outs["2"]) >= n_frames - 4      # two frames at most per class before the replays start (two classes show up)


@pytest.mark.parametrize("case", ["K30", "C3S", "HARD"])
def test_finalisation_on_board_the_pcg_launch_changes_no_byte(olib, monkeypatch, case):
    """VERDICT r03 item 9 (closed by measurement, profiles/r04_fin_pcg_fusion_ab.log): k_schur_finalize as the prologue of the one-wave PCG
    launch (VISFS_BA_FIN_PCG=1: write-through stores + drain + flag, sc1 reloads, unit-tagged hand-off words) performs the same arithmetic:
    every output, counter and trace entry equals the two-launch form, rejected trials and odometry included — also across repeated solves
    of one resident graph (the tags count the units of a call, k_reset zeroes the flags)."""
    from helpers import hard_window
    from test_gpu_parity import _stats_tuple
    w = synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11) if case == "K30" else hard_window() if case == "HARD" else synth.make_window("C3", n_kf=20, n_lm=500, n_obs=4000)
    kw = dict(iterations=20, solver=2)
    _, rc0, st0, out0 = _solve_graph(monkeypatch, w, dict(VISFS_BA_FIN_PCG="0"), **kw)
    _, rc1, st1, out1 = _solve_graph(monkeypatch, w, dict(VISFS_BA_FIN_PCG="1"), **kw)
    assert rc0 == rc1 == abi.OK
    assert _stats_tuple(st0) == _stats_tuple(st1)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out0, out1))
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


@pytest.mark.parametrize("case", ["K30", "K30_S0", "C2", "HARD"])
def test_finalisation_by_the_last_wave_to_arrive_changes_no_byte(olib, monkeypatch, case):
    """VERDICT r03 item 9 exactly as asked (closed by measurement, profiles/r04_fin_arrive_ab.log): the wavefront of k_schur_partial that
    completes a block's partials — gather chunks and, for diagonal blocks, the pose-major chunks that ran — finalises the block on the spot
    (VISFS_BA_FIN_ARRIVE=1: write-through partials, drain, per-block arrival counter, coherent reloads, sums in index order) and
    k_schur_finalize is not launched.  Every output, counter and trace entry equals the two-launch form, rejected trials included, PCG and
    direct solver, also across repeated solves of one resident graph (the last arriver leaves the counter at zero)."""
    from helpers import hard_window
    from test_gpu_parity import _stats_tuple
    w = (synth.make_window("C2") if case == "C2" else hard_window() if case == "HARD" else synth.make_window("custom", n_kf=30, n_lm=800, n_obs=8000, seed=11))
    kw = dict(iterations=20, solver=0 if case == "K30_S0" else 2)
    _, rc0, st0, out0 = _solve_graph(monkeypatch, w, dict(VISFS_BA_FIN_ARRIVE="0"), **kw)
    _, rc1, st1, out1 = _solve_graph(monkeypatch, w, dict(VISFS_BA_FIN_ARRIVE="1"), **kw)
    assert rc0 == rc1 == abi.OK
    assert _stats_tuple(st0) == _stats_tuple(st1)
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out0, out1))
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

