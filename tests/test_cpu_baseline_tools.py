"""bench.py's cpu_baseline leg: the AVX-512 / FMA build of the checker that is only ever TIMED must still compute the
same optimisation as the checker proper, the OpenMP team can be bound one thread per CPU, and the CPU list bench.py picks stays inside the
process's affinity mask and is packed by last-level-cache domain."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib
from visfs_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve(lib, threads=1):
    prm = abi.default_params(iterations=10, solver=2)
    gb, *_ = abi.pack_window_with(lib.oracle_pack_window, prm, abi.WindowBuffers(synth.make_window("C1")))
    s = oracle_lib.OracleSystem(lib, prm, gb, threads)
    rc, st, _ = s.optimize()
    out = s.download()
    s.close()
    return rc, st, out


@pytest.mark.skipif(not oracle_lib.cpu_has_avx512(), reason="host CPU without AVX-512")
def test_the_timed_avx512_build_computes_the_checker_s_optimisation():
    rc3, st3, out3 = _solve(oracle_lib.load())
    rc4, st4, out4 = _solve(oracle_lib.load(v4=True))
    assert rc3 == rc4 == abi.OK
    assert list(st3.iterations_run) == list(st4.iterations_run) and list(st3.trials_run) == list(st4.trials_run) and st3.n_outliers == st4.n_outliers
    assert abs(st3.chi2_final - st4.chi2_final) <= 1e-9 * st3.chi2_final
    assert np.abs(out3[0] - out4[0]).max() < 1e-9 and np.array_equal(out3[2], out4[2])       # contraction moves last bits only


def test_openmp_team_can_be_bound_and_the_caller_gets_its_mask_back():
    lib = oracle_lib.load(omp=True)
    before = os.sched_getaffinity(0)
    cpus = sorted(before)[:2]
    n = lib.oracle_omp_pin((C.c_int32 * len(cpus))(*cpus), len(cpus))
    assert n == len(cpus)
    assert os.sched_getaffinity(0) == {cpus[0]}                     # the calling thread is thread 0 of the team
    lib.oracle_omp_unpin()
    assert os.sched_getaffinity(0) == before
    rc, st, _ = _solve(lib, threads=len(cpus))                      # the bound team still computes
    assert rc == abi.OK
    assert oracle_lib.load().oracle_omp_pin((C.c_int32 * 1)(cpus[0]), 1) == 0      # serial build: nothing to bind


def test_bench_picks_cpus_inside_the_mask_packed_by_cache_domain():
    sys.path.insert(0, ROOT)
    import bench
    allowed = os.sched_getaffinity(0)
    for n in (1, 2, len(allowed)):
        pick, ndom = bench._cpu_topology_pick(n)
        assert len(pick) == min(n, len(allowed)) and len(set(pick)) == len(pick) and set(pick) <= allowed and ndom >= 1
