"""World-size-2 gloo test of the multi-GPU path on CPU (BASELINE config 5): independent windows are sharded over
ranks with NO data-path collective; only the bench contract's barrier / max-over-ranks / result gather use the
process group.  The per-window solver here is the CPU oracle standing in for the GPU (tests may use it)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_windows_partitions_exactly():
    from visfs_amd import dist as vd
    for n, world in ((64, 8), (64, 4), (7, 2), (1, 2), (5, 8)):
        got = [vd.shard_windows(n, r, world) for r in range(world)]
        flat = [w for s in got for w in s]
        assert flat == list(range(n))                      # contiguous blocks, every window exactly once
        assert max(len(s) for s in got) - min(len(s) for s in got) <= (n + world - 1) // world


def _solve_rows(window_ids):
    """One row per window: [id, iterations, chi2_final, n_outliers, first pose tx] from the CPU oracle."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from visfs_amd import abi, synth
    olib = oracle_lib.load()
    prm = abi.default_params(iterations=10, solver=2)
    rows = []
    for wi in window_ids:
        w = synth.make_window("PROD", window_index=wi)
        wb = abi.WindowBuffers(w); rb = abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs)
        rc = olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1)
        assert rc == abi.OK
        rows.append([wi, rb.struct.iterations_run[0] + rb.struct.iterations_run[1], rb.struct.chi2_final,
                     rb.struct.n_outliers, rb.pose_Twr_out[0, 3]])
    return torch.tensor(rows, dtype=torch.float64)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from visfs_amd import dist as vd
    vd.init_process_group("gloo", rank, world)
    mine = vd.shard_windows(6, rank, world)
    vd.barrier(world)
    rows = _solve_rows(mine)
    elapsed = 1.0 + rank                                     # fake per-rank wall time: the contract takes the MAX
    tmax = vd.reduce_max(elapsed, world)
    total_iters = vd.reduce_sum(float(rows[:, 1].sum()), world)
    allrows = vd.gather_results(rows, world)
    vd.barrier(world)
    q.put((rank, tmax, total_iters, allrows.numpy()))


def test_two_rank_gloo_run_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _solve_rows(range(6)).numpy()
    for rank, tmax, total_iters, allrows in got:
        assert tmax == 2.0                                   # max over ranks
        assert total_iters == ref[:, 1].sum()
        assert np.array_equal(allrows, ref)                  # gathered results == single-process results, in window order
