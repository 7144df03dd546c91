"""Measurement sweep over the BASELINE configs on one MI355X: resident-graph optimise rate, end-to-end
localOptimize-equivalent (host buffers in/out: pack + H2D + solve + D2H), CPU oracle rate, parity."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib
from helpers import twr_of
from visfs_amd import abi, backend, synth

olib = oracle_lib.load()
lib = backend.load_library()
rows = []
for cfg, solver, iters in (("C1", 2, 20), ("PROD", 2, 10), ("C2", 2, 20), ("C2", 0, 20), ("C3", 2, 20), ("C4", 2, 20)):
    prm = abi.default_params(iterations=iters, solver=solver)
    w = synth.make_window(cfg)
    wb = abi.WindowBuffers(w)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, wb)
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(3):
        s.reset(); s.optimize()
    ts = []
    for _ in range(20 if cfg != "C4" else 8):
        s.reset(); t0 = time.perf_counter(); rc, st = s.optimize(); ts.append(time.perf_counter() - t0)
    its = st.iterations_run[0] + st.iterations_run[1]
    pg, ptg, outg, _ = s.download()
    info = s.describe()
    # end to end through the window layer (host pointers in, host pointers out)
    te = []
    for _ in range(8 if cfg != "C4" else 3):
        wbe = abi.WindowBuffers(w); t0 = time.perf_counter(); rc2, rb = s.solve_window(wbe); te.append(time.perf_counter() - t0)
    # oracle
    o = oracle_lib.OracleSystem(olib, prm, gb, 1)
    to = []
    for _ in range(3 if cfg != "C4" else 1):
        o.reset(); rco, sto, sec = o.optimize(); to.append(sec)
    po, pto, outo, _ = o.download(); o.close()
    et, er = __import__("visfs_amd.synth", fromlist=["x"]).pose_errors(twr_of(olib.oracle_unpack_pose, pg, w["Trc"]), twr_of(olib.oracle_unpack_pose, po, w["Trc"]))
    rows.append(dict(cfg=cfg, solver=solver, iters=its, trials=st.trials_run[0] + st.trials_run[1], pcg=st.pcg_iterations,
                     gpu_ms=1e3 * float(np.median(ts)), gpu_it_s=its / float(np.median(ts)),
                     e2e_ms=1e3 * float(np.median(te)), e2e_it_s=its / float(np.median(te)),
                     cpu_ms=1e3 * float(np.median(to)), cpu_it_s=its / float(np.median(to)),
                     speedup=float(np.median(to)) / float(np.median(ts)), pose_err_t=et, pose_err_r=er,
                     outliers_equal=bool(np.array_equal(outo, outg)), n_blk=info["n_blk"], n_pairs=info["n_pairs"],
                     device_mb=info["device_bytes"] / 1e6))
    print(json.dumps(rows[-1]), flush=True)
    s.close()

# config 5 in miniature on one GPU: 8 independent C2 windows through visfs_ba_solve_batch (concurrent streams)
prm = abi.default_params(iterations=20, solver=2)
ws = [synth.make_window("C5", window_index=i) for i in range(8)]
s = backend.Solver(prm)
wbs = [abi.WindowBuffers(w) for w in ws]
s.solve_batch(wbs)
t0 = time.perf_counter(); rbs = s.solve_batch([abi.WindowBuffers(w) for w in ws]); dt = time.perf_counter() - t0
its = sum(r.struct.iterations_run[0] + r.struct.iterations_run[1] for r in rbs)
print(json.dumps(dict(cfg="C5x8 on one GPU (solve_batch, end to end incl. pack/H2D/D2H)", windows=8, iters=its, ms=1e3 * dt, it_s=its / dt)))
s.close()
