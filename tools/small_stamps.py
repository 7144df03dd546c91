"""Diagnostic: stage timeline of the first units of k_small_optimize (needs libvisfs_ba_hip_stamps.so built with -DVISFS_BA_STAMPS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
NAMES = ["stage poses", "landmark pass", "pose pass+odo", "hpp/laser/lambda", "schur chunks", "schur blocks", "dense assemble", "solve", "oplus+stage", "backsub", "odo/laser/reduce", "decide"]
for cfg, solver in (("PROD", 2), ("PROD", 0), ("C1", 2)):
    w = synth.make_window(cfg); prm = abi.default_params(iterations=10, solver=solver)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(3):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    st = out.view(np.uint64).astype(np.int64)
    for u in range(3):
        t = st[16 * u: 16 * u + 12]
        print(cfg, "solver", solver, "unit", u, " ".join(f"{NAMES[i + 1]}={(t[i + 1] - t[i]) * 10}ns" for i in range(11)), "| unit total", (t[11] - t[0]) * 10, "ns")
    s.close()
