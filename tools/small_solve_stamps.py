"""Diagnostic: timeline of k_small_solve (needs libvisfs_ba_hip_stamps.so built with -DVISFS_BA_STAMPS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
for cfg, solver in (("PROD", 0), ("PROD", 2), ("C1", 0), ("C1", 2)):
    w = synth.make_window(cfg); prm = abi.default_params(iterations=10, solver=solver)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(3):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    t = out.view(np.uint64).astype(np.int64)[64:70]
    names = ["schur blocks", "assemble", "solve (wave 0)", "barrier", "x + oplus"]
    print(cfg, "solver", solver, " ".join(f"{names[i]}={(t[i + 1] - t[i]) * 10}ns" for i in range(5)), "| total", (t[5] - t[0]) * 10, "ns")
    s.close()
