"""Diagnostic: timeline of one workgroup of k_schur_runs (needs libvisfs_ba_hip_stamps.so: tools/build_stamps.sh).
usage: python tools/schur_run_stamps.py C2"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
CFG = sys.argv[1] if len(sys.argv) > 1 else "C2"
for wg in (60, 100):
    os.environ["VISFS_BA_STAMP_WG"] = str(wg)
    w = synth.make_window(CFG); prm = abi.default_params(iterations=20, solver=2)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    d = s.describe()
    for _ in range(3):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    t = out.view(np.uint64).astype(np.int64)[96:108]
    names = ["descriptors + gate + (i,j) table + R + barrier", "global loads issued + barrier", "slot init + b_l + D_l + barrier", "tiles + barrier", "phase 2 (accumulate, LAST sub-batch)", "barrier",
             "third 0: LDS + sums + stores", "barrier", "third 1", "barrier", "third 2"]
    print(f"{CFG} runs {d['schur_runs']} x {d['schur_run_landmarks']} landmarks, workgroup {wg}: " + " | ".join(f"{names[i]} {(t[i + 1] - t[i]) * 10} ns" for i in range(11)) + f" | total {(t[11] - t[0]) * 10} ns")
    s.close()
