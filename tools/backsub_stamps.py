"""Diagnostic: timeline of one landmark workgroup of the fused speculative kernel k_backsub<LINA> (needs libvisfs_ba_hip_stamps.so built
with -DVISFS_BA_STAMPS: tools/build_stamps.sh).  usage: python tools/backsub_stamps.py C2"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
CFG = sys.argv[1] if len(sys.argv) > 1 else "C2"
for wg in (0, 100, 300):
    os.environ["VISFS_BA_STAMP_WG"] = str(wg)
    w = synth.make_window(CFG); prm = abi.default_params(iterations=20, solver=2)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(3):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    t = out.view(np.uint64).astype(np.int64)[32:38]
    names = ["gate (LmState)", "pose staging + barrier", "back-substitution + trial chi2", "two block sums", "role A of the linearisation"]
    print(f"{CFG} workgroup {wg}: " + " | ".join(f"{names[i]} {(t[i + 1] - t[i]) * 10} ns" for i in range(5)) + f" | total {(t[5] - t[0]) * 10} ns")
    s.close()
