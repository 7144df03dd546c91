"""Diagnostic: where the banded Cholesky (k_band_chol) spends a solve (needs libvisfs_ba_hip_stamps.so built with -DVISFS_BA_STAMPS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
for CFG in sys.argv[1:] or ["C2"]:
    w = synth.make_window(CFG); prm = abi.default_params(iterations=10, solver=0)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(2):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    st = out.view(np.uint64).astype(np.int64)
    ns = lambda a, b: int(st[a] - st[b]) * 10
    print(f"{CFG}: free poses {s.describe()['n_free_poses']}  load {ns(1, 0)} ns | factor loop {ns(110, 1)} | backward {ns(111, 110)} | epilogue {ns(112, 111)} | total {ns(112, 0)} | shader clock {float(st[121] - st[120]) / max(1, ns(112, 0)):.2f} GHz")
    print(f"   after the loop (first chunk): last forward step + reload {ns(113, 110) if st[113] else 0} ... X_k {ns(113, 110)} | rows scaled {ns(114, 113)} | backward chain {ns(115, 114)} | x, oplus {ns(111, 115)}")
    # round 4 layout of the stamps of step k (< 16): [2 + 6k] thread 0 after the barrier of half 1, [3 + 6k] wave 0 after the next pivot's inverse,
    # [5 + 6k] first helper thread after its trailing-update tasks, [4 + 6k] thread 0 after the barrier of half 2
    for k in range(1, 16, 2):
        prev = 4 + 6 * (k - 1)
        print(f"   step {k:2d}: half 1: wave 0 {ns(6 + 6 * k, prev):4d}, wave 3 {ns(7 + 6 * k, prev):4d}, to the barrier {ns(2 + 6 * k, prev):4d} | half 2: wave 0 factor {ns(3 + 6 * k, 2 + 6 * k):5d}, helpers' updates {ns(5 + 6 * k, 2 + 6 * k):5d}, to the barrier {ns(4 + 6 * k, 2 + 6 * k):5d} | step {ns(4 + 6 * k, prev)}")
    s.close()
