"""Diagnostic: where the one-workgroup PCG (k_pcg_cu, VISFS_BA_PCG_CU=1) spends a solve (needs libvisfs_ba_hip_stamps.so built with -DVISFS_BA_STAMPS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VISFS_BA_PCG_CU"] = "1"
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
for CFG in sys.argv[1:] or ["C2"]:
    w = synth.make_window(CFG); prm = abi.default_params(iterations=10, solver=2)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(2):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    st = out.view(np.uint64).astype(np.int64)
    ns = lambda a, b: int(st[a] - st[b]) * 10
    it = int(st[41])
    print(f"{CFG}: last solve of the run: {it} iterations | set-up {ns(1, 0)} ns | loop {ns(40, 1)} | total {ns(40, 0)}")
    for i in range(min(it, 8)):
        prev = 1 if i == 0 else 5 + 4 * (i - 1)
        print(f"   iteration {i}: S d {ns(2 + 4 * i, prev):5d} | sum d.q {ns(3 + 4 * i, 2 + 4 * i):5d} | x, r, Minv r {ns(4 + 4 * i, 3 + 4 * i):5d} | sum r.z, d {ns(5 + 4 * i, 4 + 4 * i):5d} | iteration {ns(5 + 4 * i, prev)}")
    s.close()
