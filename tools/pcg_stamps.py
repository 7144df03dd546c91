"""Diagnostic: where one PCG workgroup spends an iteration (needs libvisfs_ba_hip_stamps.so built with -DVISFS_BA_STAMPS)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from visfs_amd import abi, backend, synth
backend.LIB_PATH = os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip_stamps.so")
lib = backend.load_library()
CFG = sys.argv[1] if len(sys.argv) > 1 else "C2"
for wg in (0, 24, 48):
    os.environ["VISFS_BA_STAMP_WG"] = str(wg)
    w = synth.make_window(CFG); prm = abi.default_params(iterations=20, solver=2)
    gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
    s = backend.Solver(prm); s.upload(gb)
    for _ in range(3):
        s.reset(); s.optimize()
    out = np.zeros(128)
    s.lib.visfs_ba_stage_fetch(s.h, 100, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    st = out.view(np.uint64)
    t_start = int(st[127]); t0 = int(st[0])
    print(f"wg {wg}: setup {(t0 - t_start) * 10} ns")
    k = 0
    while 4 + 4 * k < 100 and st[4 + 4 * k] > st[0] and (k == 0 or st[4 + 4 * k] > st[4 * k]):
        a, b, c, d = (int(st[1 + 4 * k]), int(st[2 + 4 * k]), int(st[3 + 4 * k]), int(st[4 + 4 * k]))
        prev = t0 if k == 0 else int(st[4 * k])
        print(f"   iter {k}: matvec+barrier {(a - prev) * 10:6d} ns | publish {(b - a) * 10:5d} | gather+barrier {(c - b) * 10:6d} (extra sweeps {int(st[100 + k]) if k < 26 else -1}) | vector+barrier {(d - c) * 10:6d} | total {(d - prev) * 10}")
        k += 1
    s.close()
