"""Where does a whole localOptimize call spend its time?  pack (host) / upload (index build + H2D) / optimize / download."""
import sys
import time

import numpy as np

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visfs_amd import abi, backend, synth


def main():
    for cfg, it in (("PROD", 10), ("C1", 10), ("C2", 20), ("C4", 10)):
        w = synth.make_window(cfg)
        prm = abi.default_params(iterations=it, solver=2)
        s = backend.Solver(prm)
        wb = abi.WindowBuffers(w)
        gb, used, oref, mono = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, wb)
        s.upload(gb); s.optimize()
        n = 20 if cfg != "C4" else 5
        t = dict(pack=0.0, upload=0.0, optimize=0.0, download=0.0, window=0.0)
        for _ in range(n):
            t0 = time.perf_counter(); gb, used, oref, mono = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, wb)
            t1 = time.perf_counter(); s.upload(gb)
            t2 = time.perf_counter(); s.optimize()
            t3 = time.perf_counter(); s.download()
            t4 = time.perf_counter()
            t["pack"] += t1 - t0; t["upload"] += t2 - t1; t["optimize"] += t3 - t2; t["download"] += t4 - t3
            # (round 3: the caller's input and output buffers exist before the clock starts — rounds 1-2 timed their construction in
            # Python, 40 us at C2 / 250 us at C4, together with the call)
            wbw = abi.WindowBuffers(w); rbw = abi.ResultBuffers(wbw.struct.n_poses, wbw.struct.n_refs)
            rbw.outlier_feature[:] = 1; rbw.outlier_pose[:] = 1
            t0 = time.perf_counter(); s.solve_window(wbw, rbw); t["window"] += time.perf_counter() - t0
            t0 = time.perf_counter(); s.solve_window(abi.WindowBuffers(w)); t["window_r02"] = t.get("window_r02", 0.0) + time.perf_counter() - t0
        info = s.describe()
        print(cfg, {k: round(1e3 * v / n, 3) for k, v in t.items()}, "ms;  pairs", info["n_pairs"], "device MB", round(info["device_bytes"] / 1e6, 1), flush=True)
        s.close()


if __name__ == "__main__":
    main()
