"""Per-frame call path, A/B inside one process: visfs_ba_solve_window on the by-value path (VISFS_BA_FRAME_GRAPH=0) against the fixed-address
path with the launch sequence replayed across uploads (=2), interleaved call by call on two handles; medians.
usage: python tools/frame_ab.py [PROD C1 C2 C3 C4] """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from visfs_amd import abi, backend, synth


def main():
    cfgs = sys.argv[1:] or ["PROD", "C1", "C2", "C3", "C4"]
    for cfg in cfgs:
        its = 10 if cfg in ("PROD", "C4") else 20
        n = 16 if cfg == "C4" else 60
        w = synth.make_window(cfg)
        prm = abi.default_params(iterations=its, solver=2)
        hs = {m: backend.Solver(prm) for m in ("0", "2")}
        wbs = [abi.WindowBuffers(w) for _ in range(2 * (n + 4))]
        rbs = [abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs) for wb in wbs]
        for r in rbs:
            r.outlier_feature[:] = 1; r.outlier_pose[:] = 1
        t = {"0": [], "2": []}
        replays = 0
        k = 0
        for i in range(n + 4):
            for m in ("0", "2"):
                os.environ["VISFS_BA_FRAME_GRAPH"] = m
                t0 = time.perf_counter()
                rc, rb = hs[m].solve_window(wbs[k], rbs[k])
                dt = time.perf_counter() - t0
                k += 1
                if i >= 4:
                    t[m].append(dt)
                    if m == "2":
                        replays += hs[m].describe()["graph_replayed"]
        a, b = np.median(t["0"]) * 1e3, np.median(t["2"]) * 1e3
        print(f"{cfg} (Iterations={its}): by-value {a:.4f} ms | fixed address + replay {b:.4f} ms ({100 * (b / a - 1):+.1f} %) | min {min(t['0']) * 1e3:.4f} / {min(t['2']) * 1e3:.4f} | "
              f"{replays} of {n} calls replayed | status {rc}, iterations {list(rb.struct.iterations_run)}", flush=True)
        for h in hs.values():
            h.close()


if __name__ == "__main__":
    main()
