"""visfs_ba_solve_window in a loop on one handle (the per-frame call path) — for rocprofv3 --kernel-trace --stats.
usage: python tools/frame_loop.py C2 30"""
import sys
import time

from visfs_amd import abi, backend, synth


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    w = synth.make_window(cfg)
    s = backend.Solver(abi.default_params(iterations=20, solver=2))
    # the caller's buffers exist before the clock starts (inputs AND outputs, touched: a fresh numpy array is untouched pages)
    wbs = [abi.WindowBuffers(w) for _ in range(n + 3)]
    rbs = [abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs) for wb in wbs]
    for r in rbs:
        r.outlier_feature[:] = 1; r.outlier_pose[:] = 1
    for i in range(3):
        s.solve_window(wbs[i], rbs[i])
    t0 = time.perf_counter()
    for i in range(n):
        rc, rb = s.solve_window(wbs[3 + i], rbs[3 + i])
    dt = (time.perf_counter() - t0) / n
    print(f"{cfg}: visfs_ba_solve_window {dt * 1e3:.3f} ms per call over {n} calls (status {rc}, iterations {list(rb.struct.iterations_run)})")
    s.close()


if __name__ == "__main__":
    main()
