"""End-to-end (host buffers in and out) throughput of visfs_ba_solve_batch for many production-size windows."""
import time

from visfs_amd import abi, backend, synth


def main():
    prm = abi.default_params(iterations=10, solver=0)
    for n in (8, 64, 256):
        ws = [synth.make_window("PROD", window_index=i) for i in range(n)]
        s = backend.Solver(prm)
        wbs = [abi.WindowBuffers(w) for w in ws]
        s.solve_batch(wbs)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            wbs = [abi.WindowBuffers(w) for w in ws]
            t1 = time.perf_counter()
            rbs = s.solve_batch(wbs)
        dt = (time.perf_counter() - t0) / reps
        t_call = time.perf_counter() - t1
        its = sum(r.struct.iterations_run[0] + r.struct.iterations_run[1] for r in rbs)
        print(f"{n:4d} windows: solve_batch {1e3 * t_call:7.2f} ms per call  -> {n / t_call:9.0f} windows/s, {its / t_call:10.0f} it/s  (with Python marshalling {1e3 * dt:.2f} ms)", flush=True)
        s.close()


if __name__ == "__main__":
    main()
