"""Soak of the per-frame call path at sizes where the host threads and the arena growth paths are active: ONE handle solves a sequence of
windows whose shape changes from frame to frame (20 ... 60 key-frames, 2 000 ... 6 000 landmarks, 16 000 ... 60 000 references; every
few frames a small production-size window in between); every result must be the bytes of a fresh handle solving that window alone.
usage: python tools/soak_frames.py 60"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from visfs_amd import abi, backend, synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(2026)
    prm = abi.default_params(iterations=10, solver=2)
    keep = backend.Solver(prm)
    bad = 0
    for f in range(n):
        if f % 7 == 3:
            w = synth.make_window("PROD", window_index=f)
        else:
            n_kf = int(rng.integers(20, 61)); n_lm = int(rng.integers(2000, 6001))
            track = int(rng.integers(6, min(n_kf, 12) + 1))
            w = synth.make_window("custom", n_kf=n_kf, n_lm=n_lm, n_obs=n_lm * track, seed=1000 + f)
        wa, wb = abi.WindowBuffers(w), abi.WindowBuffers(w)
        rc_a, rb_a = keep.solve_window(wa)
        fresh = backend.Solver(prm); rc_b, rb_b = fresh.solve_window(wb); fresh.close()
        na = rb_a.struct.n_poses_out
        same = (rc_a == rc_b and na == rb_b.struct.n_poses_out and np.array_equal(rb_a.pose_Twr_out[:na], rb_b.pose_Twr_out[:na])
                and rb_a.outliers() == rb_b.outliers() and np.array_equal(wa.point_xyz, wb.point_xyz, equal_nan=True))
        if not same:
            bad += 1
            print(f"frame {f}: DIFFERS (rc {rc_a} / {rc_b})", flush=True)
        if f % 10 == 9:
            print(f"... {f + 1} frames, {bad} differing", flush=True)
    keep.close()
    print(f"frame soak: {n} frames on one handle, {bad} differing from a fresh handle")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
