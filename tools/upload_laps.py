"""Phases of one graph upload (VISFS_BA_TIMING laps of ws_upload, stderr) for the bench configurations."""
import os
import sys

os.environ["VISFS_BA_TIMING"] = "1"
from visfs_amd import abi, backend, synth


def main():
    for cfg in sys.argv[1:] or ("C2", "C4"):
        w = synth.make_window(cfg)
        prm = abi.default_params(iterations=10, solver=2)
        s = backend.Solver(prm)
        gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
        for rep in range(3):
            print(f"== {cfg} upload #{rep}", file=sys.stderr, flush=True)
            s.upload(gb)
        for rep in range(3):
            print(f"== {cfg} solve_window #{rep}", file=sys.stderr, flush=True)
            s.solve_window(abi.WindowBuffers(w))
        s.close()


if __name__ == "__main__":
    main()
