"""Writes the flat factor graphs of the cross-check set as .vbag files (visfs_amd/graphio.py) under tests/golden/graphs/.
These are INPUTS (synthetic windows of visfs_amd.synth packed by the product's host graph build); a machine with the real g2o
turns them into .vbar results with tools/g2o_crosscheck.cpp (the ceres_* ones: a machine with Ceres <= 2.1, tools/ceres_crosscheck.cpp),
and tools/g2o_golden_import.py makes fixtures of those."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import hard_window, ragged_window
from visfs_amd import abi, backend, graphio, synth

CASES = {
    # name: (window, params)
    "c1_pcg": (lambda: synth.make_window("C1"), dict(iterations=20, solver=2)),
    "c1_eigen_cholesky": (lambda: synth.make_window("C1"), dict(iterations=20, solver=3)),
    "c1_gauss_newton": (lambda: synth.make_window("C1"), dict(iterations=10, solver=3, trust_region=1)),
    "prod_odo_pcg": (lambda: synth.make_window("PROD"), dict(iterations=10, solver=2)),
    "prod_odo_default": (lambda: synth.make_window("PROD"), dict(iterations=10, solver=0)),
    "ragged_odo_pcg": (lambda: ragged_window(seed=7), dict(iterations=20, solver=2)),
    "hard_rejected_steps": (lambda: hard_window(), dict(iterations=20, solver=3)),
    "no_kernel": (lambda: synth.make_window("C1", window_index=1), dict(iterations=10, solver=3, robust_kernel_delta=0.0)),
    # Optimizer/Framework=1 (the Ceres branch): inputs for tools/ceres_crosscheck.cpp
    "ceres_c1": (lambda: synth.make_window("C1"), dict(framework=1, iterations=20)),
    "ceres_prod": (lambda: synth.make_window("PROD"), dict(framework=1, iterations=10)),
    "ceres_hard_rejected_steps": (lambda: hard_window(), dict(framework=1, iterations=20)),
    "ceres_ragged": (lambda: ragged_window(seed=7), dict(framework=1, iterations=20, solver=1)),
}


def main(out_dir=os.path.join(ROOT, "tests", "golden", "graphs")):
    os.makedirs(out_dir, exist_ok=True)
    lib = backend.load_library()
    for name, (mk, kw) in CASES.items():
        prm = abi.default_params(**kw)
        gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(mk()))
        path = os.path.join(out_dir, name + ".vbag")
        graphio.dump_graph(path, prm, gb)
        print(f"{path}: {gb.n_poses} poses, {gb.n_points} points, {gb.n_obs} stereo edges, {gb.struct.n_odo} odometry edges, "
              f"{os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main(*sys.argv[1:])
