"""Diagnostic: how far apart are the HIP path and the CPU checker stage by stage when the damping is tiny (lambda = 1e-8 ... 1e-2, g2o
flavour: the stage hooks) on a window whose reduced system is rank-deficient?  Prints the relative differences of S, b_s, the pose
step and the landmark step, and the component of the b_s difference along the weakest eigenvector of S.

usage: python tools/stage_precision.py 576 559"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib
import test_gpu_random as T
from helpers import rel_err
from test_gpu_parity import make_pair
from visfs_amd import abi


def main():
    olib = oracle_lib.load()
    for i in [int(a) for a in sys.argv[1:]]:
        w, kw = T.random_case(i)
        for solver in (0, 2):
            o, s, gb = make_pair(olib, w, iterations=10, solver=solver, robust_kernel_delta=kw["robust_kernel_delta"])
            o.linearize(); s.linearize()
            n6 = 6 * o.npf
            print(f"seed {i} solver {solver}: Hpp {rel_err(s.fetch(abi.BUF_HPP), o.fetch(abi.BUF_HPP)):.1e} bp {rel_err(s.fetch(abi.BUF_BP), o.fetch(abi.BUF_BP)):.1e} "
                  f"Hll {rel_err(s.fetch(abi.BUF_HLL), o.fetch(abi.BUF_HLL)):.1e} bl {rel_err(s.fetch(abi.BUF_BL), o.fetch(abi.BUF_BL)):.1e} W {rel_err(s.fetch(abi.BUF_HPL), o.fetch(abi.BUF_HPL)):.1e}")
            for lam in (1e-2, 1e-5, 1e-8):
                o.trial(lam); s.trial(lam)
                So = o.fetch(abi.BUF_S).reshape(n6, n6); Sg = s.fetch(abi.BUF_S).reshape(n6, n6)
                bo = o.fetch(abi.BUF_BS); bg = s.fetch(abi.BUF_BS)
                ev, V = np.linalg.eigh(So)
                xo = o.fetch(abi.BUF_DX_POSE); xg = s.fetch(abi.BUF_DX_POSE)
                # what each side's own (S, b_s) gives with a dense numpy solve: separates the solver from the assembly
                xo_np = np.linalg.solve(So, bo); xg_np = np.linalg.solve(Sg, bg)
                print(f"   lambda {lam:.0e}: eig min {ev[0]:.2e} max {ev[-1]:.2e} | S {rel_err(Sg, So):.1e} b_s {rel_err(bg, bo):.1e} | v0.(b_s diff) {abs(V[:, 0] @ (bg - bo)):.1e} v0.b_s {abs(V[:, 0] @ bo):.1e} |b_s| {np.abs(bo).max():.1e} | "
                      f"x {rel_err(xg, xo):.1e}; oracle x vs numpy on its S {rel_err(xo, xo_np):.1e}; HIP x vs numpy on its S {rel_err(xg, xg_np):.1e}; numpy(S_hip) vs numpy(S_oracle) {rel_err(xg_np, xo_np):.1e}")
            s.close(); o.close()


if __name__ == "__main__":
    main()
