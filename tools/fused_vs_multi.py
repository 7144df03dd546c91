"""Small windows: general multi-kernel path / + k_small_solve (default for <= 10 free poses) / fused single-workgroup kernel."""
import os
import time

import numpy as np

from visfs_amd import abi, backend, synth

MODES = {"general": dict(VISFS_BA_SMALL_SOLVE="0", VISFS_BA_FUSED="0"), "small_solve": dict(VISFS_BA_SMALL_SOLVE="1", VISFS_BA_FUSED="0"),
         "fused": dict(VISFS_BA_SMALL_SOLVE="1", VISFS_BA_FUSED="1")}


def run(cfg, solver, mode, it=10, n=40, **kw):
    os.environ.update(MODES[mode])
    w = synth.make_laser_window(**kw) if cfg == "LASER" else synth.make_window(cfg)
    prm = abi.default_params(iterations=it, solver=solver)
    s = backend.Solver(prm)
    wb = abi.WindowBuffers(w)
    gb, *_ = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, wb)
    s.upload(gb)
    info = s.describe()
    rc, st = s.optimize()
    t0 = time.perf_counter()
    for _ in range(n):
        s.reset(); s.optimize()
    dt = (time.perf_counter() - t0) / n
    pose, pt, outl, chi = s.download()
    s.close()
    return dict(fused=info["fused_path"], rc=rc, it=list(st.iterations_run), trials=list(st.trials_run), pcg=st.pcg_iterations, chi=st.chi2_final,
                nout=st.n_outliers, ms=round(1e3 * dt, 4)), pose, pt, outl


def main():
    for cfg, kw in (("PROD", {}), ("C1", {}), ("LASER", dict(with_visual=True, n_points=720))):
        for solver in (2, 0):
            ref = None
            for mode in MODES:
                a, pa, pta, oa = run(cfg, solver, mode, **kw)
                if ref is None:
                    ref = (pa, pta, oa)
                print(cfg, "solver", solver, f"{mode:12s}", a, "| vs general: max |dpose|", float(np.abs(pa - ref[0]).max()),
                      "outliers equal", bool(np.array_equal(oa, ref[2])), flush=True)
    for k in MODES["general"]:
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
