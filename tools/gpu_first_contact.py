# first-contact script: stage-by-stage parity HIP vs oracle, then full solves
import sys, time, ctypes as C
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from visfs_amd import abi, synth, backend
import oracle_lib
olib = oracle_lib.load()

def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    d = np.abs(a - b).max() if a.size else 0.0
    s = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return d / s

def stage_compare(cfg, solver, **kw):
    w = synth.make_window(cfg, **kw)
    wb = abi.WindowBuffers(w)
    prm = abi.default_params(iterations=20, solver=solver)
    gb, used, oref, mono = abi.pack_window_with(olib.oracle_pack_window, prm, wb)
    o = oracle_lib.OracleSystem(olib, prm, gb)
    s = backend.Solver(prm)
    s.upload(gb)
    oc, omd = o.linearize(); gc, gmd = s.linearize()
    print(f"[{cfg} solver={solver}] lin chi2 oracle {oc:.10g} hip {gc:.10g} rel {abs(oc-gc)/oc:.2e}; maxdiag {omd:.10g} {gmd:.10g}")
    for name, b in (("err", abi.BUF_OBS_ERR), ("chi2", abi.BUF_OBS_CHI2), ("w", abi.BUF_OBS_WEIGHT), ("Hpl", abi.BUF_HPL), ("Hll", abi.BUF_HLL), ("bl", abi.BUF_BL), ("Hpp", abi.BUF_HPP), ("bp", abi.BUF_BP)):
        print(f"   {name:5s} rel err {rel(s.fetch(b), o.fetch(b)):.3e}")
    lam = 1e-5 * omd
    ot = o.trial(lam); gt = s.trial(lam)
    print(f"   trial oracle (chi,scale,pcg,ok)={ot}\n   trial hip    (chi,scale,pcg,ok)={gt}")
    for name, b in (("S", abi.BUF_S), ("bs", abi.BUF_BS), ("dxp", abi.BUF_DX_POSE), ("dxl", abi.BUF_DX_POINT), ("poseT", abi.BUF_POSE_TRIAL), ("ptT", abi.BUF_POINT_TRIAL)):
        print(f"   {name:5s} rel err {rel(s.fetch(b), o.fetch(b)):.3e}")
    # full optimise
    o.reset(); s.reset()
    t0 = time.time(); rc_o, st_o, sec_o = o.optimize(); 
    t0 = time.time(); rc_g, st_g = s.optimize(); sec_g = time.time() - t0
    po, pto, outo, chio = o.download(); pg, ptg, outg, chig = s.download()
    print(f"   optimize: oracle rc={rc_o} iters={list(st_o.iterations_run)} trials={list(st_o.trials_run)} pcg={st_o.pcg_iterations} chi2={st_o.chi2_initial:.6g}->{st_o.chi2_phase1:.6g}->{st_o.chi2_final:.6g} out={st_o.n_outliers} {sec_o*1e3:.1f} ms")
    print(f"             hip    rc={rc_g} iters={list(st_g.iterations_run)} trials={list(st_g.trials_run)} pcg={st_g.pcg_iterations} chi2={st_g.chi2_initial:.6g}->{st_g.chi2_phase1:.6g}->{st_g.chi2_final:.6g} out={st_g.n_outliers} {sec_g*1e3:.2f} ms")
    print(f"             pose rel {rel(pg, po):.3e} pt rel {rel(ptg, pto):.3e} outlier mismatches {int((outo != outg).sum())} chi2/edge rel {rel(chig, chio):.3e}")
    # timing: repeated reset + optimize
    ts = []
    for _ in range(5):
        s.reset(); t0 = time.perf_counter(); s.optimize(); ts.append(time.perf_counter() - t0)
    its = sum(st_g.iterations_run)
    print(f"             hip reset+optimize {min(ts)*1e3:.3f} ms  => {its/min(ts):.0f} it/s (oracle {sum(st_o.iterations_run)/sec_o:.1f} it/s)")
    s.close(); o.close()

for cfg, solver in (("C1", 2), ("C1", 0), ("C3", 2), ("C2", 2), ("C2", 0), ("PROD", 2), ("C4", 2)):
    try:
        stage_compare(cfg, solver)
    except Exception as e:
        import traceback; traceback.print_exc()
        break
