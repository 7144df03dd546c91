"""Degenerate windows through `visfs_ba_solve_window`: whatever the CPU oracle does with them (status, iteration counts,
outliers, poses) the GPU path must do too — empty graphs, odometry only, all landmarks fixed, dangling references, zero /
NaN depths (the reference's uninitialised mono branch), non-finite landmarks (NaN abort, Optimizer.cpp:272-275), a pose far away."""
import ctypes as C

import numpy as np
import pytest

from helpers import drop_refs
from visfs_amd import abi, synth

pytestmark = pytest.mark.gpu


def _cases():
    w = synth.make_window("C1")
    yield "no references, no links", drop_refs(w, np.zeros(len(w["ref_feature"]), bool))
    p = synth.make_window("PROD")
    yield "links only", drop_refs(p, np.zeros(len(p["ref_feature"]), bool))
    c = dict(w); c["point_fixed"] = np.ones_like(w["point_fixed"])
    yield "all landmarks fixed", c
    c = dict(w); rp = np.asarray(w["ref_pose"]).copy(); rp[::7] = 999; c["ref_pose"] = rp
    yield "references to an unknown pose id", c
    c = dict(w); d = np.asarray(w["ref_depth"]).copy(); d[::5] = 0.0; d[1::5] = np.nan; c["ref_depth"] = d
    yield "zero and NaN depths", c
    c = dict(w); P = np.asarray(w["point_xyz"]).copy(); P[3] = [np.inf, 0, 0]; c["point_xyz"] = P
    yield "an infinite landmark", c
    c = dict(w); T = np.asarray(w["pose_Twr"]).copy(); T[2, 3] += 1e7; c["pose_Twr"] = T
    yield "a pose 1e7 m away", c


CASES = list(_cases())


@pytest.mark.parametrize("name,w", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("solver", [0, 2])
def test_degenerate_window_matches_oracle(olib, name, w, solver):
    from visfs_amd import backend
    prm = abi.default_params(iterations=10, solver=solver)
    wb_o, wb_g = abi.WindowBuffers(w), abi.WindowBuffers(w)
    rb_o = abi.ResultBuffers(wb_o.struct.n_poses, max(wb_o.struct.n_refs, 1))
    rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
    s = backend.Solver(prm)
    rc_g, rb_g = s.solve_window(wb_g)
    s.close()
    assert rc_g == rc_o
    assert rb_g.struct.n_poses_out == rb_o.struct.n_poses_out
    if rb_o.struct.chi2_final > 1e-10:      # at chi2 ~ 1e-16 (a perfectly consistent odometry chain) the accept / terminate decisions are rounding noise
        assert list(rb_g.struct.iterations_run) == list(rb_o.struct.iterations_run)
    assert rb_g.outliers() == rb_o.outliers()
    assert rb_g.struct.warn_mono_skipped == rb_o.struct.warn_mono_skipped
    n = rb_o.struct.n_poses_out
    if n:
        et, er = synth.pose_errors(rb_g.pose_Twr_out[:n], rb_o.pose_Twr_out[:n])
        assert et < 1e-6 and er < 1e-6
