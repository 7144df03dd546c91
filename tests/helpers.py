"""Shared test helpers: window variants, graph construction, comparison metrics."""
import ctypes as C

import numpy as np

from visfs_amd import abi, synth


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return float("inf")                      # NaNs (e.g. points without references) must sit in the same places
    if na.all():
        return 0.0
    a, b = a[~na], b[~nb]
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def drop_refs(w, keep_mask):
    """Return a copy of window dict `w` keeping only the references where keep_mask is True."""
    w = dict(w)
    for k in ("ref_feature", "ref_pose", "ref_u", "ref_v", "ref_depth"):
        w[k] = np.asarray(w[k])[keep_mask]
    if "gross" in w:
        w["gross"] = np.asarray(w["gross"])[keep_mask]
    return w


def ragged_window(seed=7, n_kf=12, n_lm=300, n_obs=2400, odo=True, drop=0.35, **kw):
    """A window with ragged tracks: random references removed, some landmarks left with 0 or 1 observation."""
    w = synth.make_window("custom", n_kf=n_kf, n_lm=n_lm, n_obs=n_obs, odo=odo, seed=seed, **kw)
    rng = np.random.default_rng(seed)
    keep = rng.uniform(size=len(w["ref_feature"])) > drop
    feat = np.asarray(w["ref_feature"])
    keep[feat == 3] = False                       # landmark 3: no observation at all
    idx5 = np.nonzero(feat == 5)[0]
    keep[idx5] = False; keep[idx5[:1]] = True     # landmark 5: a single observation
    return drop_refs(w, keep)


def graph_of(pack_fn, params, w):
    wb = abi.WindowBuffers(w)
    gb, used, oref, mono = abi.pack_window_with(pack_fn, params, wb)
    return wb, gb, used, oref, mono


def twr_of(lib_unpack, pose_tq, Trc):
    out = np.zeros((len(pose_tq), 12))
    trc = np.ascontiguousarray(Trc, dtype=np.float64).reshape(12)
    for i in range(len(pose_tq)):
        tq = np.ascontiguousarray(pose_tq[i])
        lib_unpack(tq.ctypes.data_as(C.POINTER(C.c_double)), trc.ctypes.data_as(C.POINTER(C.c_double)),
                   out[i].ctypes.data_as(C.POINTER(C.c_double)))
    return out


def hard_window(seed=5):
    """Large landmark noise, no fixed landmark: the LM loop rejects several damped solves (lambda *= ni path)."""
    return synth.make_window("custom", n_kf=20, n_lm=400, n_obs=4000, seed=seed, point_noise=1.0, fixed_frac=0.0)
