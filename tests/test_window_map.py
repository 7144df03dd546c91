"""the native sliding-window container (include/visfs_window.h, visfs_amd/host/WindowMap.cpp)
against the Python restatement of LocalMap (tests/localmap_oracle.py) over scripted insert / build / update / remove
sequences.  Host-only code, so these run without a GPU; the GPU end-to-end case is marked `gpu`."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from localmap_oracle import NEW_ADDED, STABLE, LocalMapOracle
from window_sim import FrontEnd, insert_native, insert_oracle
from visfs_amd import abi, build, window

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wlib():
    build.build_host()
    return window.load()


def same_state(wm, lm):
    d = wm.dump()
    assert sorted(d["signatures"]) == sorted(lm.signatures)
    for sid, T in d["signatures"].items():
        assert np.array_equal(T, np.array(lm.signatures[sid]["pose"]))
    assert sorted(d["features"]) == sorted(lm.features)
    for fid, f in d["features"].items():
        g = lm.features[fid]
        assert (f["start"], f["end"], f["state"]) == (g["start"], g["end"], g["state"]), fid
        assert np.array_equal(f["pose"], np.array(g["pose"])), fid
        assert sorted(f["obs"]) == sorted(g["obs"]), fid
        for sid in f["obs"]:
            assert np.array_equal(f["obs"][sid], g["obs"][sid], equal_nan=True), (fid, sid)
    nf, ns, par, tr = wm.counters()
    assert (nf, ns) == (lm.new_feature_count, lm.signature_count)
    assert np.array_equal(np.float32(par), np.float32(lm.parallax_count), equal_nan=True)
    assert np.array_equal(tr, np.array(lm.translation_count))
    assert wm.is_key_signature() == lm.key_signature
    assert wm.available() == lm.available()


def same_window(wm, lm, fe, with_links=True):
    w = wm.build_dict(fe.Trc, fe.fx, fe.fy, fe.cx, fe.cy, fe.baseline, 2, with_links)
    poses = lm.poses()
    assert list(w["pose_ids"]) == list(poses)
    assert np.array_equal(w["pose_Twr"], np.array([poses[k] for k in poses]).reshape(-1, 12))
    links = lm.links() if with_links else {}
    assert [(int(a), int(b)) for a, b in zip(w["link_from"], w["link_to"])] == [(v[0], v[1]) for v in links.values()]
    assert np.array_equal(w["link_T"], np.array([v[2] for v in links.values()]).reshape(-1, 12))
    points, obs = lm.points_and_observations(fe.Trc)
    assert list(w["point_ids"]) == list(points)
    assert np.array_equal(w["point_xyz"], np.array([points[k][0] for k in points]).reshape(-1, 3))
    assert list(w["point_fixed"]) == [int(points[k][1]) for k in points]
    flat = [(fid, sid, *obs[fid][sid]) for fid in obs for sid in obs[fid]]
    assert [(int(a), int(b)) for a, b in zip(w["ref_feature"], w["ref_pose"])] == [(r[0], r[1]) for r in flat]
    assert np.array_equal(w["ref_u"], np.array([r[2] for r in flat], np.float32))
    assert np.array_equal(w["ref_v"], np.array([r[3] for r in flat], np.float32))
    assert np.array_equal(w["ref_depth"], np.array([r[4] for r in flat], np.float32), equal_nan=True)
    assert w["root_id"] == max(poses) - 1
    return w


def fake_ba(rng, w, kill_features=3, p_outlier=0.03):
    """A stand-in BA result: perturbed poses / points, scattered outliers plus a few features with every reference flagged."""
    poses = {int(i): (T + rng.normal(0, 1e-3, 12)) for i, T in zip(w["pose_ids"], w["pose_Twr"])}
    points = {int(i): p + rng.normal(0, 1e-2, 3) for i, p in zip(w["point_ids"], w["point_xyz"])}
    refs = list(zip((int(v) for v in w["ref_feature"]), (int(v) for v in w["ref_pose"])))
    doomed = set(rng.choice(w["point_ids"], size=min(kill_features, len(w["point_ids"])), replace=False).tolist()) if len(w["point_ids"]) else set()
    outliers = [r for r in refs if r[0] in doomed or rng.random() < p_outlier]
    return poses, points, outliers


@pytest.mark.parametrize("seed,params", [
    (1, {}),
    (2, {"LocalMap/MapSize": 3, "Tracker/MaxFeatures": 50, "LocalMap/MinParallax": 25, "LocalMap/MinTranslation": 0.2}),
    (3, {"LocalMap/MapSize": 5, "Tracker/MaxFeatures": 80, "Estimator/MinInliers": 30}),
    (4, {"LocalMap/MapSize": 8, "Tracker/MaxFeatures": 300, "LocalMap/MinParallax": 40}),
])
def test_scripted_sequence_matches_localmap(wlib, seed, params):
    fe = FrontEnd(seed=seed, n_tracks=70 if seed != 1 else 120)
    wm, lm = window.WindowMap(params, lib=wlib), LocalMapOracle(params)
    rng = np.random.default_rng(100 + seed)
    size = int(params.get("LocalMap/MapSize", 5))
    seen_key = seen_nonkey = seen_stable = seen_error = seen_nolink = 0
    for _ in range(60):
        s = fe.frame()
        assert insert_native(wm, s) == insert_oracle(lm, s)
        same_state(wm, lm)
        seen_key += lm.key_signature; seen_nonkey += not lm.key_signature
        if len(lm.signatures) >= 2:
            w = same_window(wm, lm, fe)
            seen_nolink += len(w["link_from"]) < len(w["pose_ids"]) - 1
            if len(lm.signatures) == size + 1:               # Estimator.cpp:275: update only when the full window came back
                poses, points, outliers = fake_ba(rng, w)
                ev_native = wm.update(list(poses), np.array(list(poses.values())), list(points), np.array(list(points.values())), outliers)
                ev_oracle = lm.update(poses, points, outliers)
                assert ev_native == sorted(ev_oracle)
                seen_error += len(ev_oracle)
                same_state(wm, lm)
        wm.remove(); lm.remove()
        same_state(wm, lm)
        seen_stable += any(f["state"] == STABLE for f in lm.features.values())
    assert seen_key and seen_nonkey and seen_stable and seen_nolink
    if seed in (1, 3):
        assert seen_error


def test_min_translation_is_squared_twice_when_the_key_is_absent():
    # LocalMap.cpp:17,34-35: the default survives the parse and is squared again; an explicit 0.5 is squared once
    assert LocalMapOracle().min_translation == pytest.approx(3 * 0.75 * 0.75)
    assert LocalMapOracle({"LocalMap/MinTranslation": 0.5}).min_translation == pytest.approx(0.75)


def test_refuses_signature_without_3d_words_and_orders(wlib):
    wm = window.WindowMap(lib=wlib)
    T = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(12)
    assert wm.insert(1, T, T, [0, 0, 0], [1, 2], np.zeros((2, 4)), np.zeros((2, 3)), [0, 0], [], np.zeros((0, 2))) is False
    assert wm.dump()["signatures"] == {}
    with pytest.raises(window.WindowError):      # ids must ascend (std::map order)
        wm.insert(1, T, T, [0, 0, 0], [2, 1], np.zeros((2, 4)), np.ones((2, 3)), [1, 1], [], np.zeros((0, 2)))


def test_no_links_without_wheel_odometry_or_when_disabled(wlib):
    fe = FrontEnd(seed=9, p_no_wheel=1.0)
    wm, lm = window.WindowMap(lib=wlib), LocalMapOracle()
    for _ in range(4):
        s = fe.frame(); insert_native(wm, s); insert_oracle(lm, s)
    assert len(same_window(wm, lm, fe)["link_from"]) == 0
    fe2 = FrontEnd(seed=9, p_no_wheel=0.0)
    wm2, lm2 = window.WindowMap(lib=wlib), LocalMapOracle()
    for _ in range(4):
        s = fe2.frame(); insert_native(wm2, s); insert_oracle(lm2, s)
    assert len(same_window(wm2, lm2, fe2)["link_from"]) == 3
    assert len(same_window(wm2, lm2, fe2, with_links=False)["link_from"]) == 0


def test_header_symbols_are_exported(wlib):
    text = open(os.path.join(ROOT, "include", "visfs_window.h")).read()
    names = set(re.findall(r"\b(visfs_window_[a-z_]+)\s*\(", text))
    assert len(names) >= 13
    for n in names:
        assert hasattr(wlib, n), n


@pytest.mark.gpu
def test_estimator_loop_on_gpu_matches_localmap_plus_oracle(wlib, olib):
    """The estimator's BA step (Estimator.cpp:227-317): window → localOptimize → updateLocalMap → removeSignature, native
    container + HIP backend against LocalMap restatement + CPU oracle, frame after frame on the same front-end data."""
    import types
    from visfs_amd import backend
    from helpers import rel_err
    fe = FrontEnd(seed=21, n_tracks=150, p_no_wheel=0.1)
    wm, lm = window.WindowMap(lib=wlib), LocalMapOracle()
    prm = abi.default_params(iterations=10, solver=2)
    s = backend.Solver(prm)
    solved = culled = 0
    for _ in range(30):
        sig = fe.frame()
        assert insert_native(wm, sig) == insert_oracle(lm, sig)
        if lm.available():
            # --- native: the window struct goes to the GPU backend as built, results come back through visfs_window_apply
            wm.build(fe.Trc, fe.fx, fe.fy, fe.cx, fe.cy, fe.baseline, 2, True)
            rc_g, rb_g = s.solve_window(types.SimpleNamespace(struct=wm.struct))
            # --- checker
            poses, links = lm.poses(), lm.links()
            points, obs = lm.points_and_observations(fe.Trc)
            flat = [(fid, sid, *obs[fid][sid]) for fid in obs for sid in obs[fid]]
            w = dict(root_id=max(poses) - 1, pose_ids=list(poses), pose_Twr=np.array(list(poses.values())), link_from=[v[0] for v in links.values()],
                     link_to=[v[1] for v in links.values()], link_T=np.array([v[2] for v in links.values()]).reshape(-1, 12), n_cameras=2,
                     fx=fe.fx, fy=fe.fy, cx=fe.cx, cy=fe.cy, baseline=fe.baseline, Trc=fe.Trc, point_ids=list(points),
                     point_xyz=np.array([points[k][0] for k in points]).reshape(-1, 3), point_fixed=[int(points[k][1]) for k in points],
                     ref_feature=[r[0] for r in flat], ref_pose=[r[1] for r in flat], ref_u=[r[2] for r in flat], ref_v=[r[3] for r in flat],
                     ref_depth=[r[4] for r in flat])
            wb_o = abi.WindowBuffers(w)
            rb_o = abi.ResultBuffers(wb_o.struct.n_poses, wb_o.struct.n_refs)
            rc_o = olib.oracle_solve_window(C.byref(prm), C.byref(wb_o.struct), C.byref(rb_o.struct), 1)
            assert rc_g == rc_o
            assert rb_g.struct.n_poses_out == rb_o.struct.n_poses_out
            assert rb_g.outliers() == rb_o.outliers()
            if rb_o.struct.n_poses_out == lm.map_size + 1:      # Estimator.cpp:275-276
                ev_g = wm.apply(rb_g.struct)
                ev_o = lm.update(rb_o.poses() and {k: v.reshape(12) for k, v in rb_o.poses().items()},
                                 {int(i): p for i, p in zip(wb_o.point_ids, wb_o.point_xyz)}, rb_o.outliers())
                assert ev_g == sorted(ev_o)
                solved += 1; culled += len(rb_o.outliers())
        wm.remove(); lm.remove()
        d = wm.dump()
        assert sorted(d["signatures"]) == sorted(lm.signatures) and sorted(d["features"]) == sorted(lm.features)
        for sid, T in d["signatures"].items():
            assert np.abs(T - np.array(lm.signatures[sid]["pose"])).max() < 1e-8
        for fid, f in d["features"].items():
            g = lm.features[fid]
            assert (f["state"], sorted(f["obs"])) == (g["state"], sorted(g["obs"]))
            assert rel_err(f["pose"], np.array(g["pose"])) < 1e-7
    s.close()
    assert solved >= 20


@pytest.mark.gpu
def test_a_handle_that_lives_across_frames_equals_fresh_uploads(wlib, monkeypatch):
    """The per-frame call path (Estimator.cpp:254): ONE handle serves 30 consecutive windows of the sliding map — its device arenas,
    pinned staging and host threads survive from frame to frame while poses, landmarks and observations enter and leave — and every
    result must be the bytes of (a) a fresh handle solving that window alone and (b) the GRAPH layer fed by the serial graph build
    (visfs_ba_pack_window -> visfs_ba_graph_upload -> optimize -> download): nothing a previous frame left behind may leak into the next."""
    import types
    from visfs_amd import backend
    fe = FrontEnd(seed=33, n_tracks=180, p_no_wheel=0.1)
    wm = window.WindowMap(lib=wlib)
    prm = abi.default_params(iterations=10, solver=2)
    keep = backend.Solver(prm)
    solved = 0
    for _ in range(30):
        insert_native(wm, fe.frame())
        if wm.available():
            wm.build(fe.Trc, fe.fx, fe.fy, fe.cx, fe.cy, fe.baseline, 2, True)
            win = types.SimpleNamespace(struct=wm.struct, point_xyz=np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3)),
                                        point_fixed=np.ctypeslib.as_array(wm.struct.point_fixed, shape=(wm.struct.n_points,)), laser_xyz=np.zeros((0, 3)), grid=None)
            xyz0 = np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3)).copy()
            rc_a, rb_a = keep.solve_window(win)
            xyz_a = np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3)).copy()
            np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3))[:] = xyz0           # localOptimize writes the landmarks in place
            fresh = backend.Solver(prm)
            rc_b, rb_b = fresh.solve_window(win)
            xyz_b = np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3)).copy()
            np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3))[:] = xyz0
            assert rc_a == rc_b and rb_a.outliers() == rb_b.outliers()
            n = rb_a.struct.n_poses_out
            assert n == rb_b.struct.n_poses_out and np.array_equal(rb_a.pose_Twr_out[:n], rb_b.pose_Twr_out[:n])
            assert np.array_equal(xyz_a, xyz_b, equal_nan=True)
            # the GRAPH layer on the serial graph build
            gb, used, oref, mono = abi.pack_window_with(fresh.lib.visfs_ba_pack_window, prm, win)
            fresh.upload(gb)
            rc_c, st_c = fresh.optimize()
            pose_c, pt_c, out_c, _ = fresh.download()
            fresh.close()
            if rc_a == abi.OK:
                assert rc_c == abi.OK
                twr = np.zeros((len(pose_c), 12))
                trc = np.ascontiguousarray(fe.Trc, np.float64).reshape(12)
                for i in range(len(pose_c)):
                    keep.lib.visfs_ba_unpack_pose(np.ascontiguousarray(pose_c[i]).ctypes.data_as(C.POINTER(C.c_double)), trc.ctypes.data_as(C.POINTER(C.c_double)),
                                                  twr[i].ctypes.data_as(C.POINTER(C.c_double)))
                assert np.array_equal(twr, rb_a.pose_Twr_out[:n])
                assert [(int(wm.struct.ref_feature[oref[k]]), int(wm.struct.ref_pose[oref[k]])) for k in np.nonzero(out_c)[0]] == rb_a.outliers()
                if rb_a.struct.n_poses_out == wm.struct.n_poses:
                    np.ctypeslib.as_array(wm.struct.point_xyz, shape=(wm.struct.n_points, 3))[:] = xyz_a
                    wm.apply(rb_a.struct)
                    solved += 1
        wm.remove()
    keep.close()
    assert solved >= 15


@pytest.mark.gpu
def test_graph_build_on_host_threads_gives_the_serial_bytes(monkeypatch):
    """BASELINE C2 through visfs_ba_solve_window with one host thread and with four (references shared out at feature boundaries,
    structure summary per thread): identical poses, landmarks and outlier lists — and identical to the GRAPH layer on the serial build."""
    from visfs_amd import backend, synth
    w = synth.make_window("C2")
    prm = abi.default_params(iterations=10, solver=2)
    res = {}
    for nthreads in ("1", "4"):
        monkeypatch.setenv("VISFS_BA_THREADS", nthreads)
        s = backend.Solver(prm)
        wb = abi.WindowBuffers(w)
        rc, rb = s.solve_window(wb)
        assert rc == abi.OK
        res[nthreads] = (rb.pose_Twr_out.copy(), wb.point_xyz.copy(), rb.outliers())
        if nthreads == "4":
            gb, used, oref, mono = abi.pack_window_with(s.lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
            s.upload(gb); rcg, _ = s.optimize(); pose_g, pt_g, out_g, _ = s.download()
            assert rcg == abi.OK
            wbg = abi.WindowBuffers(w)
            assert [(int(wbg.ref_feature[oref[k]]), int(wbg.ref_pose[oref[k]])) for k in np.nonzero(out_g)[0]] == res["4"][2]
        s.close()
    assert np.array_equal(res["1"][0], res["4"][0]) and np.array_equal(res["1"][1], res["4"][1], equal_nan=True) and res["1"][2] == res["4"][2]
