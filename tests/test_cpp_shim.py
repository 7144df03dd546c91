"""The C++ drop-in `VISFS::Optimizer::Optimizer` (visfs_amd/host/) driven through the reference's own signature:
std::map poses / links / points3D / wordReferences in, std::map poses out, points updated in place, outliers appended."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import rel_err
from visfs_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory, hiplib):
    exe = str(tmp_path_factory.mktemp("shim") / "shim_driver")
    libdir = os.path.join(ROOT, "visfs_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "visfs_amd", "host"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_driver.cpp"),
           os.path.join(ROOT, "visfs_amd", "host", "Optimizer.cpp"), "-L" + libdir, "-lvisfs_ba_hip",
           "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def dump_window(path, w):
    n_laser = len(w.get("laser_xyz", [])) if w.get("grid") is not None else 0
    with open(path, "wb") as f:
        f.write(np.array([w["root_id"], len(w["pose_ids"]), len(w["link_from"]), len(w["point_ids"]), len(w["ref_feature"]),
                          w.get("n_cameras", 2), n_laser], np.int64).tobytes())
        f.write(np.array([w["fx"], w["fy"], w["cx"], w["cy"], np.float64(np.float32(w["baseline"]))], np.float64).tobytes())
        f.write(np.asarray(w["Trc"], np.float64).reshape(12).tobytes())
        for k, dt in (("pose_ids", np.uint64), ("pose_Twr", np.float64), ("link_from", np.uint64), ("link_to", np.uint64), ("link_T", np.float64),
                      ("point_ids", np.uint64), ("point_xyz", np.float64), ("point_fixed", np.uint8), ("ref_feature", np.uint64),
                      ("ref_pose", np.uint64), ("ref_u", np.float32), ("ref_v", np.float32), ("ref_depth", np.float32)):
            f.write(np.ascontiguousarray(w[k], dtype=dt).tobytes())
        if n_laser:
            g = w["grid"]
            f.write(np.array([g["resolution"], g["max_x"], g["max_y"]], np.float64).tobytes())
            f.write(np.array([g["cost"].shape[1], g["cost"].shape[0]], np.int64).tobytes())
            f.write(np.ascontiguousarray(g["cost"], np.float32).tobytes())
            f.write(np.ascontiguousarray(w["laser_xyz"], np.float64).tobytes())


def read_result(path):
    b = open(path, "rb").read()
    status, n_out, n_pts, n_outl = struct.unpack_from("4q", b, 0)
    off = 32
    poses = {}
    for _ in range(n_out):
        pid, = struct.unpack_from("Q", b, off); off += 8
        poses[pid] = np.frombuffer(b, np.float64, 12, off).copy(); off += 96
    pts = np.frombuffer(b, np.float64, 3 * n_pts, off).reshape(-1, 3).copy(); off += 24 * n_pts
    outl = np.frombuffer(b, np.uint64, 2 * n_outl, off).reshape(-1, 2).copy()
    return status, poses, pts, [tuple(int(v) for v in r) for r in outl]


def test_shim_compiles_and_fails_loudly_without_gpu(driver, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    w = synth.make_window("PROD")
    dump_window(tmp_path / "in.bin", w)
    res = subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "Optimizer/Iterations=10"], capture_output=True, text=True)
    assert res.returncode == 0 and "no CPU fallback" in res.stderr
    status, poses, pts, outl = read_result(tmp_path / "out.bin")
    assert status == abi.ERR_DEVICE and poses == {}                       # the reference's failure convention: empty map
    assert outl == [(123456, 654321)]                                     # nothing appended
    assert rel_err(pts, w["point_xyz"]) == 0.0                            # points3D untouched


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,args", [("PROD", ["Optimizer/Iterations=10", "Optimizer/Solver=2"]),
                                      ("C1", ["Optimizer/Iterations=20"]),                       # default Solver=0 → direct
                                      ("C3s", ["Optimizer/Iterations=20", "Optimizer/Solver=2"])])
def test_shim_matches_oracle_through_the_reference_signature(driver, olib, tmp_path, cfg, args):
    w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400) if cfg == "C3s" else synth.make_window(cfg)
    dump_window(tmp_path / "in.bin", w)
    subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")] + args, check=True)
    status, poses, pts, outl = read_result(tmp_path / "out.bin")
    kw = {a.split("=")[0].split("/")[1].lower(): a.split("=")[1] for a in args}
    prm = abi.default_params(iterations=int(kw.get("iterations", 10)), solver=int(kw.get("solver", 0)))
    wb = abi.WindowBuffers(w); rb = abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs)
    assert olib.oracle_solve_window(C.byref(prm), C.byref(wb.struct), C.byref(rb.struct), 1) == abi.OK == status
    ref = rb.poses()
    assert sorted(poses) == sorted(ref)
    A = np.array([poses[k] for k in sorted(poses)]); B = np.array([ref[k].reshape(12) for k in sorted(ref)])
    et, er = synth.pose_errors(A, B)
    assert et < 1e-6 and er < 1e-6
    assert outl[0] == (123456, 654321) and outl[1:] == rb.outliers()      # appended, in the reference's edge order
    assert rel_err(pts, wb.point_xyz) < 1e-6


@pytest.mark.gpu
def test_shim_laser_factor_matches_the_c_abi(driver, tmp_path):
    """Point clouds + Submap2D through the reference's signature (Optimizer.cpp:225-258) == the flat window through the C ABI."""
    from visfs_amd import backend
    w = synth.make_laser_window(with_visual=True, n_points=400, seed=2)
    dump_window(tmp_path / "in.bin", w)
    subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "Optimizer/Iterations=10", "Optimizer/Solver=2"], check=True, capture_output=True)
    status, poses, pts, outl = read_result(tmp_path / "out.bin")
    s = backend.Solver(abi.default_params(iterations=10, solver=2))
    wb = abi.WindowBuffers(w)
    rc, rb = s.solve_window(wb)
    s.close()
    assert status == rc == abi.OK and sorted(poses) == [int(i) for i in w["pose_ids"]]
    assert np.array_equal(np.array([poses[k] for k in sorted(poses)]), rb.pose_Twr_out[:len(poses)])
    assert outl[1:] == rb.outliers()
    # and the factor was really there
    w2 = dict(w); w2["grid"] = None
    dump_window(tmp_path / "in.bin", w2)
    subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "Optimizer/Iterations=10", "Optimizer/Solver=2"], check=True, capture_output=True)
    _, poses2, _, _ = read_result(tmp_path / "out.bin")
    assert not np.array_equal(np.array([poses2[k] for k in sorted(poses2)]), rb.pose_Twr_out[:len(poses)])


@pytest.mark.gpu
def test_shim_error_convention(driver, tmp_path):
    w = synth.make_window("PROD")
    # an Optimizer/Framework the reference does not know: an empty map comes back
    dump_window(tmp_path / "in.bin", w)
    subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "Optimizer/Framework=2"], check=True, capture_output=True)
    status, poses, pts, outl = read_result(tmp_path / "out.bin")
    assert status == abi.ERR_UNSUPPORTED and poses == {}
    # the Ceres branch, both trust-region strategies: same poses as the C ABI's
    from visfs_amd import backend
    for tr in (0, 1):
        subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "Optimizer/Framework=1", "Optimizer/TrustRegion=%d" % tr], check=True, capture_output=True)
        status, poses, pts, outl = read_result(tmp_path / "out.bin")
        s = backend.Solver(abi.default_params(framework=1, trust_region=tr))
        rc, rb = s.solve_window(abi.WindowBuffers(w))
        s.close()
        assert status == rc == abi.OK and np.array_equal(np.array([poses[k] for k in sorted(poses)]), rb.pose_Twr_out[:len(poses)])
    # single pose → input poses come back
    w1 = dict(w); w1["pose_ids"] = w["pose_ids"][:1]; w1["pose_Twr"] = w["pose_Twr"][:1]
    dump_window(tmp_path / "in.bin", w1)
    subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], check=True, capture_output=True)
    status, poses, pts, outl = read_result(tmp_path / "out.bin")
    assert status == abi.PASSTHROUGH and list(poses) == [1] and np.array_equal(poses[1], np.asarray(w["pose_Twr"])[0])
