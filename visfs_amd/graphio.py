"""Flat factor-graph dump (`.vbag`) and result (`.vbar`) files — the hand-off format between this repo and a machine that has
the real g2o ("the flat-problem dump is the hand-off format for a true g2o cross-check").

A `.vbag` file holds exactly what `visfs_ba_graph_upload` takes (include/visfs_ba.h, `visfs_ba_graph`: the factor graph of
Optimizer.cpp:100-223 in the camera frame) plus the eight `Optimizer/*` parameters; `tools/g2o_crosscheck.cpp` reads it,
builds the reference's g2o graph from it, runs Optimizer.cpp:261-318 with the real library and writes a `.vbar` file, which
`tools/g2o_golden_import.py` turns into a fixture under tests/golden/ with provenance "g2o".

Layout (little endian, no padding):
  .vbag  magic "VISFSBAG" | u32 version = 1 | i32 framework, solver, trust_region, iterations | f64 pixel_variance,
         odometry_covariance, laser_covariance, robust_kernel_delta | i32 n_poses, n_points, n_obs, n_odo |
         f64 fx, fy, cx, cy, bf | f64 pose_tq[n_poses][7] (tx ty tz qx qy qz qw, T_cw) | u8 pose_fixed[n_poses] |
         f64 point_xyz[n_points][3] | u8 point_fixed[n_points] | i32 obs_point[n_obs] | i32 obs_pose[n_obs] |
         f64 obs_uvr[n_obs][3] | i32 odo_from[n_odo] | i32 odo_to[n_odo] | f64 odo_tq[n_odo][7]
         (laser edges are not carried: the cross-check covers the stereo + wheel-odometry factor set)
  .vbar  magic "VISFSBAR" | u32 version = 1 | i32 status, iterations_run[2], n_outliers | f64 chi2_initial, chi2_phase1,
         chi2_final | i32 n_poses, n_points, n_obs | f64 pose_tq[n_poses][7] | f64 point_xyz[n_points][3] |
         u8 obs_outlier[n_obs] | f64 obs_chi2[n_obs] | u32 len + bytes: free-text provenance (library name and version)
"""
import struct

import numpy as np

from . import abi

MAGIC_GRAPH, MAGIC_RESULT, VERSION = b"VISFSBAG", b"VISFSBAR", 1


def dump_graph(path, params, gb):
    """Write a `.vbag` file from `abi.Params` and `abi.GraphBuffers`."""
    if gb.struct.n_laser:
        raise ValueError("laser edges are not part of the .vbag format")
    with open(path, "wb") as f:
        f.write(MAGIC_GRAPH)
        f.write(struct.pack("<I4i4d", VERSION, params.framework, params.solver, params.trust_region, params.iterations,
                            params.pixel_variance, params.odometry_covariance, params.laser_covariance, params.robust_kernel_delta))
        g = gb.struct
        f.write(struct.pack("<4i5d", g.n_poses, g.n_points, g.n_obs, g.n_odo, g.fx, g.fy, g.cx, g.cy, g.bf))
        for a, dt in ((gb.pose_tq, "<f8"), (gb.pose_fixed, "u1"), (gb.point_xyz, "<f8"), (gb.point_fixed, "u1"),
                      (gb.obs_point, "<i4"), (gb.obs_pose, "<i4"), (gb.obs_uvr, "<f8"),
                      (gb.odo_from, "<i4"), (gb.odo_to, "<i4"), (gb.odo_tq, "<f8")):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())


def load_graph(path):
    """Read a `.vbag` file → (abi.Params, abi.GraphBuffers)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC_GRAPH:
        raise ValueError("not a .vbag file")
    off = 8
    ver, fw, solver, tr, iters, pv, oc, lc, rk = struct.unpack_from("<I4i4d", buf, off); off += struct.calcsize("<I4i4d")
    if ver != VERSION:
        raise ValueError(f"unsupported .vbag version {ver}")
    Np, Nl, No, Ne, fx, fy, cx, cy, bf = struct.unpack_from("<4i5d", buf, off); off += struct.calcsize("<4i5d")

    def take(n, dt):
        nonlocal off
        a = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
        off += a.nbytes
        return a
    pose_tq = take(Np * 7, "<f8").reshape(Np, 7); pose_fixed = take(Np, "u1")
    point_xyz = take(Nl * 3, "<f8").reshape(Nl, 3); point_fixed = take(Nl, "u1")
    obs_point = take(No, "<i4"); obs_pose = take(No, "<i4"); obs_uvr = take(No * 3, "<f8").reshape(No, 3)
    odo_from = take(Ne, "<i4"); odo_to = take(Ne, "<i4"); odo_tq = take(Ne * 7, "<f8").reshape(Ne, 7)
    if off != len(buf):
        raise ValueError("trailing bytes in .vbag file")
    prm = abi.Params(fw, solver, tr, iters, pv, oc, lc, rk)
    gb = abi.GraphBuffers(pose_tq, pose_fixed, point_xyz, point_fixed, obs_point, obs_pose, obs_uvr, odo_from, odo_to, odo_tq,
                          fx, fy, cx, cy, bf)
    return prm, gb


def dump_result(path, status, iterations_run, n_outliers, chi2, pose_tq, point_xyz, obs_outlier, obs_chi2, provenance):
    """Write a `.vbar` file (used by the tests to fake the cross-check side; the real writer is tools/g2o_crosscheck.cpp)."""
    pose_tq = np.ascontiguousarray(pose_tq, "<f8").reshape(-1, 7); point_xyz = np.ascontiguousarray(point_xyz, "<f8").reshape(-1, 3)
    obs_outlier = np.ascontiguousarray(obs_outlier, "u1"); obs_chi2 = np.ascontiguousarray(obs_chi2, "<f8")
    prov = provenance.encode()
    with open(path, "wb") as f:
        f.write(MAGIC_RESULT)
        f.write(struct.pack("<I4i3d3i", VERSION, status, iterations_run[0], iterations_run[1], n_outliers, chi2[0], chi2[1], chi2[2],
                            len(pose_tq), len(point_xyz), len(obs_outlier)))
        f.write(pose_tq.tobytes()); f.write(point_xyz.tobytes()); f.write(obs_outlier.tobytes()); f.write(obs_chi2.tobytes())
        f.write(struct.pack("<I", len(prov))); f.write(prov)


def load_result(path):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC_RESULT:
        raise ValueError("not a .vbar file")
    off = 8
    fmt = "<I4i3d3i"
    ver, status, it0, it1, nout, c0, c1, c2, Np, Nl, No = struct.unpack_from(fmt, buf, off); off += struct.calcsize(fmt)
    if ver != VERSION:
        raise ValueError(f"unsupported .vbar version {ver}")

    def take(n, dt):
        nonlocal off
        a = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
        off += a.nbytes
        return a
    pose = take(Np * 7, "<f8").reshape(Np, 7); pts = take(Nl * 3, "<f8").reshape(Nl, 3)
    outl = take(No, "u1"); chi = take(No, "<f8")
    (n,) = struct.unpack_from("<I", buf, off); off += 4
    prov = buf[off:off + n].decode()
    return dict(status=status, iterations_run=(it0, it1), n_outliers=nout, chi2_initial=c0, chi2_phase1=c1, chi2_final=c2,
                pose_tq=pose, point_xyz=pts, obs_outlier=outl, obs_chi2=chi, provenance=prov)
