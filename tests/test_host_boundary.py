"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every declared symbol,
struct layouts match the header, the host-side graph build of the PRODUCT (visfs_ba_pack_window, Optimizer.cpp:100-223)
is bit-identical to the oracle's, and the product refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import graph_of, ragged_window, twr_of
from visfs_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "visfs_ba.h")


def test_library_exports_every_declared_symbol(hiplib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(visfs_ba_[a-z_]+)\s*\(", text)))
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(hiplib, name), f"{name} declared in include/visfs_ba.h but not exported"
    from visfs_amd import backend
    assert sorted(backend.EXPORTS) == declared
    assert hiplib.visfs_ba_abi_version() == abi.ABI_VERSION


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof from a C compiler vs the ctypes mirrors in visfs_amd/abi.py."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "visfs_ba.h"\nint main(void){\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(visfs_ba_params), sizeof(visfs_ba_window), sizeof(visfs_ba_result),'
                   ' sizeof(visfs_ba_graph), sizeof(visfs_ba_stats), sizeof(visfs_ba_graph_info), sizeof(visfs_ba_profile));\n'
                   'printf("%zu %zu %zu %zu\\n", offsetof(visfs_ba_window, Trc), offsetof(visfs_ba_window, n_laser_points),'
                   ' offsetof(visfs_ba_result, chi2_final), offsetof(visfs_ba_stats, trace_chi2));\nreturn 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out]
    assert sizes[:7] == [C.sizeof(abi.Params), C.sizeof(abi.Window), C.sizeof(abi.Result), C.sizeof(abi.Graph),
                         C.sizeof(abi.Stats), C.sizeof(abi.GraphInfo), C.sizeof(abi.Profile)]
    assert sizes[7:] == [abi.Window.Trc.offset, abi.Window.n_laser_points.offset, abi.Result.chi2_final.offset, abi.Stats.trace_chi2.offset]


def test_default_params_are_the_reference_defaults(hiplib):
    p = abi.Params()
    hiplib.visfs_ba_default_params(C.byref(p))
    # Parameters.h:184-191
    assert (p.framework, p.solver, p.trust_region, p.iterations) == (0, 0, 0, 10)
    assert (p.pixel_variance, p.odometry_covariance, p.laser_covariance, p.robust_kernel_delta) == (1.5, 0.00005, 0.1, 8.0)


@pytest.mark.parametrize("which", ["C1", "C3small", "ragged", "filters"])
def test_product_graph_build_is_bit_identical_to_the_oracle(hiplib, olib, which):
    if which == "C1":
        w = synth.make_window("C1")
    elif which == "C3small":
        w = synth.make_window("C3", n_kf=12, n_lm=300, n_obs=2400)
    elif which == "ragged":
        w = ragged_window(seed=5)
    else:
        w = synth.make_window("C1")
        w["ref_depth"] = np.asarray(w["ref_depth"]).copy(); w["ref_depth"][[5, 50, 500]] = [np.nan, -2.0, 0.0]
        w["ref_pose"] = np.asarray(w["ref_pose"]).copy(); w["ref_pose"][77] = 4242
        w["link_from"] = np.array([1, 2, 3, 3, 0, 99], np.uint64); w["link_to"] = np.array([2, 3, 3, 5, 4, 2], np.uint64)
        w["link_T"] = np.tile(np.eye(3, 4).reshape(1, 12), (6, 1)) + 0.01 * np.arange(72).reshape(6, 12)
    prm = abi.default_params()
    a = graph_of(hiplib.visfs_ba_pack_window, prm, w)
    b = graph_of(olib.oracle_pack_window, prm, w)
    ga, gb = a[1], b[1]
    for f in ("pose_tq", "pose_fixed", "obs_point", "obs_pose", "obs_uvr", "odo_from", "odo_to", "odo_tq"):
        assert np.array_equal(getattr(ga, f), getattr(gb, f)), f
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4]
    for f in ("fx", "fy", "cx", "cy", "bf"):
        assert getattr(ga.struct, f) == getattr(gb.struct, f)
    if which == "filters":
        assert a[4] == 3 and ga.struct.n_odo == 3 and ga.n_obs == 3000 - 4      # 3 bad depths, 1 unknown pose; self/unknown/zero-id links dropped


def test_product_write_back_matches_oracle(hiplib, olib):
    rng = np.random.default_rng(0)
    q = rng.normal(size=(20, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    tq = np.hstack([rng.normal(size=(20, 3)), q])
    a = twr_of(hiplib.visfs_ba_unpack_pose, tq, synth.TRC)
    b = twr_of(olib.oracle_unpack_pose, tq, synth.TRC)
    assert np.abs(a - b).max() < 1e-15
    # round trip through the graph build: Twr → Tcw (pack) → Twr (unpack)
    w = synth.make_window("C1")
    _, gb, *_ = graph_of(hiplib.visfs_ba_pack_window, abi.default_params(), w)
    back = twr_of(hiplib.visfs_ba_unpack_pose, gb.pose_tq, w["Trc"])
    assert np.abs(back - np.asarray(w["pose_Twr"])).max() < 1e-13


def test_pack_rejects_unsorted_references(hiplib):
    w = synth.make_window("C1")
    for k in ("ref_feature", "ref_pose", "ref_u", "ref_v", "ref_depth"):
        w[k] = np.asarray(w[k])[::-1].copy()
    with pytest.raises(RuntimeError):
        graph_of(hiplib.visfs_ba_pack_window, abi.default_params(), w)


def test_no_cpu_fallback_without_a_gpu(hiplib):
    """Without a gfx950 device the product must fail loudly (VISFS_BA_ERR_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    p = abi.default_params()
    assert hiplib.visfs_ba_create(C.byref(p), 0, C.byref(h)) == abi.ERR_DEVICE and not h.value
    from visfs_amd import backend
    with pytest.raises(backend.BackendError):
        backend.Solver(p)


def test_missing_library_raises(monkeypatch):
    from visfs_amd import backend
    monkeypatch.setattr(backend, "_lib", None)
    monkeypatch.setattr(backend, "LIB_PATH", "/nonexistent/libvisfs_ba_hip.so")
    with pytest.raises(backend.BackendError):
        backend.load_library()


def test_product_never_references_the_oracle():
    """The product tree (visfs_amd/, include/) must not import, link or name the oracle."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "visfs_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle_lib|libvisfs_ba_oracle|visfs_ba_oracle\.h|oracle_sys|oracle_solve", text):
                    bad.append(f)
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(ROOT, "visfs_amd", "lib", "libvisfs_ba_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out
