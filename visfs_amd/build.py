"""Build helper: hipcc for the gfx950 library (the test-side checker under oracle/ has its own Makefile)."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visfs_amd", "csrc")
LIB_DIR = os.path.join(ROOT, "visfs_amd", "lib")
LIB = os.path.join(LIB_DIR, "libvisfs_ba_hip.so")
SOURCES = ["ba_kernels.hip", "ba_api.cpp"]
HEADERS = ["ba_math.hpp", "ba_device.hpp", "ba_kernels.hpp", "worker_pool.hpp", os.path.join("..", "..", "include", "visfs_ba.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 → visfs_amd/lib/libvisfs_ba_hip.so (in-tree, travels with gpurun)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _stale(LIB, deps):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + srcs + ["-lpthread"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)     # a normal build takes ~30 s
    if verbose or res.returncode != 0:
        print(" ".join(cmd)); print(res.stdout); print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr)
    return LIB



HOST = os.path.join(ROOT, "visfs_amd", "host")
WINDOW_LIB = os.path.join(LIB_DIR, "libvisfs_window.so")
WINDOW_SOURCES = ["WindowMap.cpp", "window_capi.cpp"]


def build_host(force=False, verbose=False):
    """g++ → visfs_amd/lib/libvisfs_window.so: the host-side sliding-window container (include/visfs_window.h), no GPU code."""
    srcs = [os.path.join(HOST, s) for s in WINDOW_SOURCES]
    deps = srcs + [os.path.join(HOST, "WindowMap.h"), os.path.join(ROOT, "include", "visfs_window.h"), os.path.join(ROOT, "include", "visfs_ba.h")]
    if not force and not _stale(WINDOW_LIB, deps):
        return WINDOW_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-o", WINDOW_LIB] + srcs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(" ".join(cmd)); print(res.stdout); print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stderr)
    return WINDOW_LIB
