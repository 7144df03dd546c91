"""examples/estimator_step.cpp: the estimator's BA step in plain C++ on the two C ABIs (window container + optimiser)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def example(tmp_path_factory, hiplib):
    from visfs_amd import build
    build.build_host()
    exe = str(tmp_path_factory.mktemp("example") / "estimator_step")
    libdir = os.path.join(ROOT, "visfs_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "estimator_step.cpp"), "-L" + libdir, "-lvisfs_window", "-lvisfs_ba_hip",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True)
    return exe


def test_example_compiles_and_refuses_to_run_without_gpu(example):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([example, "3"], capture_output=True, text=True)
    assert res.returncode == 3 and "gfx950" in res.stderr        # visfs_ba_create fails loudly: there is no CPU solve path


@pytest.mark.gpu
def test_example_runs_the_estimator_loop(example):
    res = subprocess.run([example, "40"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["frames"] == 40 and out["solved"] >= 30            # the window is full from frame 6 on
    assert out["signatures"] == 5 and out["features"] > 50 and out["last_chi2"] > 0
