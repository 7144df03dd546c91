"""visfs_amd/csrc/worker_pool.hpp — the host threads of the window layer — on its own: 3 000 regions of random size, every task exactly
once (a stale worker must never take or repeat a task of a later region), with the workers pinned to the caller's cache domain (default)
and left to the scheduler.  CPU only; compiles a small driver with g++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pool_test(tmp_path_factory):
    exe = tmp_path_factory.mktemp("pool") / "pool_test"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "cpp", "pool_test.cpp"), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("workers,affinity", [(1, "1"), (3, "1"), (7, "1"), (3, "0")])
def test_every_task_of_every_region_runs_exactly_once(pool_test, workers, affinity):
    env = dict(os.environ, VISFS_BA_POOL_AFFINITY=affinity)
    r = subprocess.run([pool_test, str(workers)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok 3000"), r.stdout + r.stderr
