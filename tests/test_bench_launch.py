"""`python bench.py --gpus N` started PLAINLY (no torchrun environment) must launch its N ranks itself — as a child process, never by
replacing the running program — relay rank 0's JSON line and the exit code.  CPU test: the ranks run bench.py's
dry-run leg (gloo rendezvous + the contract's collectives on dummy numbers; no GPU, no solver)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VISFS_BENCH_DRY_RUN"] = "1"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_start_with_two_gpus_launches_two_ranks_and_relays_one_json_line():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["self_launched"] is True
    assert d["max_elapsed"] == 2.0 and d["sum"] == 20.0 and d["steps"] == 3 and d["warmup"] == 1     # the flags travelled to the ranks


def test_plain_start_with_one_gpu_stays_in_process():
    p = _run(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["self_launched"] is False


def test_a_failing_rank_fails_the_launcher():
    # (an unknown flag makes every rank exit with argparse's code 2: the launcher must not report success)
    p = _run(["--gpus", "2", "--no-such-flag"])
    assert p.returncode != 0


def test_a_world_size_that_contradicts_the_flag_is_refused():
    p = _run(["--gpus", "2"], extra_env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
