"""Sharding of independent windows over ranks (BASELINE config 5) — plumbing around torch.distributed.

The BA path shards embarrassingly: every window is a self-contained localOptimize call
(reference: corelib/include/Optimizer/Optimizer.h:64-71, stateless between calls), so there is
NO data-path collective.  The only collectives are the barrier / max-over-ranks of the bench
contract and an optional all_gather of the small per-window results.
"""
import os


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def self_launch(script, argv, n_ranks, python=None, timeout=None):
    """`python script --gpus N ...` started PLAINLY (no torchrun environment): start the N ranks as a CHILD process —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> script argv...` —
    relay its stdout / stderr and return its exit code.  Never an exec: the caller must not have touched the GPU (and does not need
    to: it only waits), the ranks are its grandchildren.  The port is one the kernel has just handed out as free."""
    import socket
    import subprocess
    import sys
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env["VISFS_BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.run(cmd, env=env, timeout=timeout)      # stdout / stderr inherited: rank 0's JSON line goes straight through
    return proc.returncode


def shard_windows(n_windows, rank, world_size):
    """Contiguous blocks: window w → rank w // ceil(n/world). Returns the window indices owned by `rank`."""
    per = (n_windows + world_size - 1) // world_size
    return list(range(rank * per, min(n_windows, (rank + 1) * per)))


def init_process_group(backend, rank, world_size):
    import torch.distributed as dist
    if world_size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)


def shutdown(world_size):
    """Tear the process group down once the last collective is done (avoids the resource-leak warning at exit)."""
    if world_size > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def barrier(world_size, device=None):
    if world_size > 1:
        import torch.distributed as dist
        if device is not None:
            dist.barrier(device_ids=[device])
        else:
            dist.barrier()


def reduce_max(value, world_size, device="cpu"):
    """max over ranks of a python float (the bench contract's timing rule)."""
    if world_size == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, world_size, device="cpu"):
    if world_size == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_results(local, world_size, device="cpu"):
    """all_gather of per-window result rows (fp64 tensor [n_local, k]); every rank solves the same number of windows."""
    import torch
    if world_size == 1:
        return local
    import torch.distributed as dist
    local = local.to(device)
    out = [torch.empty_like(local) for _ in range(world_size)]
    dist.all_gather(out, local)
    return torch.cat(out, 0)
