#!/usr/bin/env python3
"""bench.py — BA iterations/sec of the MI355X sliding-window bundle-adjustment backend.

Contract (one JSON line on rank 0):
  metric  = BASELINE.json's "BA iterations/sec (50 KF, 5k pts, 50k obs) @1 GPU"
  step    = one full two-phase optimise (Optimizer.cpp:261-318) of every window resident on this rank,
            graph already in HBM when the timed region starts (reset of the estimates included);
  value   = outer LM iterations executed by ALL ranks / max-over-ranks wall time  (weak scaling: each
            rank owns `--windows-per-gpu` independent windows, no data-path collective).
Also reported: `roofline` of the dominant kernel (HIP-event durations measured in the timed region on
the library's own stream) and `cpu_baseline` = the CPU oracle timed on this box (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np


def algorithmic_bytes(kernel, d):
    """SURVEY.md §8(d) per-unit bytes x units of one launch, averaged over the launches of one solve: a launch
    of phase p touches the d["edges"][p] stereo edges that are still in the active set (phase 2 runs without the culled ones), and
    a solve has d["lin_launches"][p] linearisations and d["trial_launches"][p] damped solves in phase p."""
    Nl, Np, nblk = d["n_points"], d["n_poses"], d["n_blk"]
    edges, lin, tr = d["edges"], d["lin_launches"], d["trial_launches"]

    def avg(weights, fn):
        tot = sum(weights)
        return sum(wt * fn(e) for wt, e in zip(weights, edges)) / tot if tot > 0 else fn(edges[0])
    # The fused speculative unit (unit_form 2) moves stage A of an accepted trial into the two launches around it: its
    # landmark-major half (the point Jacobians, weights and errors: 72 + 40 B/obs, 96 B/landmark) into the k_backsub launch, its
    # pose-major half (the pose Jacobians: 144 B/obs, 336 B/pose) behind the next k_schur_partial launch; k_linearize itself only runs
    # at the first iteration of a phase.  Per launch of the class: lin[p] - 1 of a phase's tr[p] launches carry the extra half.
    fused = d.get("unit_form", 0) == 2
    extra = [max(l - 1, 0) / t if t > 0 else 0.0 for l, t in zip(lin, tr)]

    def avg_x(fn, fx):
        tot = sum(tr)
        return sum(wt * (fn(e) + (x * fx(e) if fused else 0.0)) for wt, e, x in zip(tr, edges, extra)) / tot if tot > 0 else fn(edges[0])
    if kernel == "k_linearize":      # stage A of §8d: 256 B/obs + 96 B/landmark + 336 B/pose
        return avg(lin, lambda No: 256 * No + 96 * Nl + 336 * Np)
    if kernel == "k_schur_partial":  # stage B: 144 B/obs (Hpl) + 96 B/landmark + 288 B/stored block + 48 B/pose
        return avg_x(lambda No: 144 * No + 96 * Nl + 288 * nblk + 48 * Np, lambda No: 144 * No + 336 * Np)
    if kernel == "k_backsub":        # stages D+E: 144 + 112 + 8 B/obs, (72+24+24)+48 B/landmark, 112 B/pose
        return avg_x(lambda No: 264 * No + 168 * Nl + 112 * Np, lambda No: 112 * No + 96 * Nl)
    if kernel == "k_pcg":            # stage C, k iterations in one launch: k * (288 B/block + 4*48 B/pose)
        return d["pcg_iters_per_launch"] * (288 * nblk + 192 * Np)
    if kernel == "k_direct":         # direct solve of the reduced system: every stored block of S once, b_s in, x out
        return 288 * nblk + 2 * 48 * Np
    raise KeyError(kernel)


def phase_counts(d, st):
    """Per-phase launch and edge counts of one solve from its visfs_ba_stats (ABI 3)."""
    if st is None:
        d["edges"] = [d["n_obs"], d["n_obs"]]; d["lin_launches"] = [1, 1]; d["trial_launches"] = [1, 1]; d["pcg_phase"] = [0, 0]
        return d
    d["edges"] = [int(st.n_active_edges[0]), int(st.n_active_edges[1])]
    d["lin_launches"] = [int(st.iterations_run[0]), int(st.iterations_run[1])]
    d["trial_launches"] = [int(st.trials_run[0]), int(st.trials_run[1])]
    d["pcg_phase"] = [int(st.pcg_iterations_phase[0]), int(st.pcg_iterations_phase[1])]
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C2", help="C1..C5 / PROD (BASELINE.json configs); C2 is the headline; C4R = C4 started from 1e-3 rad rotation error (phase 1 converges), C4C = C4 with the world origin at the trajectory centroid")
    ap.add_argument("--windows-per-gpu", type=int, default=1)
    ap.add_argument("--solver", type=int, default=2, help="Optimizer/Solver: 2 = PCG (headline), 0 = direct Cholesky")
    ap.add_argument("--framework", type=int, default=0, help="Optimizer/Framework: 0 = the g2o branch (headline), 1 = the Ceres branch (one pass of <= Iterations trust-region iterations, direct solver)")
    ap.add_argument("--trust-region", type=int, default=0, help="Optimizer/TrustRegion: g2o branch 0 = Levenberg, 1 = Gauss-Newton; Ceres branch 0 = LEVENBERG_MARQUARDT, 1 = DOGLEG")
    ap.add_argument("--iterations", type=int, default=20, help="Optimizer/Iterations (10+10, as the shipped launch files)")
    ap.add_argument("--batch-mode", default="launch", choices=("launch", "streams"),
                    help="--windows-per-gpu > 1: 'launch' = one sequence of batched launches for all resident windows "
                         "(blockIdx.y = window), 'streams' = one host thread + HIP stream per window")
    ap.add_argument("--handles", type=int, default=1, help="with --windows-per-gpu B > 1 in launch mode: split the B resident windows over this many handles "
                    "(each its own stream and host thread, B / handles windows sharing every launch of a handle)")
    ap.add_argument("--tuning", choices=("auto", "latency", "throughput"), default="auto",
                    help="visfs_ba_set_tuning of the handles: auto = throughput for handles that hold a batch of resident windows (BASELINE config 5: the PCG "
                         "of a window in one workgroup), latency for a window on its own (the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config5", choices=("auto", "on", "off"), default="auto",
                    help="also measure BASELINE config 5's per-GPU share in the same run (8 resident C2-size windows per GPU sharing every launch, "
                         "results all_gather'ed and checked): 'auto' = on, so every driver-run line (N = 1 included) carries a config-5 number")
    args = ap.parse_args()

    from visfs_amd import dist as vdist
    rank, local_rank, world = vdist.env_rank()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): the N ranks run as a CHILD process group under torch.distributed.run; this
        # process has not touched (and never touches) a GPU, relays rank 0's JSON line and exits with the child's code
        raise SystemExit(vdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if world != args.gpus:
        # never a silent single-GPU line for a multi-GPU request
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    if os.environ.get("VISFS_BENCH_DRY_RUN") == "1":
        return dry_run(args, rank, world, vdist)
    from visfs_amd import abi, synth, backend
    import torch
    dev = local_rank if world > 1 else 0
    # rehearsal on a 1-GPU box: VISFS_BENCH_DEVICE pins every rank to one device, VISFS_BENCH_DIST_BACKEND=gloo replaces RCCL
    if os.environ.get("VISFS_BENCH_DEVICE") is not None:
        dev = int(os.environ["VISFS_BENCH_DEVICE"])
    dist_backend = os.environ.get("VISFS_BENCH_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    torch.cuda.set_device(dev)
    if world > 1:
        vdist.init_process_group(dist_backend, rank, world)
    red_dev = f"cuda:{dev}" if (world > 1 and dist_backend == "nccl") else "cpu"
    bar_dev = dev if (world > 1 and dist_backend == "nccl") else None

    prm = abi.default_params(iterations=args.iterations, solver=args.solver, framework=args.framework, trust_region=args.trust_region)
    if args.framework == 1:
        args.solver = 0                                                                   # the branch's dense solvers: Schur + direct Cholesky
    lib = backend.load_library()
    B = args.windows_per_gpu
    solvers, descs, gbs = [], [], []
    batched = B > 1 and args.batch_mode == "launch"
    tuning = {"latency": abi.TUNE_LATENCY, "throughput": abi.TUNE_THROUGHPUT}.get(args.tuning, abi.TUNE_THROUGHPUT if batched else abi.TUNE_LATENCY)
    for b in range(B):
        w = synth.make_window(args.config, window_index=rank * B + b)
        wb = abi.WindowBuffers(w)
        gb, used, oref, mono = abi.pack_window_with(lib.visfs_ba_pack_window, prm, wb)    # host graph build (product code)
        gbs.append(gb)
        if not batched or b == 0:
            s = backend.Solver(prm, device=dev, tuning=tuning)
            s.upload(gb)                                                                  # inputs resident in HBM
            solvers.append(s)
            descs.append(s.describe())
    H = max(1, min(args.handles, B)) if batched else 1
    if batched:
        per = (B + H - 1) // H
        for k in range(1, H):
            solvers.append(backend.Solver(prm, device=dev, tuning=tuning))
        for k in range(H):
            solvers[k].batch_upload(gbs[k * per:(k + 1) * per])                           # the handle's windows resident side by side

    pool = None
    if batched and H > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=H)
    if B > 1 and not batched:
        # independent windows: one host thread per resident window, each on its own HIP stream (ctypes drops the GIL),
        # so their kernels overlap on the device — BASELINE config 5 puts 8 windows on every GPU
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=B)

    def solve_one(s):
        s.reset()
        rc, st = s.optimize()
        assert rc == abi.OK, rc
        return st.iterations_run[0] + st.iterations_run[1], st

    def step():
        if B == 1:
            return solve_one(solvers[0])
        if batched and H == 1:
            solvers[0].batch_reset()
            rc, stats = solvers[0].batch_optimize()
            assert rc == abi.OK, rc
            return sum(st.iterations_run[0] + st.iterations_run[1] for st in stats), stats[0]
        if batched:
            def run(s):
                s.batch_reset()
                rc, stats = s.batch_optimize()
                assert rc == abi.OK, rc
                return sum(st.iterations_run[0] + st.iterations_run[1] for st in stats), stats[0]
            res = list(pool.map(run, solvers[:H]))
            return sum(r[0] for r in res), res[0][1]
        res = list(pool.map(solve_one, solvers))
        return sum(r[0] for r in res), res[0][1]

    for _ in range(args.warmup):
        step()
    # untimed calibration pass: a HIP-event pair on every kernel launch → per-kernel breakdown + the dominant kernel
    data_kernels = ("k_linearize", "k_schur_partial", "k_backsub", "k_pcg", "k_direct")
    solvers[0].profile_enable(True)
    for _ in range(3):
        step()
    calib = solvers[0].profile_read()
    null_us = 1e3 * getattr(solvers[0], "null_pair_ms", 0.0)     # an empty event pair: the mechanism's own share of every duration
    cand = [k for k in data_kernels if calib.get(k, {}).get("active_launches", 0) > 0]
    dom = max(cand, key=lambda k: calib[k]["active_ms"]) if cand else None
    # timed region: no instrumentation at all (event bookkeeping is host work between launches)
    solvers[0].profile_enable(False)
    vdist.barrier(world, bar_dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    last = None
    for k in range(args.steps):
        n, last = step()
        iters += n
    torch.cuda.synchronize()
    vdist.barrier(world, bar_dev)
    t1 = time.perf_counter()
    elapsed = vdist.reduce_max(t1 - t0, world, red_dev)
    total_iters = vdist.reduce_sum(iters, world, red_dev)
    graph_replay = bool(solvers[0].describe().get("graph_replayed", 0)) if B == 1 else False    # how the LAST timed solve was launched
    # the dominant kernel's launch durations: the SAME steps once more, right behind the timed region, with a HIP-event pair
    # attached to every launch of that kernel on the library's own stream (max(3, steps / 8) steps)
    if dom:
        solvers[0].profile_enable([dom])
        for _ in range(max(3, args.steps // 8)):
            step()
        solvers[0].profile_enable(False)
    # config 5 as north_star words it: after the timed region every rank's window results travel to every rank in ONE
    # all_gather (RCCL over xGMI when the backend is nccl); rank 0 checks them against its own single-rank solves of the same windows
    gathered = gather_and_check(args, prm, lib, solvers, batched, B, rank, world, dev, red_dev, vdist, abi, synth, backend, torch)
    c5 = None
    # 'auto' = always (round 4): the driver's plain `bench.py --gpus 1` line then carries a driver-timed config-5 number too
    if args.config5 in ("on", "auto"):
        # (the headline line must survive whatever happens in this extra leg: every rank takes the same path through its collectives
        # unless its own solve fails, and a failure is reported instead of raised)
        try:
            c5 = config5_record(args, lib, rank, world, dev, red_dev, bar_dev, vdist, abi, synth, backend, torch)
        except Exception as e:          # noqa: BLE001
            c5 = {"error": f"{type(e).__name__}: {e}"[:200]}
    vdist.shutdown(world)                  # last collective done

    if rank != 0:
        return
    prof = solvers[0].profile_read()
    d = descs[0]
    roofline = None
    d["pcg_iters_per_launch"] = (last.pcg_iterations / max(1, last.trials_run[0] + last.trials_run[1])) if last is not None else 0
    phase_counts(d, last)

    def roof(kernel, p):
        # avg_launch_us: HIP events ATTACHED to the launch (hipExtLaunchKernelGGL start / stop events = the dispatch's own
        # timestamps, what rocprofv3 reports: profiles/*kernel_stats.csv agrees within a few per cent).  Kernel classes of several
        # launches (direct solver, linearise with odometry) are still bracketed by a pair recorded on the stream, which carries
        # the event mechanism's own share: event_pair_null_us = what an EMPTY recorded pair measures there.
        avg_us = 1e3 * p["active_ms"] / p["active_launches"]
        byts = algorithmic_bytes(kernel, d)
        achieved = byts / (avg_us * 1e-6) / 1e9
        symbol = {1: "k_pcg1", 2: "k_pcg", 3: "k_pcg", 4: "k_pcg_cu", 5: "k_small_solve", 6: "k_chol_update + k_chol_solve + ...", 7: "k_band_chol"}.get(d.get("solver_kernel"), kernel) \
            if kernel in ("k_pcg", "k_direct") else kernel
        return {"bound": "hbm", "kernel": kernel, "kernel_symbol": symbol, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": pmc_traffic(args.config, kernel), "bytes_per_launch": byts,
                "avg_launch_us": round(avg_us, 3), "event_pair_null_us": round(null_us, 3), "launches": p["active_launches"]}

    if dom and dom in prof and prof[dom]["active_launches"] > 0:
        roofline = roof(dom, prof[dom])
    # the other data-path kernels, from the calibration pass (three instrumented steps before the timed region)
    roofline_kernels = [roof(k, calib[k]) for k in cand if k != dom]
    out = {
        "metric": "BA iterations/sec (50 KF, 5k pts, 50k obs) @1 GPU; max-pose-err vs g2o",
        "value": round(total_iters / elapsed, 2), "unit": "BA iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: {d['n_poses']} KF / {d['n_points']} landmarks / {d['n_obs']} stereo observations"
                               f" / {d['n_odo']} odometry edges, Schur + {'PCG' if args.solver == 2 else 'direct Cholesky'}, "
                               + (f"Iterations={args.iterations} ({args.iterations // 2}+{args.iterations // 2})" if args.framework == 0 else
                                  f"Optimizer/Framework=1 (Ceres branch, {'DOGLEG' if args.trust_region == 1 else 'LEVENBERG_MARQUARDT'}: one pass, <= {args.iterations} trust-region iterations, each counted as one BA iteration)")
                               + f", {B} window(s) per GPU",
                   "windows_per_gpu": B, "solver": args.solver, "iterations_per_solve": int(total_iters / (args.steps * world * B)),
                   "pcg_iterations_per_solve": int(last.pcg_iterations) if last is not None else 0,
                   "tuning": "throughput" if tuning == abi.TUNE_THROUGHPUT else "latency",
                   "parallelism": f"{world} rank(s) x {B} independent window(s), no data-path collective"
                                  + (f"; windows of a rank share every launch (blockIdx.y = window)" if batched else "")
                                  + (f"; {H} handles (streams / host threads) of {(B + H - 1) // H} windows each" if batched and H > 1 else "")},
        "roofline": roofline,
        "roofline_other_kernels": roofline_kernels,
        # the headline re-optimises ONE resident graph: from its second solve on the library replays the launch sequence as a hipGraph
        # (C2 +2 %, PROD +9 %); a per-frame visfs_ba_solve_window never gets that — its rate is in profiles/*e2e_breakdown.log
        "graph_replay": graph_replay,
        "result_gather": gathered,
        "config5": c5,
        "kernel_us_per_step_calibration": {k: round(1e3 * v["total_ms"] / 3, 2) for k, v in calib.items()},
    }
    if last is not None and args.solver == 2:
        # whole-iteration view: bytes of one solve = sum over its two phases of (iterations x linearise bytes + damped
        # solves x (Schur + back-substitution + PCG bytes)), each phase with ITS active edge count (phase 2 runs without the edges the
        # outlier pass moved to level 1), against the cache-agnostic floor (bytes / 8 TB/s); rank 0's window stands for all
        its = max(1, last.iterations_run[0] + last.iterations_run[1])
        tr = max(1, last.trials_run[0] + last.trials_run[1])
        Nl, Np, nb = d["n_points"], d["n_poses"], d["n_blk"]
        solve_bytes = 0.0
        for ph in range(2):
            No_p, it_p, tr_p, k_p = d["edges"][ph], d["lin_launches"][ph], d["trial_launches"][ph], d["pcg_phase"][ph]
            solve_bytes += it_p * (256 * No_p + 96 * Nl + 336 * Np) + tr_p * (408 * No_p + 264 * Nl + 160 * Np + 288 * nb) + k_p * (288 * nb + 192 * Np)
        byts = solve_bytes / its
        rate = byts * (total_iters / elapsed) / 1e9
        out["iteration_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_iteration": int(byts), "trials_per_iteration": round(tr / its, 3),
                                     "pcg_iterations_per_solve": round(last.pcg_iterations / tr, 3),
                                     "active_edges_per_phase": d["edges"], "iterations_per_phase": d["lin_launches"], "damped_solves_per_phase": d["trial_launches"],
                                     "achieved": round(rate, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(rate / 8000.0, 4), "floor_us_per_iteration": round(byts / 8e6, 2),
                                     "window_us_per_iteration": round(1e6 * elapsed * world * B / total_iters, 2),
                                     "aggregate_us_per_iteration_per_gpu": round(1e6 * elapsed * world / total_iters, 2)}
    if world == 1 and B == 1:
        # the drop-in's real call (Estimator.cpp:254): host buffers in, results out, one handle across frames — reported beside the
        # resident headline, never as `value` (it crosses PCIe both ways)
        try:
            out["per_frame_call"] = per_frame_call(args, prm)
        except Exception as e:
            out["per_frame_call"] = {"error": str(e)[:120]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, prm)
        # parity of the timed configuration against the oracle (max pose error metric of BASELINE.json)
        out["max_pose_err_vs_oracle"] = parity_vs_oracle(args, prm, solvers[0])
    print(json.dumps(out))


def dry_run(args, rank, world, vdist):
    """VISFS_BENCH_DRY_RUN=1 (CPU test of the launch path, tests/test_bench_launch.py): the ranks rendezvous over gloo, run the
    contract's collectives on dummy numbers and rank 0 prints a line — no GPU, no solver."""
    import torch
    if world > 1:
        vdist.init_process_group("gloo", rank, world)
    vdist.barrier(world)
    elapsed = vdist.reduce_max(1.0 + rank, world)
    total = vdist.reduce_sum(10.0, world)
    ids = vdist.gather_results(torch.tensor([[float(rank)]], dtype=torch.float64), world).numpy().ravel()
    vdist.barrier(world)
    vdist.shutdown(world)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "max_elapsed": elapsed, "sum": total,
                          "ranks_seen": [int(x) for x in ids], "self_launched": os.environ.get("VISFS_BENCH_SELF_LAUNCHED") == "1"}))


def config5_record(args, lib, rank, world, dev, red_dev, bar_dev, vdist, abi, synth, backend, torch, per_gpu=8):
    """BASELINE config 5 (64 independent 50-KF windows, 8 per GPU): this rank keeps windows rank * 8 .. rank * 8 + 7 of the C5 set
    resident and solves them as one batch (blockIdx.y = window); timed like the headline (barrier + synchronize on both sides, max
    over ranks); afterwards one all_gather of every window's poses, rank 0 re-solves the first window of every rank alone and
    compares bit for bit."""
    import time as _t
    prm = abi.default_params(iterations=20, solver=2)
    gbs = []
    for b in range(per_gpu):
        w = synth.make_window("C5", window_index=rank * per_gpu + b)
        gbs.append(abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))[0])
    tuning = abi.TUNE_LATENCY if args.tuning == "latency" else abi.TUNE_THROUGHPUT       # (a handle that holds a batch: visfs_ba_set_tuning)
    s = backend.Solver(prm, device=dev, tuning=tuning)
    s.batch_upload(gbs)

    def step():
        s.batch_reset()
        rc, stats = s.batch_optimize()
        assert rc == abi.OK, rc
        return sum(st.iterations_run[0] + st.iterations_run[1] for st in stats)
    for _ in range(3):
        step()
    steps = max(5, args.steps // 4)
    vdist.barrier(world, bar_dev); torch.cuda.synchronize()
    t0 = _t.perf_counter()
    iters = 0
    for _ in range(steps):
        iters += step()
    torch.cuda.synchronize(); vdist.barrier(world, bar_dev)
    elapsed = vdist.reduce_max(_t.perf_counter() - t0, world, red_dev)
    total = vdist.reduce_sum(iters, world, red_dev)
    local = torch.from_numpy(np.stack([s.batch_download(b)[0] for b in range(per_gpu)], 0))
    rec = {"workload": f"{per_gpu} resident C2-size windows per GPU x {world} GPU(s) = {per_gpu * world} windows, Schur + PCG, Iterations=20, batched launches",
           "tuning": "throughput" if tuning == abi.TUNE_THROUGHPUT else "latency",
           "value": round(total / elapsed, 2), "unit": "BA iterations/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 4), "scaling": "weak"}
    if world > 1:
        ids = vdist.gather_results(torch.tensor([[float(rank)]], dtype=torch.float64), world, red_dev).cpu().numpy().ravel()
        allp = vdist.gather_results(local, world, red_dev).cpu().numpy()
    else:
        ids, allp = np.array([0.0]), local.numpy()
    if rank == 0:
        same = True
        for r in range(world):
            w = synth.make_window("C5", window_index=r * per_gpu)
            gb = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))[0]
            s1 = backend.Solver(prm, device=dev, tuning=tuning)
            s1.batch_upload([gb]); s1.batch_reset(); s1.batch_optimize(); ref = s1.batch_download(0)[0]
            s1.close()
            same = same and bool(np.array_equal(ref, allp[r * per_gpu]))
        rec.update({"ranks_seen": [int(x) for x in ids], "windows": int(allp.shape[0]), "bit_identical": same})
    s.close()
    return rec if rank == 0 else None


def gather_and_check(args, prm, lib, solvers, batched, B, rank, world, dev, red_dev, vdist, abi, synth, backend, torch):
    """One all_gather of every window's optimised poses ([B, Np, 7] fp64 per rank) over the bench's process group; rank 0 re-solves
    the FIRST window of every rank itself (same chunking: as a batch member when the ranks ran batches) and compares bit for bit —
    the kernels are run-to-run and device-to-device deterministic (fixed-order reductions, no floating-point atomics)."""
    poses = []
    for b in range(B):
        if batched:
            per = (B + max(1, min(args.handles, B)) - 1) // max(1, min(args.handles, B))     # windows per handle (--handles)
            pose = solvers[b // per].batch_download(b % per)[0]
        else:
            pose = solvers[b].download()[0]
        poses.append(pose)
    local = torch.from_numpy(np.stack(poses, 0))
    if world == 1:
        return {"backend": None, "ranks_seen": [0], "windows": B, "bytes": int(local.numel() * 8), "checked_windows": 0, "bit_identical": None}
    ids = vdist.gather_results(torch.tensor([[float(rank)]], dtype=torch.float64), world, red_dev).cpu().numpy().ravel()
    allp = vdist.gather_results(local, world, red_dev).cpu().numpy()          # [world * B, Np, 7]
    if rank != 0:
        return None
    same, checked = True, 0
    for r in range(world):
        w = synth.make_window(args.config, window_index=r * B)
        gb, *_ = abi.pack_window_with(lib.visfs_ba_pack_window, prm, abi.WindowBuffers(w))
        tuning = {"latency": abi.TUNE_LATENCY, "throughput": abi.TUNE_THROUGHPUT}.get(args.tuning, abi.TUNE_THROUGHPUT if batched else abi.TUNE_LATENCY)
        s = backend.Solver(prm, device=dev, tuning=tuning)
        if batched:
            s.batch_upload([gb]); s.batch_reset(); s.batch_optimize(); ref = s.batch_download(0)[0]
        else:
            s.upload(gb); s.reset(); s.optimize(); ref = s.download()[0]
        s.close()
        same = same and bool(np.array_equal(ref, allp[r * B]))
        checked += 1
    import torch.distributed as dist
    return {"backend": dist.get_backend(), "ranks_seen": [int(x) for x in ids], "windows": int(allp.shape[0]),
            "bytes": int(allp.size * 8), "checked_windows": checked, "bit_identical": same}


def pmc_traffic(config, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this bench, gfx950 correction 2*FETCH applied — see the
    file's _method).  None when no counters were collected for this config / kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))[config]
        # the PCG class is one of three kernels by window size (k_pcg1: <= 64 free poses, k_pcg beyond, k_pcg_cu opt-in)
        names = [kernel] + (["k_pcg1", "k_pcg_cu"] if kernel == "k_pcg" else [])
        for n in names:
            if n in d:
                return round(d[n]["traffic_bytes"])
        return None
    except (KeyError, ValueError):
        return None


def per_frame_call(args, prm, calls=24):
    """visfs_ba_solve_window on ONE handle, `calls` times after 3 warm-up calls: graph build on the host threads, upload, index build,
    the whole optimisation, write-back.  The caller's input and output buffers exist before the clock starts."""
    from visfs_amd import abi, backend, synth
    w = synth.make_window(args.config, window_index=0)
    s = backend.Solver(prm)
    n = calls if len(w["ref_feature"]) < 200000 else max(6, calls // 3)
    wbs = [abi.WindowBuffers(w) for _ in range(n + 3)]
    rbs = [abi.ResultBuffers(wb.struct.n_poses, wb.struct.n_refs) for wb in wbs]
    for r in rbs:
        r.outlier_feature[:] = 1; r.outlier_pose[:] = 1            # touched pages
    for i in range(3):
        s.solve_window(wbs[i], rbs[i])
    each = []
    t0 = time.perf_counter()
    for i in range(n):
        t1 = time.perf_counter()
        rc, rb = s.solve_window(wbs[3 + i], rbs[3 + i])
        each.append(time.perf_counter() - t1)
    dt = (time.perf_counter() - t0) / n
    its = int(rb.struct.iterations_run[0] + rb.struct.iterations_run[1])
    s.close()
    return {"ms_per_call": round(1e3 * dt, 4), "ms_per_call_median": round(1e3 * float(np.median(each)), 4), "calls": n, "status": int(rc), "outer_iterations": its,
            "iterations_per_s_incl_pcie": round(its / dt, 1), "host_threads": os.environ.get("VISFS_BA_THREADS", "default")}


def _cpu_topology_pick(n):
    """n CPUs for the OpenMP baseline, packed into as few last-level-cache domains as possible, starting with the domain this process
    runs on; one hardware thread per core first (SMT siblings only when the allowed cores run out).  Returns (cpus, n_domains)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))

    def read_list(path):
        try:
            out = []
            for part in open(path).read().strip().split(","):
                if not part:
                    continue
                lo, _, hi = part.partition("-")
                out += list(range(int(lo), int(hi or lo) + 1))
            return out
        except (OSError, ValueError):
            return None
    dom_of, core_of = {}, {}
    for c in allowed:
        d = read_list(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list")
        sib = read_list(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")
        dom_of[c] = min(d) if d else 0
        core_of[c] = min(sib) if sib else c
    try:
        here = os.sched_getcpu() if hasattr(os, "sched_getcpu") else allowed[0]
    except OSError:
        here = allowed[0]
    doms = sorted(set(dom_of.values()), key=lambda d: (d != dom_of.get(here, -1), d))
    primary, siblings = [], []
    for d in doms:
        seen = set()
        for c in allowed:
            if dom_of[c] != d:
                continue
            (siblings if core_of[c] in seen else primary).append(c)
            seen.add(core_of[c])
    pick = (primary + siblings)[:n]
    return pick, len({dom_of[c] for c in pick})


def _time_oracle(olib, prm, gb, threads, pin, budget_s, min_runs=10, max_runs=40):
    """Median / spread of full solves of the resident window on `threads` threads (pin: CPU list for oracle_omp_pin, or None)."""
    import oracle_lib
    s = oracle_lib.OracleSystem(olib, prm, gb, threads)
    bound = 0
    if pin:
        arr = (C.c_int32 * len(pin))(*pin)
        bound = olib.oracle_omp_pin(arr, len(pin))
    secs, its = [], 0
    for _ in range(2):                                  # warm-ups (page faults, thread team start)
        s.reset(); s.optimize()
    t_end = time.perf_counter() + budget_s
    while len(secs) < min_runs or (time.perf_counter() < t_end and len(secs) < max_runs):
        s.reset()
        rc, st, sec = s.optimize()
        secs.append(sec); its = st.iterations_run[0] + st.iterations_run[1]
    s.close()
    if pin:
        olib.oracle_omp_unpin()
    secs = np.array(secs)
    return {"its": int(its), "runs": int(len(secs)), "median_s": float(np.median(secs)), "min_s": float(secs.min()), "max_s": float(secs.max()), "threads_bound": int(bound)}


def cpu_baseline(args, prm):
    """The CPU oracle ('port' of the g2o algorithm) timed on this box's host cores: bounded sample (>= 10 full solves each,
    median).  Two builds — x86-64-v3 (the checker's: AVX2, no contraction) and, where the CPU has AVX-512, x86-64-v4 with contraction
    into FMAs (what -O3 -march=native gives on the GPU boxes' Zen 5 hosts) — the faster one is the reported value.  The OpenMP team is
    PINNED: one thread per core, packed into as few L3 domains as the granted core count allows, starting with this process's own
    (unpinned, the same binary measured 181 / 673 / 465 it/s in rounds 1-3: VERDICT r03)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from visfs_amd import abi, synth
    import oracle_lib
    olib = oracle_lib.load()
    w = synth.make_window(args.config, window_index=0)
    wb = abi.WindowBuffers(w)
    gb, *_ = abi.pack_window_with(olib.oracle_pack_window, prm, wb)
    big = len(w["ref_feature"]) >= 200000
    builds = [("x86-64-v3", False)] + ([("x86-64-v4+fma", True)] if oracle_lib.cpu_has_avx512() else [])
    serial = {}
    for name, v4 in builds:
        lib = oracle_lib.load(v4=v4)
        serial[name] = _time_oracle(lib, prm, gb, 1, None, 4.0 if big else 6.0, min_runs=3 if big else 10)
    best = min(serial, key=lambda k: serial[k]["median_s"])
    r = serial[best]
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None                      # cgroup v2 CPU quota of this box, in cores (the GPU box gives 16 of its 256 hardware threads)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    out = {"value": round(r["its"] / r["median_s"], 2), "unit": "BA iterations/s", "cores": 1, "kind": "port",
           "cpu_model": cpu_model, "nproc": os.cpu_count(), "usable_cores": usable, "cgroup_cpu_quota_cores": quota,
           "build": best, "builds": {k: round(v["its"] / v["median_s"], 2) for k, v in serial.items()},
           "spread": {"runs": r["runs"], "min": round(r["its"] / r["max_s"], 2), "max": round(r["its"] / r["min_s"], 2)},
           "sample": f"{r['runs']} full solves of the same {args.config} window ({r['its']} outer iterations each), median; "
                     f"g2o-algorithm restatement in C (oracle/), single thread, gcc -O3, fastest of {list(serial)}"}
    # SURVEY §8d also asks for the OpenMP build of the same restatement over the host cores (g2o's own default is serial)
    try:
        # every core this process may use: the affinity mask, capped by the cgroup CPU quota (threads beyond the quota only get throttled)
        nthr = max(1, int(min(usable, quota) if quota else usable))
        pin, ndom = _cpu_topology_pick(nthr)
        omp = {}
        for name, v4 in builds:
            lib = oracle_lib.load(omp=True, v4=v4)
            omp[name] = _time_oracle(lib, prm, gb, nthr, pin, 3.0 if big else 4.0, min_runs=3 if big else 10)
        bo = min(omp, key=lambda k: omp[k]["median_s"])
        ro = omp[bo]
        out["openmp"] = {"value": round(ro["its"] / ro["median_s"], 2), "cores": nthr, "build": bo,
                         "builds": {k: round(v["its"] / v["median_s"], 2) for k, v in omp.items()},
                         "spread": {"runs": ro["runs"], "min": round(ro["its"] / ro["max_s"], 2), "max": round(ro["its"] / ro["min_s"], 2)},
                         "placement": f"{ro['threads_bound']} of {nthr} threads bound one per CPU {pin[:4]}..{pin[-1]} in {ndom} L3 domain(s), starting with this process's own",
                         "note": "OpenMP build of the same restatement; threads = usable cores (affinity mask capped by the cgroup CPU quota)"}
    except Exception as e:      # the OpenMP library is optional
        out["openmp"] = {"error": str(e)[:120]}
    return out


def parity_vs_oracle(args, prm, solver):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from visfs_amd import abi, synth
    import oracle_lib
    from helpers import twr_of
    olib = oracle_lib.load()
    w = synth.make_window(args.config, window_index=0)
    gb, *_ = abi.pack_window_with(olib.oracle_pack_window, prm, abi.WindowBuffers(w))
    o = oracle_lib.OracleSystem(olib, prm, gb, 1)
    o.optimize()
    po, pto, outo, _ = o.download(); o.close()
    solver.reset(); solver.optimize()
    pg, ptg, outg, _ = solver.download()
    et, er = synth.pose_errors(twr_of(olib.oracle_unpack_pose, pg, w["Trc"]), twr_of(olib.oracle_unpack_pose, po, w["Trc"]))
    return {"translation_rel": et, "rotation_rad": er, "landmark_abs_max": float(np.abs(ptg - pto).max()),
            "outlier_set_equal": bool(np.array_equal(outo, outg))}


if __name__ == "__main__":
    main()
