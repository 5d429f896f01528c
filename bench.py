#!/usr/bin/env python3
"""Benchmark of the Frenetix-Occlusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode full|pair|reduced] [--M 10000] [--A 256]

One "step" = one pass of the per-planning-step hot path over one synthetic batch that is already resident in HBM:
agent-table preparation + trajectory-tile preparation + the trajectory x agent metric sweep (DCE/TTC/TTCE/WTTC/CP/
harm/risk) + threshold reduction (+ one RCCL all-gather of the per-trajectory cost vectors when N > 1).
Workload at N = 1: BASELINE.json configs[2] (synthetic 10 000 trajectories x 256 phantom predictions, T = 31), the
configuration the north-star target is quoted on.  N > 1: every rank gets its own 10 000-trajectory shard (weak
scaling), agents are replicated, cost vectors are all-gathered.

Prints ONE JSON line on rank 0.  `value` = trajectory x agent metric evaluations per second over all ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(M, A, T, mode):
    """float64 storage.  Inputs read once, outputs written once (DESIGN.md 'Algorithmic bytes')."""
    traj_in = M * T * 6 * 8              # tile table the sweep kernel reads: x, y, cos, sin, theta, v
    agent_in = A * (T * 8 * 8 + 8 * 8)   # agent table rows + per-agent constants
    out = M * (16 * 8 + 1)               # cost vector + safe flag (written by the reduce kernel; counted with the path)
    if mode in ("pair", "full"):
        out += M * A * (12 * 8 + 4 * 4)
    if mode == "full":
        out += M * A * 5 * (T - 1) * 8
    return traj_in + agent_in + out


def cpu_baseline(S, traj, agents, M_sample, threads):
    from oracle import fo_oracle as O  # checker / baseline only
    O.build()
    sub = {k: v[:M_sample] for k, v in traj.items()}
    O.sweep({k: v[:8] for k, v in sub.items()}, agents, S.VEHICLE_BMW320I, 0.1, nthreads=threads)  # warm-up
    best = float("inf")
    bufs = None
    for _ in range(3):  # first pass page-faults the output buffers; they are reused afterwards
        t0 = time.perf_counter()
        bufs = O.sweep(sub, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.1, "risk": 1}, want_lists=True,
                       nthreads=threads, out=bufs)
        best = min(best, time.perf_counter() - t0)
    A = agents["pos"].shape[0]
    return {"value": M_sample * A / best, "unit": "pair-evals/s", "cores": threads, "kind": "port",
            "sample": f"first {M_sample} trajectories x {A} agents of the same batch, full outputs, "
                      f"oracle/fo_oracle.c with OpenMP over trajectories, best of 3 with reused output buffers ({best:.2f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="full", choices=["full", "pair", "reduced"])
    ap.add_argument("--M", type=int, default=10000)
    ap.add_argument("--A", type=int, default=256)
    ap.add_argument("--T", type=int, default=31)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--order", default="sampler", choices=["sampler", "random"],
                    help="row order of the synthetic trajectories (synthetic.make_trajectories)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    M, A, T = args.M, args.A, args.T
    traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 3 + 1000 * rank, order=args.order)   # config 3; one shard per rank
    agents = S.make_agents(A, T, 0.1, seed=20240131 + 3)                      # replicated
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, device=local_rank)
    sw.reserve(M, T, A, T)
    d = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a)).to(dev, dt)
    tx, ty, tth, tv, ta = (d(traj[k]) for k in ("x", "y", "theta", "v", "a"))
    ag = [d(agents[k]) for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims")] + \
         [d(agents["type"], torch.int32), d(agents["len"], torch.int32)]
    out = None
    gathered = torch.empty((world * M, 16), dtype=torch.float64, device=dev) if world > 1 else None

    def step():
        nonlocal out
        sw.set_agents(*ag, check=False)
        out = sw.run(tx, ty, tth, tv, ta, mode=args.mode, out=out)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out.cost)

    for _ in range(args.warmup):
        step()
    sw.ctx.call("fo_sweep_check", torch.cuda.current_stream().cuda_stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sw.ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = sw.ctx.timing_read()
    sw.ctx.timing(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        pairs = world * M * A * args.steps
        kern_s = kern_ms / 1e3 / max(kern_n, 1)
        abytes = algorithmic_bytes(M, A, T, args.mode)
        achieved = abytes / kern_s / 1e9
        launch = sw.ctx.last_launch()
        res = {
            "metric": "trajectory_x_agent_metric_evals_per_sec", "value": pairs / elapsed, "unit": "pair-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: synthetic 10k trajectories x 256 phantom predictions, T=31 "
                                   "(metric sweep; per-rank shard when n_gpus>1)",
                       "M_per_gpu": M, "A": A, "T": T, "output_mode": args.mode, "traj_order": args.order,
                       "metrics": ["hr", "ttc", "ttce", "dce", "wttc", "cp"], "parallelism": f"traj-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "fo_sweep_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": kern_s * 1e3, "launches_timed": kern_n,
                         "grid": launch["grid"], "block": launch["block"],
                         "kernel_pair_evals_per_sec": M * A / kern_s},
        }
        if out.pair_f is not None:
            from frenetix_occlusion import _native as N
            res["config"]["gate_pair_frac"] = float((out.pair_f[N.PF["max_collision_probability"]] > 0).double().mean())
            res["config"]["collision_pair_frac"] = float((out.pair_f[N.PF["dce"]] == 0).double().mean())
            res["config"]["safe_traj_frac"] = float(out.safe.double().mean())
        if world == 1 and not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            res["cpu_baseline"] = cpu_baseline(S, traj, agents, min(M, 50 * threads), threads)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
