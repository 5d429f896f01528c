#!/usr/bin/env python3
"""Benchmark of the Frenetix-Occlusion per-planning-step hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode full|pair|reduced] [--M 10000] [--A 256]

One "step" = one planning step of BASELINE.json configs[2] with every input already resident in HBM:
    visibility ray fan (720 rays @ 0.5 deg, r = 50 m) over the synthetic urban lanelet net (~9.4e3 boundary edges,
    64 parked cars)  ->  cell classes + occluded-cell list (0.5 m cells)  ->  256 phantom agents sampled in the
    occluded cells + their predictions  ->  agent table  ->  trajectory x agent metric sweep (DCE / TTC / TTCE / WTTC /
    CP / harm / risk over T = 31) for 10 000 candidate trajectories  ->  threshold reduction
    (+ one RCCL all-gather of the per-trajectory cost vectors when N > 1).
N > 1: every rank holds its own 10 000-trajectory shard (weak scaling); the scene stage and the agents are replicated
(SURVEY 8e), cost vectors are all-gathered.  `--scene synthetic` replaces the scene stage by a fixed synthetic agent
set (the sweep-only workload of earlier profiles).

Set-up (untimed, before the W warm-up steps): the real step is timed for four agents-per-wave settings of the sweep
kernel and the best one kept -- which also brings the GPU to its sustained clocks, so that the timed K steps do not
depend on W (`--no-autotune` skips it).  `--scene scenario1 --M 2000 --A 32` runs BASELINE configs[1] instead.

Prints ONE JSON line on rank 0.  `value` = trajectory x agent metric evaluations per second over all ranks.
With the default workload on one GPU the line also carries `config.reduced_outputs` (the same step with the cost vectors and
flags only, no per-pair data) and `config.small_batch`: the same planning step at BASELINE
configs[1] (scenario1 geometry, 2 000 candidates x 32 phantom slots), i.e. ms per planning step at the reference's own
problem size (a few hundred steps of ~0.1 ms after the timed region).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(M, A, T, mode):
    """float64 storage.  Inputs read once, outputs written once (DESIGN.md 'Algorithmic bytes')."""
    traj_in = M * T * 8 * 8              # tile table the sweep kernel reads: x, y, cos, sin, theta, v, v cos, v sin
    agent_in = A * (T * 12 * 8 + 8 * 8)  # agent table rows (96 B) + per-agent constants
    out = 0
    if mode in ("pair", "full"):
        out += M * A * (12 * 8 + 4 * 4)
    if mode == "full":
        out += M * A * 5 * (T - 1) * 8
    # per-chunk partial maxima the sweep kernel writes for the reduce kernel are workspace, not counted
    return traj_in + agent_in + out


def usable_cores():
    """cores this process may really use: cgroup quota if there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(S, traj, agents, thr, threads):
    """oracle/fo_oracle.c (a C port of the reference's per-trajectory loops) on the host cores: bounded sample"""
    from oracle import fo_oracle as O  # checker / baseline only
    O.build()
    A = agents["pos"].shape[0]
    probe = {k: v[: 4 * threads] for k, v in traj.items()}
    t0 = time.perf_counter()
    O.sweep(probe, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, nthreads=threads)
    rate = 4 * threads * A / max(time.perf_counter() - t0, 1e-6)           # first guess (includes page faults)
    Ms = int(min(len(traj["x"]), max(8 * threads, rate * 6.0 / A)))         # ~6 s per pass, 3 passes
    sub = {k: v[:Ms] for k, v in traj.items()}
    best, bufs = float("inf"), None
    for _ in range(3):  # first pass page-faults the output buffers; they are reused afterwards
        t0 = time.perf_counter()
        bufs = O.sweep(sub, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, want_lists=True, nthreads=threads, out=bufs)
        best = min(best, time.perf_counter() - t0)
    return {"value": Ms * A / best, "unit": "pair-evals/s", "cores": threads, "kind": "port",
            "sample": f"first {Ms} trajectories x {A} agents of the same batch (same phantom set the GPU step produced), "
                      f"full outputs, oracle/fo_oracle.c, OpenMP over trajectories on {threads} threads, best of 3 "
                      f"passes ({best:.2f} s each)"}


def small_batch_step(local_rank, steps=300):
    """BASELINE configs[1] beside the headline: scenario1 geometry, 2 000 candidates x 32 phantom slots, the same planning
    step (scene stage + sampling + sweep + reduction, reduced outputs as a planner consumes them), own context; a few
    hundred steps of ~0.1 ms.  Reported under config.small_batch -- ms per planning step at the reference's own size."""
    import math
    import numpy as np
    import torch
    import yaml
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as SC
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.sweep import MetricSweep
    M, A, T = 2000, 32, 31
    ctx = N.Context(local_rank)
    sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
    ego = sc.ego_initial
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
    yaw = float(ego[2])
    ref = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    sm = SensorModel(sc.lanelets, ref, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=local_rank)
    sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
    sl = SpawnLocator(None, ref, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, device=local_rank, ctx=ctx)
    sw.reserve(M, T, A, T)
    traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 2, ego_pos=ego[:2], ego_yaw=yaw)
    tr = [torch.as_tensor(traj[k]).to(f"cuda:{local_rank}") for k in ("x", "y", "theta", "v", "a")]
    out = None

    def step():
        nonlocal out
        sm.launch(ego[:2], yaw)
        sw.set_agents(*sl.sample(ego[:2], yaw, float(ego[3])).sweep_args(), check=False)
        out = sw.run(*tr, mode="reduced", out=out)

    for _ in range(100):
        step()
    torch.cuda.synchronize()
    sw.ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kms, kn = sw.ctx.timing_read()
    sw.ctx.timing(False)
    return {"workload": "BASELINE configs[1]: scenario1 geometry, 2000 trajectories x 32 phantom slots, T=31, reduced outputs",
            "ms_per_step": dt * 1e3, "sweep_kernel_ms": kms / max(kn, 1), "A_active": int(sl.batch.n.item()), "steps": steps,
            "sweep_grid": sw.ctx.last_launch()["grid"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="full", choices=["full", "pair", "reduced"])
    ap.add_argument("--M", type=int, default=10000)
    ap.add_argument("--A", type=int, default=256)
    ap.add_argument("--T", type=int, default=31)
    ap.add_argument("--scene", default="urban", choices=["urban", "scenario1", "synthetic"],
                    help="urban: full planning step on the synthetic urban grid; synthetic: sweep only, fixed agents")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --M trajectories per rank; strong: --M trajectories in total, split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true", help="skip the agents-per-wave selection pass of the set-up")
    ap.add_argument("--order", default="sampler", choices=["sampler", "random"],
                    help="row order of the synthetic trajectories (synthetic.make_trajectories)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    use_dist = world > 1 or os.environ.get("FO_BENCH_FORCE_DIST") == "1"   # env: exercise the RCCL path on one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    M, A, T = args.M, args.A, args.T
    if args.scaling == "strong":      # BASELINE configs[3] read literally: ONE batch of --M trajectories split over the ranks
        M = (M + world - 1) // world
    thr = {"harm": 0.1, "risk": 1}   # configurations/simulation/occlusion.yaml:20-28 of the reference's example
    ctx = N.Context(local_rank)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=thr, device=local_rank, ctx=ctx)
    sw.reserve(M, T, A, T)
    d = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a)).to(dev, dt)

    scene = None
    if args.scene in ("urban", "scenario1"):
        import yaml
        from frenetix_occlusion import interface
        from frenetix_occlusion import scenario as SC
        from frenetix_occlusion.sensor_model import SensorModel
        from frenetix_occlusion.spawn_locator import SpawnLocator
        if args.scene == "urban":
            sc = SC.synthetic_urban_grid()
        else:   # BASELINE configs[1]: scenario1.xml geometry (committed fixture), use with --M 2000 --A 32
            sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
        ego = sc.ego_initial
        with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
            cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
        ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
        sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5,
                         ctx=ctx, device=local_rank)
        sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
        sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
        scene = dict(sm=sm, sl=sl, ego=ego, edges=len(sm.map_geometry.edges), obstacles=len(sc.obstacles))
        traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 3 + 1000 * rank, ego_pos=ego[:2], ego_yaw=float(ego[2]),
                                   order=args.order)
        agents = None
    else:
        traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 3 + 1000 * rank, order=args.order)
        agents = S.make_agents(A, T, 0.1, seed=20240131 + 3)
        ag = [d(agents[k]) for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims")] + \
             [d(agents["type"], torch.int32), d(agents["len"], torch.int32)]
    tx, ty, tth, tv, ta = (d(traj[k]) for k in ("x", "y", "theta", "v", "a"))
    out = None
    gathered = torch.empty((world * M, N.NC), dtype=torch.float64, device=dev) if use_dist else None

    def scene_stage():
        sm, sl, ego = scene["sm"], scene["sl"], scene["ego"]
        sm.launch(ego[:2], float(ego[2]))
        return sl.sample(ego[:2], float(ego[2]), float(ego[3])).sweep_args()

    def step():
        nonlocal out
        a_args = scene_stage() if scene is not None else ag
        sw.set_agents(*a_args, check=False)
        out = sw.run(tx, ty, tth, tv, ta, mode=args.mode, out=out)
        if use_dist:
            dist.all_gather_into_tensor(gathered, out.cost)

    # Everything slow on the host side happens first (the first timing() call creates the event pool; the check and the
    # agent count synchronise), so that from here to the timed region the GPU is never idle for longer than a
    # synchronize takes (microseconds): after an idle of milliseconds the MI355X runs two sweeps fast and the next ~15
    # 6-8 % slow, decaying over ~40 steps (power management; per-launch series in DESIGN §7) -- a 20-step window right
    # behind such an idle would consist of that dip.
    sw.ctx.timing(True)
    step()
    sw.ctx.call("fo_sweep_check", torch.cuda.current_stream().cuda_stream)
    n_active = A
    if scene is not None:
        n_active = int(scene["sl"].batch.n.item())
    sw.ctx.timing(False)
    # Set-up, before the W warm-up steps: pick the sweep kernel's agents-per-wave for this batch shape by timing the
    # real step (4 candidates x (20 + 100) steps, ~0.4 s).  The same pass brings the GPU to its sustained clocks: a
    # cold MI355X runs the first few dozen steps 10-15 % slower, so without it the result would depend on W.
    tune = {}
    if not os.environ.get("FO_SWEEP_APW") and not args.no_autotune:
        for apw in (1, 2, 4, 8):
            os.environ["FO_SWEEP_APW"] = str(apw)
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            tune[apw] = (time.perf_counter() - t_a) / 100
        os.environ["FO_SWEEP_APW"] = str(min(tune, key=tune.get))
    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events around the sweep kernel of every timed step (FO_BENCH_TIME_EVERY=k: every k-th; measured: no gain)
    sw.ctx.timing(True, every=int(os.environ.get("FO_BENCH_TIME_EVERY", "1")))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_each = sw.ctx.timing_read_each()
    kern_ms, kern_n = float(sum(kern_each)), len(kern_each)
    if os.environ.get("FO_BENCH_DUMP_SERIES"):
        with open(os.environ["FO_BENCH_DUMP_SERIES"], "w") as f:
            f.write(" ".join(f"{v:.4f}" for v in kern_each) + "\n")
    sw.ctx.timing(False)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # stage breakdown (outside the timed region): the scene stage alone, HIP events on the launch stream
    scene_ms = None
    if scene is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            scene_stage()
        e1.record()
        torch.cuda.synchronize()
        scene_ms = e0.elapsed_time(e1) / 20

    if rank == 0:
        pairs = world * M * n_active * args.steps
        kern_s = kern_ms / 1e3 / max(kern_n, 1)
        abytes = algorithmic_bytes(M, n_active, T, args.mode)
        achieved = abytes / kern_s / 1e9
        launch = sw.ctx.last_launch()
        # HBM traffic of the dominant kernel from the PMC passes of the same command (profiles/, see README there):
        # WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction), KB -> bytes; null when no summary has been committed
        traffic = None
        default_workload = (args.scene == "urban" and M == 10000 and A == 256 and T == 31 and args.mode == "full")
        try:
            if not default_workload:
                raise LookupError("the committed PMC summary belongs to the default workload")
            import csv
            tag = os.environ.get("FO_PROFILE_TAG", "r02_final")
            with open(os.path.join(ROOT, "profiles", f"{tag}_summary.csv")) as f:
                for row in csv.DictReader(f):
                    # the full-output instantiation (PAIR, LISTS = true, true); the small-batch step of the same
                    # command runs another one
                    if row["kernel"].startswith("fo_sweep_queue_kernel<true, true") and row.get("WRITE_SIZE") \
                            and row.get("FETCH_SIZE") and traffic is None:
                        traffic = (float(row["WRITE_SIZE"]) + 2.0 * float(row["FETCH_SIZE"])) * 1024.0
        except Exception:
            traffic = None
        res = {
            "metric": "trajectory_x_agent_metric_evals_per_sec", "value": pairs / elapsed, "unit": "pair-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: synthetic urban lanelet net, 10k trajectories x 256 phantoms, "
                                    "360 deg ray-cast @ 0.5 deg, T=31 (full planning step; per-rank trajectory shard when "
                                    "n_gpus>1)") if args.scene == "urban" else
                                   (f"BASELINE configs[1]: scenario1 geometry, {M} trajectories x {A} phantom slots, full "
                                    "metric set, T=31 (full planning step)") if args.scene == "scenario1" else
                                   "sweep only: 10k synthetic trajectories x 256 synthetic phantom predictions, T=31",
                       "M_per_gpu": M, "A": A, "A_active": n_active, "T": T, "output_mode": args.mode,
                       "traj_order": args.order, "metrics": ["hr", "ttc", "ttce", "dce", "wttc", "cp"],
                       "scene_stage_ms": scene_ms, "sweep_kernel_ms": kern_s * 1e3,
                       "agents_per_wave": int(os.environ["FO_SWEEP_APW"]) if os.environ.get("FO_SWEEP_APW") else None,
                       "setup_autotune_ms_per_step": {str(k): round(v * 1e3, 4) for k, v in tune.items()},
                       "boundary_edges": scene["edges"] if scene else None, "rays": 720 if scene else None,
                       "parallelism": f"traj-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": "fo_sweep_queue_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": kern_s * 1e3, "launches_timed": kern_n,
                         "kernel_ms_p50": float(np.percentile(kern_each, 50)) if kern_each else None,
                         "kernel_ms_p95": float(np.percentile(kern_each, 95)) if kern_each else None,
                         "grid": launch["grid"], "block": launch["block"],
                         "kernel_pair_evals_per_sec": M * n_active / kern_s},
        }
        if out.pair_f is not None:
            res["config"]["gate_pair_frac"] = float((out.pair_f[N.PF["max_collision_probability"]] > 0).double().mean())
            res["config"]["collision_pair_frac"] = float((out.pair_f[N.PF["dce"]] == 0).double().mean())
            res["config"]["safe_traj_frac"] = float(out.safe.double().mean())
        if world == 1 and default_workload:
            # the same step with reduced outputs (cost vectors + flags: what a planner loop consumes), beside the headline
            red = None
            for _ in range(60):
                a_args = scene_stage()
                sw.set_agents(*a_args, check=False)
                red = sw.run(tx, ty, tth, tv, ta, mode="reduced", out=red)
            torch.cuda.synchronize()
            sw.ctx.timing(True)
            t_r = time.perf_counter()
            for _ in range(100):
                a_args = scene_stage()
                sw.set_agents(*a_args, check=False)
                red = sw.run(tx, ty, tth, tv, ta, mode="reduced", out=red)
            torch.cuda.synchronize()
            dt_r = (time.perf_counter() - t_r) / 100
            kms_r, kn_r = sw.ctx.timing_read()
            sw.ctx.timing(False)
            res["config"]["reduced_outputs"] = {"ms_per_step": dt_r * 1e3, "pair_evals_per_sec": M * n_active / dt_r,
                                                "sweep_kernel_ms": kms_r / max(kn_r, 1), "steps": 100}
            res["config"]["small_batch"] = small_batch_step(local_rank)
        if world == 1 and not args.no_cpu_baseline:
            if scene is not None:
                b = scene["sl"].batch
                agents = {k: getattr(b, k).cpu().numpy() for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")}
            res["cpu_baseline"] = cpu_baseline(S, traj, agents, thr, usable_cores())
        else:
            res["cpu_baseline"] = None
        if use_dist:   # RCCL's start-up banner sits in the C library's stdout buffer: let it out first, the JSON line last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
