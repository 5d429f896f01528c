#!/usr/bin/env python3
"""BASELINE configs[4]: four egos x 2 000 candidate trajectories each, one fo_ctx / HIP stream per ego, egos dealt
round-robin to the GPUs of the node (one process per GPU; launch with torch.distributed.run for N > 1, no collective:
the egos are independent planners).  The scenario XMLs hold one planning problem each, so three more ego poses are
placed along lanelet centre lines (scenario2 / scenario3 geometry fixtures, alternating).

    python tools/multi_ego_bench.py [--egos 4] [--M 2000] [--A 32] [--steps 200] [--mode reduced]

One "step" = one planning step of EVERY ego of this rank (scene stage + phantom sampling + sweep + reduction), all
queued on the egos' own streams before one synchronisation.  Prints one JSON line per rank.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402
from frenetix_occlusion import _native as N, interface, scenario as SC, synthetic as S  # noqa: E402
from frenetix_occlusion.sensor_model import SensorModel  # noqa: E402
from frenetix_occlusion.spawn_locator import SpawnLocator  # noqa: E402
from frenetix_occlusion.sweep import MetricSweep  # noqa: E402


def candidate_poses():
    """the scenario's own planning problem first, then poses along the lanelet centre lines (three per lanelet)"""
    sc0 = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario2_geometry.npz"))
    poses = [sc0.ego_initial.copy()]
    for ll in sc0.lanelets:
        c = ll.center
        if len(c) >= 4:
            for frac in (0.25, 0.5, 0.75):
                i = min(int(frac * (len(c) - 1)), len(c) - 2)
                poses.append(np.array([c[i, 0], c[i, 1], math.atan2(c[i + 1, 1] - c[i, 1], c[i + 1, 0] - c[i, 0]), 6.0]))
    return poses


def ego_poses(n, dev, A, T):
    """n poses whose surroundings hide something: a candidate is kept if its first planning step yields phantoms (an
    ego with an empty occluded area would make its share of the batch a no-op)"""
    keep = []
    for p in candidate_poses():
        if len(keep) == n:
            break
        e = Ego(len(keep), p, dev, 64, A, T)
        e.step("reduced")
        e.stream.synchronize()
        if int(e.sl.batch.n.item()) > 0 or not keep:
            keep.append(p)
    return keep


class Ego:
    def __init__(self, idx, pose, dev, M, A, T, share_with=None):
        self.pose = pose
        self.stream = torch.cuda.Stream(device=dev)
        ctx = N.Context(dev)
        sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", f"scenario{2 + idx % 2}_geometry.npz"))
        with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
            cfg = yaml.safe_load(f)
            cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
        cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=40.0)
        yaw = float(pose[2])
        ref = pose[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
        # egos on the same scenario and GPU read one copy of the static map (fo_scene_share_map)
        self.sm = SensorModel(sc.lanelets, ref, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx,
                              device=dev, share_map_with=share_with.sm if share_with is not None else None)
        self.sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
        self.sl = SpawnLocator(None, ref, cfg, self.sm, dt=0.1, horizon=(T - 1) * 0.1)
        self.sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, device=dev, ctx=ctx)
        self.sw.reserve(M, T, A, T)
        traj = S.make_trajectories(M, T, 0.1, seed=11 + idx, ego_pos=pose[:2], ego_yaw=yaw)
        self.traj = [torch.as_tensor(traj[k]).to(f"cuda:{dev}") for k in ("x", "y", "theta", "v", "a")]
        self.out = None
        self.ps = None

    def step(self, mode, one_call=True):
        with torch.cuda.stream(self.stream):
            if one_call:      # the whole step through fo_step_run: one FFI crossing per ego and step
                if self.ps is None:
                    from frenetix_occlusion.step import PlanningStep
                    self.ps = PlanningStep(self.sm, self.sl, self.sw, *self.traj, mode=mode)
                self.out = self.ps.run(self.pose[:2], float(self.pose[2]), float(self.pose[3]))
                return
            self.sm.launch(self.pose[:2], float(self.pose[2]))
            args = self.sl.sample(self.pose[:2], float(self.pose[2]), float(self.pose[3])).sweep_args()
            self.sw.set_agents(*args, check=False)
            self.out = self.sw.run(*self.traj, mode=mode, out=self.out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--egos", type=int, default=4)
    ap.add_argument("--M", type=int, default=2000)
    ap.add_argument("--A", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--mode", default="reduced", choices=["reduced", "pair", "full"])
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    T = 31
    mine = [(i, p) for i, p in enumerate(ego_poses(args.egos, dev, args.A, T)) if i % world == rank]
    egos = []
    for i, p in mine:
        same = next((e for e in egos if e.scenario == 2 + i % 2), None)
        egos.append(Ego(i, p, dev, args.M, args.A, T, share_with=same))
        egos[-1].scenario = 2 + i % 2
    torch.cuda.synchronize()

    def step():
        for e in egos:
            e.step(args.mode)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # the same egos one after the other, each drained before the next starts (what sharing the GPU buys)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for e in egos:
            e.step(args.mode)
            e.stream.synchronize()
    serial = (time.perf_counter() - t0) / args.steps
    # and queued through the five stage calls from Python instead of fo_step_run (round 2's way: host-bound)
    for _ in range(20):
        for e in egos:
            e.step(args.mode, one_call=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for e in egos:
            e.step(args.mode, one_call=False)
    torch.cuda.synchronize()
    stage_calls = (time.perf_counter() - t0) / args.steps
    # one host thread per ego, as a planner per ego would run (ctypes drops the GIL inside fo_step_run, so the native
    # halves of the egos' steps overlap on the host as their kernels do on the GPU)
    import threading
    for e in egos:
        e.step(args.mode)
    torch.cuda.synchronize()
    go = threading.Barrier(len(egos) + 1)

    def worker(e):
        torch.cuda.set_device(dev)
        go.wait()
        for _ in range(args.steps):
            e.step(args.mode)
        e.stream.synchronize()

    th = [threading.Thread(target=worker, args=(e,)) for e in egos]
    for t_ in th:
        t_.start()
    go.wait()
    t0 = time.perf_counter()
    for t_ in th:
        t_.join()
    threaded = (time.perf_counter() - t0) / args.steps
    n_act = [int(e.sl.batch.n.item()) for e in egos]
    print(json.dumps({"workload": "BASELINE configs[4]: multi-ego, one fo_ctx and stream per ego", "rank": rank, "n_gpus": world,
                      "egos_on_this_gpu": len(egos), "M_per_ego": args.M, "phantoms_per_ego": n_act, "mode": args.mode,
                      "ms_per_step_all_egos": dt * 1e3, "ms_per_step_egos_one_after_the_other": serial * 1e3,
                      "ms_per_step_all_egos_stage_calls": stage_calls * 1e3,
                      "ms_per_step_all_egos_one_host_thread_per_ego": threaded * 1e3,
                      "pair_evals_per_sec": sum(args.M * a for a in n_act) / dt}), flush=True)


if __name__ == "__main__":
    main()
