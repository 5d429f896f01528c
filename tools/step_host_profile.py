"""Host-side cost of issuing one bench step (no synchronisation inside the loop) and its parts, plus a cProfile of the
hot functions.  Run on the GPU box: python tools/step_host_profile.py [scenario1|urban] [M] [A]"""
import cProfile
import math
import os
import pstats
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
from frenetix_occlusion import _native as N, interface, scenario as SC, synthetic as S  # noqa: E402
from frenetix_occlusion.sensor_model import SensorModel  # noqa: E402
from frenetix_occlusion.spawn_locator import SpawnLocator  # noqa: E402
from frenetix_occlusion.sweep import MetricSweep  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "scenario1"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
A = int(sys.argv[3]) if len(sys.argv) > 3 else 32
T = 31
ctx = N.Context(0)
sc = SC.synthetic_urban_grid() if scene == "urban" else SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
ego = sc.ego_initial
with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
    cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
yaw = float(ego[2])
ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=0)
sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1, horizon=3.0)
sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, device=0, ctx=ctx)
sw.reserve(M, T, A, T)
traj = S.make_trajectories(M, T, 0.1, seed=1, ego_pos=ego[:2], ego_yaw=yaw)
tx, ty, tth, tv, ta = (torch.as_tensor(traj[k]).cuda() for k in ("x", "y", "theta", "v", "a"))
out = None


def step():
    global out
    sm.launch(ego[:2], yaw)
    a_args = sl.sample(ego[:2], yaw, float(ego[3])).sweep_args()
    sw.set_agents(*a_args, check=False)
    out = sw.run(tx, ty, tth, tv, ta, mode="reduced", out=out)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    total = (time.perf_counter() - t) / n
    return host * 1e6, total * 1e6


print("step: host issue %.1f us, with the GPU drained %.1f us per step" % timeit(step))
print("  sm.launch          %.1f us" % timeit(lambda: sm.launch(ego[:2], yaw))[0])
print("  sl.sample+args     %.1f us" % timeit(lambda: sl.sample(ego[:2], yaw, float(ego[3])).sweep_args())[0])
args = sl.sample(ego[:2], yaw, float(ego[3])).sweep_args()
print("  sw.set_agents      %.1f us" % timeit(lambda: sw.set_agents(*args, check=False))[0])
print("  sw.run             %.1f us" % timeit(lambda: sw.run(tx, ty, tth, tv, ta, mode="reduced", out=out))[0])
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
