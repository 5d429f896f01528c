#!/usr/bin/env python3
"""Does replaying the planning step as a HIP graph shorten it?  (experiment: the bench's small step has identical arguments every
step, so a captured graph replays it exactly)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np, torch, yaml
from frenetix_occlusion import _native as N, interface, scenario as SC, synthetic as S
from frenetix_occlusion.sensor_model import SensorModel
from frenetix_occlusion.spawn_locator import SpawnLocator
from frenetix_occlusion.step import PlanningStep
from frenetix_occlusion.sweep import MetricSweep
from frenetix_occlusion.utils.fo_obstacle import FOObstacles
import bench
M, T = 2000, 31
ctx = N.Context(0)
sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
ego = sc.ego_initial
with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
    cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
cfg["accelerator"]["spawn"].update(max_agents=32, routes=0)
yaw = float(ego[2])
path = ego[None, :2] + np.linspace(0.0, 60.0, 61)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
obs = FOObstacles(sc.obstacles); obs.update(0)
sm = SensorModel(sc.lanelets, path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=0)
sm.upload_obstacles(obs)
sl = SpawnLocator(None, path, cfg, sm, dt=0.1)
sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=bench.THR, device=0, ctx=ctx)
traj = S.make_trajectories(M, T, 0.1, seed=7, ego_pos=ego[:2], ego_yaw=yaw)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    tr = [torch.as_tensor(traj[k]).to("cuda:0") for k in ("x", "y", "theta", "v", "a")]
    ps = PlanningStep(sm, sl, sw, *tr, mode="reduced")
    for _ in range(50):
        ps.run(ego[:2], yaw, float(ego[3]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        ps.run(ego[:2], yaw, float(ego[3]))
    torch.cuda.synchronize()
    print("launches: %.4f ms per step" % ((time.perf_counter() - t0) / 500 * 1e3))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            ps.run(ego[:2], yaw, float(ego[3]))
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            g.replay()
        torch.cuda.synchronize()
        print("graph replay: %.4f ms per step" % ((time.perf_counter() - t0) / 500 * 1e3))
    except Exception as e:
        print("capture failed:", repr(e)[:300])
