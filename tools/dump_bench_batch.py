#!/usr/bin/env python3
"""Dump the phantom set of the bench workload (BASELINE configs[2]) so that the branch statistics of the sweep kernel
can be studied off the GPU (tools/sweep_stats.py).  Run on the GPU box; writes gpurun_out/bench_agents.npz."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np
import torch
import yaml
from frenetix_occlusion import _native as N, interface, scenario as SC
from frenetix_occlusion.sensor_model import SensorModel
from frenetix_occlusion.spawn_locator import SpawnLocator

A, T = 256, 31
ctx = N.Context(0)
sc = SC.synthetic_urban_grid()
ego = sc.ego_initial
with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
    cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=0)
sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
sm.launch(ego[:2], float(ego[2]))
b = sl.sample(ego[:2], float(ego[2]), float(ego[3]))
torch.cuda.synchronize()
out = {k: getattr(b, k).cpu().numpy() for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")}
out["ego"] = ego
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "bench_agents.npz"), **out)
print("types", np.bincount(out["type"], minlength=11), "len", np.bincount(out["len"]))
