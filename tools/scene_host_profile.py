"""Host-side cost of one scene stage (launch overhead of the Python layer), measured without synchronising: the GPU
work is queued, the timers see only what the host spends issuing it.  Run on the GPU box."""
import math
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
from frenetix_occlusion import interface, scenario as SC  # noqa: E402
from frenetix_occlusion.sensor_model import SensorModel  # noqa: E402
from frenetix_occlusion.spawn_locator import SpawnLocator  # noqa: E402


def timeit(fn, n=300):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return dt * 1e6


def main():
    sc = SC.synthetic_urban_grid()
    ego = sc.ego_initial
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    cfg["accelerator"]["spawn"].update(max_agents=256, all_occluded=True, max_dist=45.0)
    yaw = float(ego[2])
    ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5)
    sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
    sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1, horizon=3.0)
    sm.launch(ego[:2], yaw)
    sl.sample(ego[:2], yaw, float(ego[3]))
    print(f"fan()                 {timeit(lambda: sm.fan(yaw)):7.1f} us")
    print(f"enclosed_hole_rings() {timeit(lambda: sm.enclosed_hole_rings(ego[:2], yaw)):7.1f} us")
    print(f"_window_for()         {timeit(lambda: sm._window_for(sm.ego_pos)):7.1f} us")
    print(f"launch()              {timeit(lambda: sm.launch(ego[:2], yaw)):7.1f} us")
    print(f"sample()              {timeit(lambda: sl.sample(ego[:2], yaw, float(ego[3]))):7.1f} us")
    print(f"sample().sweep_args() {timeit(lambda: sl.sample(ego[:2], yaw, float(ego[3])).sweep_args()):7.1f} us")


if __name__ == "__main__":
    main()
