#!/usr/bin/env python3
"""ms per step of `spawn.mode: rules` (the reference's rule families on the device) on the scenario-1 fixture:
fo_scene_spawn_rules + fo_scene_spawn_rule_agents as SpawnLocator.find_spawn_points issues them, with and without the
read-back of the spawn-point list (the planning step itself never reads it: lazy view)."""
import math
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np
import torch
import yaml
from frenetix_occlusion import interface
from frenetix_occlusion import scenario as S
from frenetix_occlusion.sensor_model import SensorModel
from frenetix_occlusion.spawn_locator import SpawnLocator
from frenetix_occlusion.utils.fo_obstacle import FOObstacles

sc = S.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
    cfg = yaml.safe_load(f)
cfg["accelerator"]["spawn"]["mode"] = "rules"
ego0 = sc.ego_initial
yaw = float(ego0[2])
path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
obs = FOObstacles(sc.obstacles)
sm = SensorModel(sc.lanelets, path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, intersections=sc.intersections)
sl = SpawnLocator(SimpleNamespace(scenario=sc), path, cfg, sm, fo_obstacles=obs)
for step in (0, 8, 25, 60):
    ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
    obs.update(step)
    sm.calc_visible_and_occluded_area(step, ego, yaw, obs)
    pts = sl.find_spawn_points(ego, yaw, None, float(ego0[3]))
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        pts = sl.find_spawn_points(ego, yaw, None, float(ego0[3]))
    dt_rules = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    # device time of the two rule kernels alone (events around the launches, no read-back)
    import frenetix_occlusion.spawn_locator as SLM
    cpu = torch.Tensor.cpu
    e0.record()
    for _ in range(50):
        sl.find_spawn_points(ego, yaw, None, float(ego0[3]), lazy=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"step {step}: {len(pts)} rule points ({[p.agent_type for p in pts]}), find_spawn_points {dt_rules * 1e3:.3f} ms "
          f"(host + device + read-back), device only (rules + agents, lazy list) by events {e0.elapsed_time(e1) / 50:.3f} ms")
    if os.environ.get("FO_RULE_TRACE"):     # tuning builds: tools/build_variant_scene.sh rtrace -DFO_RULE_TRACE=1 (or =2: stamps inside the first fit)
        h = sl.batch.rule_points.cpu().numpy()[-1]
        if h[0] != 0:
            two = os.environ["FO_RULE_TRACE"] == "2"
            if os.environ["FO_RULE_TRACE"] == "3":
                h = h.copy(); h[:7] = h[:7]
            d = np.diff(h[:8] if two else h[:7]) * 0.01
            print("   dynamic-rule phases (us): ", np.round(d, 1).tolist(),
                  "(membership, labelling, sizes, centroid+checks | car fit: clip, sums + rows, hull + rectangle)" if two else
                  "(membership, labelling, sizes, centroid+checks, car fit, bicycle fit)")
            # where the chain sits in the launch: the first wave of any workgroup (RL_WTICK(0)) to the phase stamps
            import ctypes
            from frenetix_occlusion import _native as N
            t = (ctypes.c_longlong * (8 * 1024))()
            if N.load().fo_debug_rule_wticks(t) == 0:
                w = np.array(list(t), dtype=np.int64).reshape(1024, 8)
                ok = w[:, 0] > 0
                k0 = w[ok, 0].min()
                if os.environ["FO_RULE_TRACE"] == "3":      # stamps inside the dynamic rule's set-up (trace build -DFO_RULE_TRACE=3)
                    dyn3 = w[:, 7] > 0
                    for col, name in ((1, "path table in LDS"), (2, "cleared, first barrier"), (3, "lanelets of ego / obstacle asked"), (4, "obstacle projected (wave 0)"),
                                      (5, "intersection found"), (6, "relevance flags set"), (7, "decisions taken")):
                        if dyn3.any():
                            print("      set-up, %-34s %.1f .. %.1f" % (name + ":", (w[dyn3, col].min() - k0) * 0.01, (w[dyn3, col].max() - k0) * 0.01))
                if os.environ["FO_RULE_TRACE"] == "4":      # thread 0 inside the first fit (trace build -DFO_RULE_TRACE=4), from the phase stamp in front of the fits
                    r = int(np.argmax(w[:, 7]))
                    print("      first fit, thread 0's first point (us from the centroid phase's end): point computed %.2f, lattice label + hint read %.2f, distances passed %.2f, "
                          "shadow / class passed %.2f, polygons asked + flag stored %.2f, summed %.2f" % tuple((w[r, 2:8] - h[4]) * 0.01))
                dyn = (w[:, 5] > 0) & (os.environ["FO_RULE_TRACE"] not in ("3", "4"))         # workgroups of the dynamic rule's lattice: tables staged | set-up barrier | nodes done | ticket taken
                if dyn.any():
                    for col, name in ((2, "decisions + offsets"), (3, "polygons staged"), (4, "nodes decided"), (5, "ticket taken")):
                        print("      lattice workgroups, %-20s %.1f .. %.1f" % (name + ":", (w[dyn, col].min() - k0) * 0.01, (w[dyn, col].max() - k0) * 0.01))
                print("   from the launch's first wave (us): last workgroup started %.1f, path tables in LDS (latest) %.1f, phase stamps %s"
                      % ((w[ok, 0].max() - k0) * 0.01, (w[ok, 1].max() - k0) * 0.01, np.round((h[:7] - k0) * 0.01, 1).tolist()))
