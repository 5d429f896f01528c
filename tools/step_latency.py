"""Wall-clock latency of one planning step through the drop-in boundary itself (FOInterface.evaluate_scenario +
trajectory_safety_assessment_batch, synchronised each step) -- what a planner calling the Python interface sees, as
opposed to bench.py's lean loop.  Run on the GPU box: python tools/step_latency.py [cells|rules|both] [M]."""
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
from frenetix_occlusion import interface, scenario as S, synthetic as SY  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "cells"
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    cfg["accelerator"]["spawn"].update(mode=mode, max_agents=32)
    sc = S.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario3_geometry.npz"))
    by = {l.lanelet_id: l for l in sc.lanelets}
    parts = [by[1].center]
    for lid in (12, 9):
        c = by[lid].center
        parts.append(c[1:] if np.linalg.norm(c[0] - parts[-1][-1]) < 1e-2 else c)
    ref_path = np.concatenate(parts)
    ego = np.array([12.0, 0.0, 0.0, 8.0])
    veh = SimpleNamespace(**dict(zip(("length", "width", "wb_rear_axle", "mass", "a_max"), SY.VEHICLE_BMW320I)))
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
    fo = interface.FOInterface(sc, ref_path, veh, 0.1, config_path=f.name)
    traj = SY.make_trajectories(M, seed=3, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    stages = {"evaluate_scenario": [], "assessment_batch": []}
    for it in range(25):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fo.evaluate_scenario({}, ego[:2], float(ego[2]), None, float(ego[3]), it, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ba = fo.trajectory_safety_assessment_batch(traj, mode="reduced")
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 5:
            stages["evaluate_scenario"].append(t1 - t0)
            stages["assessment_batch"].append(t2 - t1)
    # the reference's own calling pattern: one call per candidate (interface.py:216-219), no batch beforehand
    objs = [SimpleNamespace(cartesian=SimpleNamespace(**{k: v[i] for k, v in traj.items()})) for i in range(min(M, 200))]
    fo.metrics.invalidate()
    t0 = time.perf_counter()
    for o in objs:
        res, safe = fo.trajectory_safety_assessment(o)
    per_call = (time.perf_counter() - t0) / len(objs)
    # ... and the same objects served from one full-output batch
    t0 = time.perf_counter()
    fo.trajectory_safety_assessment_batch(objs, mode="full")
    for o in objs:
        res, safe = fo.trajectory_safety_assessment(o)
    served = (time.perf_counter() - t0) / len(objs)
    print(f"  per-trajectory call     {per_call * 1e3:8.3f} ms each (no batch);  {served * 1e3:8.3f} ms each when served "
          f"from one full-output batch of {len(objs)}")
    print(f"mode={mode} M={M} agents={len(fo.agent_manager.phantom_agents)} spawn_points={len(fo.spawn_points)}")
    for k, v in stages.items():
        print(f"  {k:20s} median {np.median(v) * 1e3:8.3f} ms   max {np.max(v) * 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
