#!/usr/bin/env python3
"""Soak run of the one-call planning step (fo_step_run: fan, candidate flags and agent table fused into their neighbour
kernels) against the five stage calls on a second context: n random ego poses along scenario 1, every output compared
bit for bit.  usage (GPU box): python tools/step_soak.py [n] [seed]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import torch  # noqa: E402
import yaml  # noqa: E402
from frenetix_occlusion import _native as N, interface, scenario as SC, synthetic as S  # noqa: E402
from frenetix_occlusion.sensor_model import SensorModel  # noqa: E402
from frenetix_occlusion.spawn_locator import SpawnLocator  # noqa: E402
from frenetix_occlusion.step import PlanningStep  # noqa: E402
from frenetix_occlusion.sweep import MetricSweep  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    M, A, T = 700, 32, 31
    sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    cfg["accelerator"]["spawn"].update(max_agents=A // 2, all_occluded=True, routes=2, pattern=["Car", "Bicycle", "Pedestrian", "Car"])
    yaw0 = float(ego0[2])
    ref = ego0[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw0), math.sin(yaw0)]])
    traj = S.make_trajectories(M, T, 0.1, seed=5, ego_pos=ego0[:2], ego_yaw=yaw0)
    sides = []
    for how in ("stages", "one-call"):
        ctx = N.Context(0)
        sm = SensorModel(sc.lanelets, ref, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, routes=2)
        sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
        sl = SpawnLocator(None, ref, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
        sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, ctx=ctx)
        tr = [torch.as_tensor(traj[k]).cuda() for k in ("x", "y", "theta", "v", "a")]
        ps = PlanningStep(sm, sl, sw, *tr, mode="pair") if how == "one-call" else None
        sides.append((sm, sl, sw, tr, ps))
    n_ph = 0
    for i in range(n):
        ego = ego0[:2] + rng.uniform(0.0, 45.0) * np.array([math.cos(yaw0), math.sin(yaw0)]) + rng.normal(0, 0.4, 2)
        yaw, v = yaw0 + rng.normal(0, 0.15), rng.uniform(0.0, 15.0)
        got = []
        for sm, sl, sw, tr, ps in sides:
            if ps is not None:
                out = ps.run(ego, yaw, v)
            else:
                sm.launch(ego, yaw)
                sw.set_agents(*sl.sample(ego, yaw, v).sweep_args(), check=False)
                out = sw.run(*tr, mode="pair")
            torch.cuda.synchronize()
            got.append((out.cost.cpu().numpy(), out.safe.cpu().numpy(), out.pair_f.cpu().numpy(), out.pair_i.cpu().numpy(),
                        sm.cell_class.cpu().numpy(), sl.batch.pos.cpu().numpy(), sl.batch.len.cpu().numpy(), int(sl.batch.n.item())))
        a, b = got
        assert a[7] == b[7], (i, a[7], b[7])
        names = ("cost", "safe", "pair_f", "pair_i", "cell_class", "agent pos", "agent len")
        for nm, x, y in zip(names, a[:7], b[:7]):
            if not np.array_equal(x, y, equal_nan=True):
                d = np.argwhere(~((x == y) | ((x != x) & (y != y))))
                raise AssertionError(f"pose {i} ego {ego} yaw {yaw} v {v}: {nm} differs in {len(d)} of {x.size} elements, first at {d[0]}: "
                                     f"{x[tuple(d[0])]!r} (stages) vs {y[tuple(d[0])]!r} (one call)")
        n_ph += a[7]
    print(f"{n} random poses: the one-call step and the stage calls agree bit for bit ({n_ph / n:.1f} phantoms per step)")


if __name__ == "__main__":
    main()
