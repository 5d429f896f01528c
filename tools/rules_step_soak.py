#!/usr/bin/env python3
"""Soak run of the one-call planning step WITH the rule stage (fo_step_run, spawn_mode rules / both) against the stage calls on a
second context: n random ego poses (position along and beside the path, heading) at random time steps of scenarios 1-3, every
output compared bit for bit.  usage (GPU box): python tools/rules_step_soak.py [n] [seed]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from frenetix_occlusion import scenario as S  # noqa: E402
import test_rules_step_gpu as TR  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tot = 0
    for k in (1, 2, 3):
        sc = S.load_geometry_npz(os.path.join(ROOT, "tests", "golden", f"scenario{k}_geometry.npz"))
        ego0 = sc.ego_initial
        yaw0 = float(ego0[2])
        path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw0), math.sin(yaw0)]])
        for mode in ("rules", "both"):
            stacks = [TR._stack(torch, sc.lanelets, sc.obstacles, path, sc.intersections, 0, mode=mode, ego=ego0[:2], yaw=yaw0, M=300)
                      for _ in range(2)]
            ps = stacks[1].step()
            for it in range(max(n // 6, 1)):
                step = int(rng.integers(0, 70))
                ego = ego0[:2] + rng.uniform(0.0, 50.0) * np.array([math.cos(yaw0), math.sin(yaw0)]) + rng.normal(0, 0.5, 2)
                yaw, v = yaw0 + rng.normal(0, 0.2), float(rng.uniform(0.0, 14.0))
                res = []
                for q, kk in enumerate(stacks):
                    kk.obs.update(step)
                    kk.sm.upload_obstacles(kk.obs)
                    if q == 1:
                        out = ps.run(ego, yaw, v)
                    else:
                        kk.sm.launch(ego, yaw)
                        kk.sl.find_spawn_points(ego, yaw, None, v, lazy=True)
                        kk.sw.set_agents(*kk.sl.batch.sweep_args(), check=False)
                        out = kk.sw.run(*kk.tr, mode="pair")
                    torch.cuda.synchronize()
                    b = kk.sl.batch
                    res.append([t.cpu().numpy().copy() for t in (out.cost, out.safe, out.pair_f, out.pair_i, kk.sm.cell_class, b.pos, b.yaw,
                                                                  b.v, b.len, b.type, b.head)])
                names = ("cost", "safe", "pair_f", "pair_i", "cell_class", "pos", "yaw", "v", "len", "type", "head")
                for nm, x, y in zip(names, *res):
                    if not np.array_equal(x, y, equal_nan=True):
                        raise AssertionError(f"scenario {k} mode {mode} case {it} step {step} ego {ego} yaw {yaw} v {v}: {nm} differs")
                tot += int((res[0][8] > 0).sum())
            print("scenario", k, mode, "ok", flush=True)
    print(f"{6 * max(n // 6, 1)} random cases: the one-call rules step and the stage calls agree bit for bit ({tot} active prediction slots)")


if __name__ == "__main__":
    main()
