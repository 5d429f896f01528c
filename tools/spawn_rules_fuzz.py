"""One-off robustness run on the GPU box: the spawn rule families on the device (fo_scene_spawn_rules) against their NumPy
checker (oracle/fo_spawn_rules_ref.py) on random ego poses, reference paths (straight ahead / along the lanelet's centre
line and its successors) and time steps of the three scenario fixtures.  usage: python tools/spawn_rules_fuzz.py [n] [seed] [densify] [cell size]
densify = k > 1: every lanelet bound is subdivided into k pieces per segment (polygons of k times the vertices: at 4 the dynamic
rule's candidate polygons hold hundreds of edges, its per-band edge lists run over many chunks; at 12 they no longer fit its
1 024-vertex table in LDS and the rule reads them from HBM)."""
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
import test_spawn_rules_gpu as TG  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    scs = [S.load_geometry_npz(os.path.join(ROOT, "tests", "golden", f"scenario{i}_geometry.npz")) for i in (1, 2, 3)]
    dens = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    cell = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5    # (>= 0.32: the turn rule samples its 40 m window every cell / 8)
    if dens > 1:
        def sub(b):
            t = np.arange(dens)[None, :, None] / dens
            return np.concatenate(((b[:-1, None, :] + t * (b[1:, None, :] - b[:-1, None, :])).reshape(-1, 2), b[-1:]))
        for sc in scs:
            for l in sc.lanelets:
                l.left, l.right = sub(l.left), sub(l.right)
    kinds, n_pts = {}, 0
    for it in range(n):
        si = int(rng.integers(3))
        sc = scs[si]
        by = {l.lanelet_id: l for l in sc.lanelets}
        ll = sc.lanelets[int(rng.integers(len(sc.lanelets)))]
        c = ll.center
        i = int(rng.integers(0, max(len(c) - 2, 1)))
        ego = c[i] + rng.normal(0.0, 0.3, 2)
        yaw = math.atan2(c[i + 1, 1] - c[i, 1], c[i + 1, 0] - c[i, 0]) + float(rng.normal(0.0, 0.05))
        if rng.random() < 0.5:      # straight ahead
            path = ego[None] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
        else:                        # along the centre lines: this lanelet, then successors (a turn where the map has one)
            parts, cur = [c[max(i - 3, 0):]], ll
            for _ in range(3):
                if not cur.successors:
                    break
                cur = by.get(cur.successors[int(rng.integers(len(cur.successors)))])
                if cur is None:
                    break
                parts.append(cur.center[1:])
            path = np.concatenate(parts)
            keep = np.concatenate(([True], np.hypot(np.diff(path[:, 0]), np.diff(path[:, 1])) > 1e-6))
            path = path[keep]
            if len(path) < 4:
                continue
        step = int(rng.integers(0, 80))
        v = float(rng.uniform(2.0, 12.0))
        try:
            dev, ref, view = TG._both(torch, sc.lanelets, sc.obstacles, path, ego, yaw, v, intersections=sc.intersections,
                                      timestep=step, n_rays=360, cell_size=cell)
        except ValueError:          # ego outside the path's projection domain: nothing to compare
            continue
        try:
            TG._same(dev, ref, view)
        except AssertionError:
            print("MISMATCH at case", it, "scenario", si + 1, "step", step, "ego", ego.tolist(), "yaw", yaw, "v", v, flush=True)
            print("  device :", [(p.agent_type, p.source, np.round(p.position, 6).tolist()) for p in dev])
            print("  checker:", [(p.agent_type, p.source, np.round(p.position, 6).tolist()) for p in ref], flush=True)
            raise
        n_pts += len(ref)
        for p in ref:
            kinds[p.source.split(" ")[0] + ":" + p.agent_type] = kinds.get(p.source.split(" ")[0] + ":" + p.agent_type, 0) + 1
        print(it, "scenario", si + 1, "step", step, "points", [(p.agent_type, p.source) for p in ref], flush=True)
    print("all", n, "cases: device == checker;", n_pts, "spawn points:", kinds)


if __name__ == "__main__":
    main()
