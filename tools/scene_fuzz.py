"""One-off robustness run on the GPU box: the scene stage against the oracle (bit-exact ranges, hit ids, cell classes,
occluded indices, visibility flags, spawn cells) at many random poses / fans / radii on the three scenario maps and
the city grid.  Reuses the checker of tests/test_scene_gpu.py.  usage: python tools/scene_fuzz.py [n] [seed] [near|far] [k]
k > 1: the lanelet bounds of the three scenario maps subdivided k-fold (k times the boundary pieces: at 12 the maps no longer fit
the one-wave shape of the ray / settle kernels).
near: one to three extra obstacles of random size and heading 0.2-4 m from the ego (so close that the end of their occlusion
polygons falls inside the sensor range) and a random shadow length -- the far-chord half-planes (16 arccos per obstacle, device
libm against the host's) under the bit-exact comparison of the cell classes."""
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
from oracle import fo_oracle as oracle  # noqa: E402
import test_scene_gpu as TG  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    near = len(sys.argv) > 3 and sys.argv[3] == "near"
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    maps = [S.load_geometry_npz(os.path.join(ROOT, "tests", "golden", f"scenario{k}_geometry.npz")) for k in (1, 2, 3)]
    dens = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    if dens > 1:
        t = np.arange(dens)[None, :, None] / dens
        sub = lambda b: np.concatenate(((b[:-1, None, :] + t * (b[1:, None, :] - b[:-1, None, :])).reshape(-1, 2), b[-1:]))
        for sc in maps:
            for l in sc.lanelets:
                l.left, l.right = sub(l.left), sub(l.right)
    maps.append(S.synthetic_urban_grid())
    tot = dict(n_exact=0, skipped=0, n_occ=0)
    for it in range(n):
        k = int(rng.integers(len(maps))) if it % 6 else 3
        sc = maps[k]
        c = sc.lanelets[int(rng.integers(len(sc.lanelets)))].center
        i = int(rng.integers(len(c) - 1))
        pos = c[i] + rng.uniform(0, 1) * (c[i + 1] - c[i]) + rng.normal(0, 0.4, 2)
        yaw = math.atan2(*(c[i + 1] - c[i])[::-1]) + rng.normal(0, 0.2)
        fov = float(rng.choice([360.0, 360.0, 120.0, 220.0, 90.0]))
        r = float(rng.choice([50.0, 30.0, 42.5]))
        n_rays = int(rng.choice([720, 720, 361, 97, 180]))
        ts = int(rng.integers(0, 80))
        ego = np.array([pos[0], pos[1], yaw, 8.0])
        L = 100.0
        if near:
            extra = []
            for q in range(int(rng.integers(1, 4))):
                ln, wd = float(rng.uniform(1.5, 14.0)), float(rng.uniform(0.8, 3.0))
                oy = float(rng.uniform(-math.pi, math.pi))
                # centre such that the nearest point of the rectangle is 0.2-4 m from the ego, along a random bearing
                b = float(rng.uniform(-math.pi, math.pi))
                gap = float(rng.uniform(0.2, 4.0))
                # support of the rectangle along -bearing
                hx, hy = math.cos(oy), math.sin(oy)
                sup = abs(0.5 * ln * (hx * math.cos(b) + hy * math.sin(b))) + abs(0.5 * wd * (-hy * math.cos(b) + hx * math.sin(b)))
                cen = pos + (gap + sup) * np.array([math.cos(b), math.sin(b)])
                typ = "bicycle" if rng.integers(8) == 0 else "truck"
                extra.append(S.Obstacle(9000 + q, "static", typ, ln, wd, 0, np.array([cen[0], cen[1], oy, 0.0]), np.zeros((0, 4))))
            sc = S.Scenario(sc.dt, sc.lanelets, extra + list(sc.obstacles), sc.intersections, sc.ego_initial, sc.benchmark_id)
            L = float(rng.choice([100.0, 100.0, 30.0, 10.0, math.inf]))
        st = TG._check_step(torch, oracle, sc, ego, 8.0, ts, sensor_angle=fov, n_rays=n_rays, radius=r,
                            max_agents=int(rng.choice([5, 32, 256])), all_occluded=bool(rng.integers(2)), shadow_length=L)
        for key in tot:
            tot[key] += st[key]
        print(it, "map", k, np.round(pos, 2), round(yaw, 2), fov, r, n_rays, ts, st, flush=True)
    print("all", n, "poses bit-exact;", tot)


if __name__ == "__main__":
    main()
