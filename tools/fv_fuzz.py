"""One-off robustness run on the GPU box: the future-visibility extension against its CPU restatement at random ego
poses, trajectory sets, strides, ray counts and radii.  usage: python tools/fv_fuzz.py [n] [seed]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import torch  # noqa: E402
from frenetix_occlusion import scenario as S, synthetic as SY  # noqa: E402
from frenetix_occlusion.sensor_model import SensorModel  # noqa: E402
from frenetix_occlusion.utils.fo_obstacle import FOObstacles  # noqa: E402
from oracle import fo_oracle as oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    maps = [S.load_geometry_npz(os.path.join(ROOT, "tests", "golden", f"scenario{k}_geometry.npz")) for k in (1, 2, 3)]
    maps.append(S.synthetic_urban_grid())
    models = {}
    for it in range(n):
        k = int(rng.integers(len(maps)))
        sc = maps[k]
        r = float(rng.choice([50.0, 30.0, 42.5]))
        if (k, r) not in models:
            models[(k, r)] = SensorModel(sc.lanelets, None, sensor_radius=r, sensor_angle=360.0)
        sm = models[(k, r)]
        c = sc.lanelets[int(rng.integers(len(sc.lanelets)))].center
        i = int(rng.integers(len(c) - 1))
        pos = c[i] + rng.uniform(0, 1) * (c[i + 1] - c[i]) + rng.normal(0, 0.4, 2)
        yaw = math.atan2(*(c[i + 1] - c[i])[::-1]) + rng.normal(0, 0.2)
        ts = int(rng.integers(0, 80))
        ob = FOObstacles(sc.obstacles)
        ob.update(ts)
        sm.calc_visible_and_occluded_area(ts, pos, yaw, ob)
        M = int(rng.choice([1, 3, 17, 64]))
        T = int(rng.choice([2, 11, 31]))
        stride = int(rng.choice([1, 3, 5, 40]))
        n_rays = int(rng.choice([4, 5, 64, 97, 192, 255, 256, 257, 360, 511, 720, 768]))
        traj = SY.make_trajectories(M, T, 0.1, seed=int(rng.integers(1 << 30)), ego_pos=pos, ego_yaw=yaw)
        rev, area = sm.future_visibility(traj["x"], traj["y"], t_stride=stride, n_rays=n_rays)
        torch.cuda.synchronize()
        corn, cen, flags = ob.arrays()
        w = sm.window
        x0, y0 = sm.raster_origin
        ref_rev, ref_area = oracle.future_visibility(traj["x"], traj["y"], stride, sm._fv_dirs.cpu().numpy(), r,
                                                     sm.map_geometry.edges, corn, flags, sm.occluded_cells().cpu().numpy(),
                                                     x0, y0, sm.cell_size, w.ix0, w.iy0, w.nx)
        assert np.array_equal(rev.cpu().numpy(), ref_rev), (it, k, pos, yaw, r, M, T, stride, n_rays)
        np.testing.assert_allclose(area.cpu().numpy(), ref_area, rtol=1e-12, atol=1e-9)
        print(it, "map", k, "r", r, "M", M, "T", T, "stride", stride, "rays", n_rays, "revealed max", int(ref_rev.max()), flush=True)
    print("all", n, "cases: revealed counts bit-exact, areas to 1e-12")


if __name__ == "__main__":
    main()
