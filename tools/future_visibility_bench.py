"""Throughput of the future-visibility extension at the BASELINE configs[2] size (10 000 trajectories, city grid):
HIP-event time of fo_scene_future_visibility for a few (stride, rays) settings.  Run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
from frenetix_occlusion import scenario as SC, synthetic as SY  # noqa: E402
from frenetix_occlusion.sensor_model import SensorModel  # noqa: E402


def main():
    sc = SC.synthetic_urban_grid()
    ego = sc.ego_initial
    sm = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0)
    sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
    sm.launch(ego[:2], float(ego[2]))
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    traj = SY.make_trajectories(M, 31, 0.1, seed=20240134, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    dev = sm.device
    tx, ty = torch.as_tensor(traj["x"]).to(dev), torch.as_tensor(traj["y"]).to(dev)
    n_occ = int(sm.n_occluded.item())
    for stride, rays in ((5, 192), (5, 720), (10, 192), (5, 96), (5, 256), (5, 384), (1, 192)):
        sm.future_visibility(tx, ty, t_stride=stride, n_rays=rays)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rev, area = sm.future_visibility(tx, ty, t_stride=stride, n_rays=rays)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        K = (31 + stride - 1) // stride
        print(f"M={M} stride={stride} (K={K}) rays={rays}: {ms:8.3f} ms  = {M * K * rays / ms / 1e6:6.2f} Grays/s, "
              f"{M * K / ms / 1e3:6.2f} Mposes/s; occluded cells {n_occ}; mean revealed at last pose "
              f"{float(rev[:, -1].double().mean()):.1f}")


if __name__ == "__main__":
    main()
