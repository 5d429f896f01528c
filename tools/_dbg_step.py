import os, sys, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import torch, yaml
from frenetix_occlusion import _native as N
from frenetix_occlusion import interface
from frenetix_occlusion import scenario as SC
from frenetix_occlusion import synthetic as S
from frenetix_occlusion.sensor_model import SensorModel
from frenetix_occlusion.spawn_locator import SpawnLocator
from frenetix_occlusion.step import PlanningStep
from frenetix_occlusion.sweep import MetricSweep
GOLDEN = os.path.join(ROOT, "tests", "golden")
M, A, T = 2000, 32, 31
sc = SC.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
ego0 = sc.ego_initial
with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
    cfg = yaml.safe_load(f)
cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True)
yaw0 = float(ego0[2])
ref = ego0[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw0), math.sin(yaw0)]])
traj = S.make_trajectories(M, T, 0.1, seed=5, ego_pos=ego0[:2], ego_yaw=yaw0)
results = {}
for how in ("stages", "one-call"):
    ctx = N.Context(0)
    sm = SensorModel(sc.lanelets, ref, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx)
    sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
    sl = SpawnLocator(None, ref, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, ctx=ctx)
    tr = [torch.as_tensor(traj[k]).cuda() for k in ("x", "y", "theta", "v", "a")]
    ps = PlanningStep(sm, sl, sw, *tr, mode="pair") if how == "one-call" else None
    got = []
    for i in range(4):
        ego = ego0[:2] + 1.3 * i * np.array([math.cos(yaw0), math.sin(yaw0)])
        yaw, v = yaw0 + 0.02 * i, 5.0 + 2.0 * i
        if ps is not None:
            out = ps.run(ego, yaw, v)
        else:
            sm.launch(ego, yaw)
            sw.set_agents(*sl.sample(ego, yaw, v).sweep_args(), check=False)
            out = sw.run(*tr, mode="pair")
        torch.cuda.synchronize()
        dirs, rmax, half = sm._fan_buffers()
        got.append(dict(cost=out.cost.cpu().numpy().copy(), safe=out.safe.cpu().numpy().copy(), pair_f=out.pair_f.cpu().numpy().copy(),
                    cls=sm.cell_class.cpu().numpy().copy(), pos=sl.batch.pos.cpu().numpy().copy(), n=int(sl.batch.n.item()),
                    dirs=dirs.cpu().numpy().copy(), cell=sl.batch.cell.cpu().numpy().copy(), len=sl.batch.len.cpu().numpy().copy(),
                    yaw=sl.batch.yaw.cpu().numpy().copy()))
    results[how] = got
for i, (a, b) in enumerate(zip(results["stages"], results["one-call"])):
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        same = np.array_equal(x, y, equal_nan=True)
        extra = ""
        if not same and x.dtype.kind == "f":
            d = np.abs(np.nan_to_num(x) - np.nan_to_num(y)); extra = f" max diff {d.max():.3e} n diff {(d > 0).sum()} of {d.size}"
        elif not same:
            extra = f" n diff {(x != y).sum()}"
        print(i, k, "same" if same else "DIFFERENT" + extra)
