import os, sys
sys.path.insert(0, "frenetix-occlusion_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from frenetix_occlusion import synthetic as S
from frenetix_occlusion.sweep import MetricSweep
traj, agents = S.make_batch(300, 16, config_id=1)
thr = {"harm": 0.3, "risk": 0.2, "ttc": 1.0, "dce": 0.05, "cp": 0.8}
def run(lists, split=None):
    if split is None: os.environ.pop("FO_SWEEP_SPLIT", None)
    else: os.environ["FO_SWEEP_SPLIT"] = split
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=thr)
    sw.set_agents(*[agents[k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")])
    o = sw.run(*[traj[k] for k in ("x", "y", "theta", "v", "a")], mode="full", lists=lists)
    torch.cuda.synchronize()
    return o.cost.cpu().numpy(), o.pair_f.cpu().numpy(), sw.ctx.last_launch()
a = run("f64"); b = run("f32x"); c = run("f64", "0"); d = run("f32", "0"); e = run("f32")
for name, (x, y) in {"f64split-vs-f32x": (a, b), "f64split-vs-f64nosplit": (a, c), "f64nosplit-vs-f32x": (c, b), "f32nosplit-vs-f32x": (d, b), "f32split-vs-f64split": (e, a)}.items():
    dc = ~((x[0] == y[0]) | (np.isnan(x[0]) & np.isnan(y[0])))
    dp = ~((x[1] == y[1]) | (np.isnan(x[1]) & np.isnan(y[1])))
    print(name, "launch", x[2], y[2], "cost cols differing", np.unique(np.nonzero(dc)[1]), "n", dc.sum(), "pair rows", np.unique(np.nonzero(dp)[0]), "n", dp.sum(),
          "max abs", np.nanmax(np.abs(np.where(dc, x[0] - y[0], 0))), np.nanmax(np.abs(np.where(dp, x[1] - y[1], 0))))
k = agents["type"]; print("types", k)
dp = ~((a[1] == b[1]) | (np.isnan(a[1]) & np.isnan(b[1])))
print("agents with diffs", np.unique(np.nonzero(dp)[1]))
