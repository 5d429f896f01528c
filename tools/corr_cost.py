#!/usr/bin/env python3
"""Cost of the correlated-covariance body of the sweep kernel (run on the GPU box): sweep-only workload (10k synthetic
trajectories x 256 synthetic agents, T=31), every n-th agent given a covariance with correlation rho; kernel time from
the context's HIP events.  `python tools/corr_cost.py [mode]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np
import torch
from frenetix_occlusion import _native as N
from frenetix_occlusion import synthetic as S
from frenetix_occlusion.sweep import MetricSweep

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
M, A, T = 10000, 256, 31
dev = torch.device("cuda", 0)
ctx = N.Context(0)
sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1}, device=0, ctx=ctx)
sw.reserve(M, T, A, T)
d = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a)).to(dev, dt)
traj = S.make_trajectories(M, T, 0.1, seed=20240134)
tx, ty, tth, tv, ta = (d(traj[k]) for k in ("x", "y", "theta", "v", "a"))
base = S.make_agents(A, T, 0.1, seed=20240134)
for every, rho in ((0, 0.0), (64, 0.5), (16, 0.5), (4, 0.5), (1, 0.5), (1, 0.95)):
    cov = base["cov"].copy()
    if every:
        s = np.sqrt(cov[::every, :, 0, 0] * cov[::every, :, 1, 1])
        cov[::every, :, 0, 1] = rho * s
        cov[::every, :, 1, 0] = rho * s
    ag = [d(base[k]) for k in ("pos", "yaw", "v")] + [d(cov)] + [d(base[k]) for k in ("shape", "raw_dims")] + \
         [d(base["type"], torch.int32), d(base["len"], torch.int32)]
    out = None
    for _ in range(150):
        sw.set_agents(*ag, check=False)
        out = sw.run(tx, ty, tth, tv, ta, mode=mode, out=out)
    torch.cuda.synchronize()
    ctx.timing(True)
    for _ in range(50):
        sw.set_agents(*ag, check=False)
        out = sw.run(tx, ty, tth, tv, ta, mode=mode, out=out)
    torch.cuda.synchronize()
    ms, n = ctx.timing_read()
    ctx.timing(False)
    ctx.call("fo_sweep_check", torch.cuda.current_stream().cuda_stream)
    print(f"{mode}: correlated agents {0 if not every else A // every:>3} of {A}, rho {rho:4.2f}: kernel {ms / n:.4f} ms")
