"""One rank of a sharded planning step on the GPU (started by tests/test_distributed_gpu.py, one process per rank, before
the child has touched a GPU): the scenario-1 scene, the same candidate batch on every rank, split through the product path --
``FOInterface.trajectory_safety_assessment_batch(..., shard=...)`` and ``PlanningStep(..., shard=...)`` -- one all-gather of the
cost rows.  Rank 0 writes what it gathered, next to the unsharded result it computes itself, into an .npz.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment;  --backend nccl: one GPU per rank (LOCAL_RANK),
    RCCL;  --backend gloo: every rank on GPU 0, the collective over the CPU (what a one-GPU box can run with two ranks)
"""
import argparse
import math
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "frenetix-occlusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--M", type=int, default=333)
    ap.add_argument("--out", required=True)
    ap.add_argument("--config3", action="store_true",
                    help="BASELINE configs[3] itself: the urban grid, 256 phantom slots, --M candidates (10 000), reduced outputs, "
                         "split through PlanningStep(shard=...) -- instead of the scenario-1 scene")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.backend == "nccl" else 0

    import numpy as np
    import torch
    import torch.distributed as dist
    import yaml
    from frenetix_occlusion import distributed as D
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as SC
    from frenetix_occlusion import synthetic as SY
    from frenetix_occlusion.step import PlanningStep

    torch.cuda.set_device(local)
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert dist.get_world_size() == world
    if args.config3:
        return config3(args, rank, world, local)

    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["accelerator"]["device"] = local
    cfg["accelerator"]["spawn"].update(mode="cells", max_agents=12)
    cfg["metrics"]["metric_thresholds"].update(harm=0.1, risk=0.05)
    cfg_path = args.out + f".rank{rank}.yaml"
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
    ego = sc.ego_initial
    ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
    v = SY.VEHICLE_BMW320I
    veh = SimpleNamespace(length=v[0], width=v[1], wb_rear_axle=v[2], mass=v[3], a_max=v[4])
    fo = interface.FOInterface(sc, ref_path, veh, 0.1, config_path=cfg_path)
    traj = SY.make_trajectories(args.M, seed=4711, ego_pos=ego[:2], ego_yaw=float(ego[2]))    # the same batch on every rank

    # (1) the plugin surface: replicated evaluate_scenario, sharded batch assessment
    fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    dev_coll = torch.device("cuda", local) if args.backend == "nccl" else torch.device("cpu")
    cg = D.CostGather(args.M, device=dev_coll)
    assert (cg.world, cg.rank) == (world, rank)
    ptrs = (cg.mine.data_ptr(), cg.gathered.data_ptr())
    for _ in range(3):                                       # repeated steps reuse the two blocks of the collective
        ba = fo.trajectory_safety_assessment_batch(traj, mode="pair", shard=cg)
    assert ptrs == (cg.mine.data_ptr(), cg.gathered.data_ptr()) and cg.calls == 3
    torch.cuda.synchronize()
    assert ba.rows == D.shard_bounds(args.M, world, rank) and len(ba) == args.M
    cost_iface = ba.cost.cpu().numpy().copy()
    safe_iface = ba.safe.cpu().numpy().copy()
    pick = D.select_trajectory(ba.cost)
    local_pair = ba.result.pair_f.cpu().numpy().copy()      # this rank's rows only

    # (2) the one-call planning step with the split inside
    t = lambda k: torch.as_tensor(traj[k]).to(torch.device("cuda", local))
    ps = PlanningStep(fo.sensor_model, fo.spawn_locator, fo.metrics.sweep, t("x"), t("y"), t("theta"), t("v"), t("a"),
                      mode="reduced", shard=D.CostGather(args.M, device=dev_coll))
    for _ in range(2):
        o = ps.run(ego[:2], float(ego[2]), float(ego[3]))
    torch.cuda.synchronize()
    assert o.rows == (cg.lo, cg.hi) and tuple(o.cost.shape) == (cg.hi - cg.lo, 16)
    cost_step = o.cost_all.cpu().numpy().copy()

    # every rank must hold the same gathered matrix and pick the same trajectory
    picks = [None] * world
    dist.all_gather_object(picks, (pick, float(np.nansum(cost_iface)), float(np.nansum(cost_step))))
    assert len(set(picks)) == 1, picks

    if rank == 0:
        # the unsharded answers, on this rank's GPU
        ref = fo.trajectory_safety_assessment_batch(traj, mode="pair")
        torch.cuda.synchronize()
        ps1 = PlanningStep(fo.sensor_model, fo.spawn_locator, fo.metrics.sweep, t("x"), t("y"), t("theta"), t("v"), t("a"),
                           mode="reduced")
        o1 = ps1.run(ego[:2], float(ego[2]), float(ego[3]))
        torch.cuda.synchronize()
        np.savez(args.out, cost_iface=cost_iface, safe_iface=safe_iface, cost_step=cost_step, pick=pick,
                 ref_cost=ref.cost.cpu().numpy(), ref_safe=ref.safe.cpu().numpy(), ref_pick=D.select_trajectory(ref.cost),
                 ref_pair=ref.result.pair_f.cpu().numpy(), local_pair=local_pair, rows=np.array(ba.rows),
                 ref_cost_step=o1.cost.cpu().numpy(), world=world, n_agents=int(fo.spawn_locator.batch.n.item()))
    dist.barrier()
    dist.destroy_process_group()
    os.remove(cfg_path)


def config3(args, rank, world, local):
    """The partition BASELINE configs[3] names -- the synthetic 10k x 256 batch of the bench line, block-partitioned over
    `world` ranks (8: 1 250 trajectories per rank), scene stage and phantoms replicated, ONE all-gather of the cost rows --
    through the product's one-call step.  Rank 0 also runs the unsharded step and writes both."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import yaml
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import distributed as D
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as SC
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.step import PlanningStep
    from frenetix_occlusion.sweep import MetricSweep
    M, A, T = args.M, 256, 31
    thr = {"harm": 0.1, "risk": 1}
    ctx = N.Context(local)
    sc = SC.synthetic_urban_grid()
    ego = sc.ego_initial
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"]["mode"] = "cells"
    cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
    ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
    sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=local)
    sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
    sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=thr, device=local, ctx=ctx)
    traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 3, ego_pos=ego[:2], ego_yaw=float(ego[2]))   # bench.py's batch, on every rank
    dev = torch.device("cuda", local)
    t = lambda k: torch.as_tensor(traj[k]).to(dev)
    dev_coll = dev if args.backend == "nccl" else torch.device("cpu")
    cg = D.CostGather(M, device=dev_coll)
    assert (cg.world, cg.rank) == (world, rank) and (cg.lo, cg.hi) == D.shard_bounds(M, world, rank)
    ps = PlanningStep(sm, sl, sw, t("x"), t("y"), t("theta"), t("v"), t("a"), mode="reduced", shard=cg)
    for _ in range(3):
        o = ps.run(ego[:2], float(ego[2]), float(ego[3]))
    torch.cuda.synchronize()
    per = -(-M // world)
    assert o.rows == (min(rank * per, M), min(rank * per + per, M)) and tuple(o.cost.shape) == (o.rows[1] - o.rows[0], 16)
    assert tuple(o.cost_all.shape) == (M, 16) and cg.calls == 3
    cost_all = o.cost_all.cpu().numpy().copy()
    pick = D.select_trajectory(o.cost_all)
    n_agents = int(sl.batch.n.item())
    # what the padding of the last block looks like before it is cut off: NaN rows behind row M of the gathered blocks
    pad = cg.gathered[M:].cpu().numpy() if cg.collective else np.zeros((0, 16))
    got = [None] * world
    dist.all_gather_object(got, (pick, o.rows, n_agents, float(np.nansum(cost_all))))
    assert len({(g[0], g[2], g[3]) for g in got}) == 1, got          # every rank: the same pick, agent count and matrix
    if rank == 0:
        ps1 = PlanningStep(sm, sl, sw, t("x"), t("y"), t("theta"), t("v"), t("a"), mode="reduced")
        o1 = ps1.run(ego[:2], float(ego[2]), float(ego[3]))
        torch.cuda.synchronize()
        np.savez(args.out, cost_all=cost_all, ref_cost=o1.cost.cpu().numpy(), ref_safe=o1.safe.cpu().numpy(), pick=pick,
                 ref_pick=D.select_trajectory(o1.cost), rows=np.array([g[1] for g in got]), world=world, n_agents=n_agents,
                 pad_is_nan=bool(np.isnan(pad).all()), pad_rows=pad.shape[0], safe_col=N.COST["safe"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
