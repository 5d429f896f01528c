"""The N > 1 path on CPU: two processes, gloo, each evaluates its trajectory shard (with the oracle standing in for the
HIP sweep), one all-gather of the cost vectors -- the result must be bit-identical to the single-process evaluation."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, A, q):
    for p in (ROOT, os.path.join(ROOT, "frenetix-occlusion_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from frenetix_occlusion import distributed as D
    from frenetix_occlusion import synthetic as S
    from oracle import fo_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    traj, agents = S.make_batch(M, A, config_id=21)
    thr = {"harm": 0.3, "risk": 0.2}

    def compute(shard):
        r = O.sweep(shard, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, want_lists=False)
        return torch.from_numpy(r["cost"])

    cost = D.ShardedAssessment(compute).run(traj)
    pick = D.select_trajectory(cost)
    q.put((rank, cost.numpy(), pick, D.shard_bounds(M, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("M,A", [(101, 5), (64, 3), (1, 2)])
def test_two_rank_shard_and_all_gather_matches_single_process(oracle, M, A):
    import torch.multiprocessing as mp
    from frenetix_occlusion import distributed as D
    from frenetix_occlusion import synthetic as S
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, M, A, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    traj, agents = S.make_batch(M, A, config_id=21)
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.3, "risk": 0.2}, want_lists=False)["cost"]
    bounds = sorted(g[3] for g in got)
    assert bounds[0][0] == 0 and bounds[-1][1] == M and bounds[0][1] == bounds[1][0]      # contiguous cover
    for rank, cost, pick, _ in got:
        assert cost.shape == (M, 16)
        assert np.array_equal(cost, ref, equal_nan=True), f"rank {rank}"
    import torch
    assert got[0][2] == got[1][2] == D.select_trajectory(torch.from_numpy(ref))


def test_shard_bounds_cover_every_row_once():
    from frenetix_occlusion.distributed import shard_bounds
    for M in (0, 1, 7, 8, 9, 10000):
        for world in (1, 2, 4, 8):
            rows = []
            for r in range(world):
                lo, hi = shard_bounds(M, world, r)
                assert 0 <= lo <= hi <= M
                rows += list(range(lo, hi))
            assert rows == list(range(M))


def test_select_trajectory_prefers_safe_then_falls_back():
    import torch
    from frenetix_occlusion import _native as N
    from frenetix_occlusion.distributed import select_trajectory
    c = torch.zeros((4, 16), dtype=torch.float64)
    c[:, N.COST["max_obst_risk_all"]] = torch.tensor([0.5, 0.1, 0.3, 0.05])
    c[:, N.COST["safe"]] = torch.tensor([1.0, 1.0, 1.0, 0.0])
    assert select_trajectory(c) == 1                      # the unsafe one has the lowest risk but is excluded
    c[:, N.COST["safe"]] = 0.0
    c[:, N.COST["max_obst_harm_with_cp_all"]] = torch.tensor([0.9, 0.8, 0.2, 0.7])
    assert select_trajectory(c) == 2
    assert select_trajectory(c[:0]) == -1
