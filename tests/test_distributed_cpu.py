"""The N > 1 path on CPU: two and eight processes, gloo, each evaluates its trajectory shard (with the oracle standing in for the
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


@pytest.mark.parametrize("M,A,world", [(101, 5, 2), (64, 3, 2), (1, 2, 2),
                                       # BASELINE configs[3]'s rank count: an even split (8 x 10), the uneven one (the last
                                       # block padded with NaN, ceil(81 / 8) = 11 rows per rank, rank 7 holds 4) and more
                                       # ranks than rows (ranks 5..7 contribute padding only)
                                       (80, 3, 8), (81, 3, 8), (5, 2, 8)])
def test_shard_and_all_gather_matches_single_process(oracle, M, A, world):
    import torch.multiprocessing as mp
    from frenetix_occlusion import distributed as D
    from frenetix_occlusion import synthetic as S
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, A, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    traj, agents = S.make_batch(M, A, config_id=21)
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.3, "risk": 0.2}, want_lists=False)["cost"]
    bounds = [b for _, b in sorted((g[0], g[3]) for g in got)]                            # by rank
    assert bounds[0][0] == 0 and bounds[-1][1] == M
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))                # contiguous cover, in rank order
    per = -(-M // world)
    assert bounds == [(min(r * per, M), min(r * per + per, M)) for r in range(world)]
    for rank, cost, pick, _ in got:
        assert cost.shape == (M, 16)
        assert np.array_equal(cost, ref, equal_nan=True), f"rank {rank}"
    import torch
    assert {g[2] for g in got} == {D.select_trajectory(torch.from_numpy(ref))}


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
