"""Trajectory-batch sharding over the GPUs of one node.

The candidate trajectories of a planning step are independent units: each rank evaluates a contiguous block of them
against the full (replicated) phantom-agent set, and ONE all-gather of the per-trajectory cost vectors
(``[M/R, 16]`` float64, RCCL over xGMI when the tensors live on GPUs) gives every rank the complete cost matrix.  There
is no other collective on the path; the final selection (argmin over the safe trajectories) is done redundantly on
every rank.  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import _native as N


def shard_bounds(M: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous block partition with equal padded size ceil(M / world): rows [lo, hi) belong to `rank`"""
    per = -(-M // world) if M > 0 else 0
    lo = min(rank * per, M)
    return lo, min(lo + per, M)


def shard_arrays(arrays: Dict[str, "torch.Tensor"], world: int, rank: int):
    M = len(next(iter(arrays.values())))
    lo, hi = shard_bounds(M, world, rank)
    return {k: v[lo:hi] for k, v in arrays.items()}, (lo, hi), M


class ShardedAssessment:
    """``run(trajectories)``: shard -> ``compute(shard) -> cost [n,16]`` -> all-gather -> cost [M,16] on every rank."""

    def __init__(self, compute: Callable[[Dict], Optional[torch.Tensor]], group=None):
        self.compute = compute
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def run(self, arrays: Dict) -> torch.Tensor:
        shard, (lo, hi), M = shard_arrays(arrays, self.world, self.rank)
        per = -(-M // self.world) if M > 0 else 0
        cost = self.compute(shard) if hi > lo else None
        ref = cost if cost is not None else None
        device = ref.device if ref is not None else torch.device("cpu")
        if dist.is_initialized() and dist.get_backend(self.group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        mine = torch.full((per, N.NC), float("nan"), dtype=torch.float64, device=device)
        if cost is not None:
            mine[: hi - lo] = cost
        if self.world == 1:
            return mine[:M]
        out = torch.empty((self.world * per, N.NC), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, mine, group=self.group)   # the one collective of the path
        return out[:M]


def select_trajectory(cost: torch.Tensor, key: str = "max_obst_risk_all") -> int:
    """index of the trajectory the planner would take: the safe one with the smallest `key` (ties: lowest index);
    if none is safe, the one with the smallest `max_obst_harm_with_cp_all` (emergency fallback).  -1 for M = 0."""
    if cost.shape[0] == 0:
        return -1
    safe = cost[:, N.COST["safe"]] > 0.5
    if bool(safe.any()):
        v = torch.where(safe, cost[:, N.COST[key]], torch.full_like(cost[:, 0], float("inf")))
    else:
        v = cost[:, N.COST["max_obst_harm_with_cp_all"]]
    return int(torch.argmin(v).item())
