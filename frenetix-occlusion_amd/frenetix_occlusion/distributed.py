"""Trajectory-batch sharding over the GPUs of one node (BASELINE configs[3]).

The candidate trajectories of a planning step are independent units: each rank evaluates a contiguous block of them
against the full (replicated) phantom-agent set, and ONE all-gather of the per-trajectory cost vectors
(``[M/R, 16]`` float64, RCCL over xGMI when the tensors live on GPUs) gives every rank the complete cost matrix.  There
is no other collective on the path; the final selection (argmin over the safe trajectories) is done redundantly on
every rank.  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

:class:`CostGather` is that split as an object of the product path: ``Metric.evaluate_batch(..., shard=...)``,
``FOInterface.trajectory_safety_assessment_batch(..., shard=...)``, ``PlanningStep(..., shard=...)`` and ``bench.py
--gpus N`` all go through it -- block bounds, the two pre-allocated blocks of the collective (nothing is allocated per
step) and the collective itself.
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


class CostGather:
    """The trajectory split of one batch size over the ranks of a process group, and its one collective.

    ``lo, hi``: this rank's rows of the ``M`` candidates; ``mine``: the block this rank contributes (``per = ceil(M /
    world)`` rows, the last rank's padded with NaN -- ``all_gather_into_tensor`` wants equal blocks); ``gathered``: every
    rank's block in rank order.  Both live for as long as the object: a planning loop gathers thousands of times per
    second and must not allocate.  ``block()`` is where a rank's sweep should write its cost rows (the HIP sweep takes
    ``block()`` as its ``cost`` output, so nothing is copied before the collective); ``gather()`` queues the collective
    on the current stream and returns ``cost [M, 16]`` (a view of ``gathered``).

    Without an initialised process group (or with ``world_size`` 1 and ``force=False``) the object degenerates to the
    unsharded case: ``lo, hi = 0, M`` and ``gather()`` hands the block back."""

    def __init__(self, M: int, group=None, device=None, force: bool = False):
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.M = int(M)
        self.lo, self.hi = shard_bounds(self.M, self.world, self.rank)
        self.per = -(-self.M // self.world) if self.M > 0 else 0
        if device is None:
            device = (torch.device("cuda", torch.cuda.current_device())
                      if self.active and dist.get_backend(group) == "nccl" else torch.device("cpu"))
        self.device = torch.device(device)
        self.collective = self.active and (self.world > 1 or force)
        self.mine = torch.full((self.per, N.NC), float("nan"), dtype=torch.float64, device=self.device)
        self.gathered = (torch.empty((self.world * self.per, N.NC), dtype=torch.float64, device=self.device)
                         if self.collective else self.mine)
        self.calls = 0

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def block(self) -> torch.Tensor:
        """this rank's rows of ``mine``: [hi - lo, 16], contiguous (row-major: a leading slice of a contiguous tensor)"""
        return self.mine[: self.n_local]

    def gather(self, cost: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``cost``: this rank's cost rows if they were not written into ``block()`` directly.  Returns cost [M, 16] -- a VIEW
        of this object's gathered matrix, which the next ``gather()`` overwrites: valid until the next step of whoever
        reuses this CostGather (``PlanningStep.out.cost_all``, ``BatchAssessment.cost``); ``.clone()`` what must outlive it."""
        if cost is not None and self.n_local and cost.data_ptr() != self.mine.data_ptr():
            self.mine[: self.n_local].copy_(cost)
        if self.collective:
            dist.all_gather_into_tensor(self.gathered, self.mine, group=self.group)   # the one collective of the path
            self.calls += 1
        return self.gathered[: self.M]


def as_gather(shard, M: int, device=None) -> Optional[CostGather]:
    """what the ``shard=`` arguments of the product path accept: None / False (no split), True (the default process
    group), a process group, or a CostGather built for this batch size"""
    if shard is None or shard is False:
        return None
    if isinstance(shard, CostGather):
        if shard.M != int(M):
            raise ValueError(f"CostGather was built for {shard.M} trajectories, the batch has {M}")
        return shard
    return CostGather(M, group=None if shard is True else shard, device=device)


class ShardedAssessment:
    """``run(trajectories)``: shard -> ``compute(shard) -> cost [n,16]`` -> all-gather -> cost [M,16] on every rank.
    ``compute`` is any callable (the CPU tests put the oracle there; the product classes call the HIP sweep themselves and
    use :class:`CostGather` directly).  The blocks of the collective are kept between calls of one batch size."""

    def __init__(self, compute: Callable[[Dict], Optional[torch.Tensor]], group=None):
        self.compute = compute
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._cg = None

    def run(self, arrays: Dict) -> torch.Tensor:
        shard, (lo, hi), M = shard_arrays(arrays, self.world, self.rank)
        cost = self.compute(shard) if hi > lo else None
        device = cost.device if cost is not None else None
        if self._cg is None or self._cg.M != M or (device is not None and self._cg.device != device):
            self._cg = CostGather(M, group=self.group, device=device)
        return self._cg.gather(cost)


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
