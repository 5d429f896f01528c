"""Metric orchestrator + thresholds on top of the batched HIP sweep (mirrors ref: metrics/metric.py:18-147).

``Metric.evaluate_metrics(trajectory)`` keeps the reference's per-trajectory contract -- ``(results, safety_check)``
with the nested result dict of SURVEY Appendix B -- while the arithmetic for *all* candidate trajectories of a planning
step happens in one launch (``evaluate_batch``).  The per-trajectory call is served from the cached batch when the
trajectory object was part of it, otherwise a one-trajectory sweep is launched.  Threshold logic (metric.py:50-100) and
the dependency closure (:125-147) run inside libfo_hip.so (fo_reduce_kernel / fo_sweep_configure).
"""
import json
import os

import numpy as np
import torch

from .. import _native as N
from ..sweep import DEFAULT_HARM_COEFF, MetricSweep, SweepResult, list_views

AVAILABLE = ("dce", "cp", "ttc", "ttce", "wttc", "be", "hr")   # metric.py:109-117
_NP_DTYPE = {torch.float64: np.float64, torch.float32: np.float32, torch.int32: np.int32, torch.uint8: np.uint8}


def check_required_metrics(metric_names):
    """metric.py:125-147 (order of evaluation; the kernel computes everything in one pass, the order only decides
    the key order of the result dict)"""
    names = list(metric_names)
    if "wttc" in names:
        if "ttc" in names:
            names.remove("ttc")
        names.insert(0, "ttc")
    if "ttc" in names or "ttce" in names or "be" in names:
        if "dce" in names:
            names.remove("dce")
        names.insert(0, "dce")
    if "hr" in names:
        if "cp" in names:
            names.remove("cp")
        names.insert(0, "cp")
    return names


def load_harm_coeff(path=None):
    """the entries of harm_params.json the reference reads (hr.py:21-40, harm_model.py:109-155)"""
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "config", "harm_params.json")
    if not os.path.exists(path):
        return dict(DEFAULT_HARM_COEFF)
    with open(path) as f:
        d = json.load(f)
    lr = d["log_reg"]
    return dict(lr4s_const=lr["reduced_sym_angle_areas"]["const"], lr4s_speed=lr["reduced_sym_angle_areas"]["speed"],
                lr4s_side=lr["reduced_sym_angle_areas"]["side"], lr4s_rear=lr["reduced_sym_angle_areas"]["rear"],
                lr1s_const=lr["ignore_angle"]["const"], lr1s_speed=lr["ignore_angle"]["speed"],
                ped_const=d["pedestrian"]["const"], ped_speed=d["pedestrian"]["speed"])


def trajectories_to_arrays(trajectories):
    """list of duck-typed trajectories (``.cartesian.{x,y,theta,v,a}``, ref: collision_probability.py:32,97;
    harm_model.py:81-94) or a dict of [M,T] arrays/tensors -> dict of [M,T] float64"""
    if isinstance(trajectories, dict):
        return trajectories
    out = {}
    M = len(trajectories)
    if M == 0:
        return {k: np.zeros((0, 0)) for k in ("x", "y", "theta", "v", "a")}
    carts = [t.cartesian for t in trajectories]
    T = len(carts[0].x)
    for k in ("x", "y", "theta", "v", "a"):
        rows = [getattr(c, k) for c in carts]
        if any(len(r) != T for r in rows):
            raise ValueError("all trajectories of a batch must have the same number of samples")
        # one C-level concatenation instead of M small conversions (the planner hands over thousands of objects)
        out[k] = np.concatenate(rows, axis=None).astype(np.float64, copy=False).reshape(M, T)
    return out


class HostMirrorPool:
    """Host memory for the full mirror of a batch's per-pair outputs (what a planner that opens result sub-dicts makes
    ``BatchAssessment._to_host`` fetch: tens of MB per step for 2 000 x 32 x 30 lists).  Fresh pageable memory costs ~35 ms
    of page faults to fill and ~20 ms for the operating system to take back at the next step; a pinned buffer kept across
    steps costs the copy alone.  The arrays handed out are VIEWS of the pool: a buffer is written again only when nobody
    outside holds a view of it any more (the reference returns fresh arrays per call, and a planner may keep a step's
    results) -- otherwise the holder keeps that buffer and the pool takes a new one."""

    def __init__(self, pinned=True):
        self.pinned = bool(pinned)
        self.bufs = {}            # name -> (uint8 tensor, its numpy array)
        self.allocations = 0

    def fetch(self, name, src):
        """device tensor -> host array of the same shape and dtype (a view of the pool's buffer `name`)"""
        import sys
        src = src.contiguous()
        n = src.numel() * src.element_size()
        ent = self.bufs.get(name)
        if ent is not None and (ent[0].numel() < n or sys.getrefcount(ent[1]) > 2):    # (the tuple's reference + the call's)
            ent = None
        if ent is None:
            t = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=self.pinned)
            ent = self.bufs[name] = (t, t.numpy())
            self.allocations += 1
        t, base = ent
        t[:n].view(src.dtype).view(src.shape).copy_(src)
        return base[:n].view(_NP_DTYPE[src.dtype]).reshape(tuple(src.shape))


class BatchAssessment:
    """result of ``evaluate_batch``: device tensors of the sweep + accessors in the reference's vocabulary"""

    def __init__(self, res: SweepResult, prediction_slots, metric_order, mode):
        self.result = res
        self.cost, self.safe = res.cost, res.safe
        # A batch split over the ranks of a process group (evaluate_batch(..., shard=...)): `cost` / `safe` / column() cover
        # ALL trajectories on every rank (the all-gathered cost matrix); the per-pair outputs and result_dict() cover this
        # rank's rows [rows[0], rows[1]) and are addressed by the LOCAL index m - rows[0]
        self.rows = res.rows
        if res.cost_all is not None:
            self.cost = res.cost_all
            self.safe = (res.cost_all[:, N.COST["safe"]] > 0.5).to(torch.uint8)
        self.prediction_slots = prediction_slots
        self.metric_order = metric_order
        self.mode = mode
        self._host = None
        self._small = None
        self._fast = None
        self._hr_tpl = None
        self._pool = None                # HostMirrorPool of the Metric this batch belongs to (None: plain .cpu() copies)

    def __len__(self):
        return int(self.cost.shape[0])

    def safety(self):
        return self.safe.bool()

    def column(self, name):
        return self.cost[:, N.COST[name]]

    HOST_CACHE_BYTES = 256 << 20     # per-pair outputs up to this size are mirrored on the host in one copy

    def _host_small(self):
        """cost vectors and flags of the batch on the host (one small copy: what a planner that reads the flag and the six
        'hr' maxima per candidate needs -- the per-pair outputs stay in HBM until somebody opens a sub-dict)"""
        if self._small is None:
            r = self.result
            self._small = {"cost": r.cost.cpu().numpy(), "safe": r.safe.cpu().numpy()}
        return self._small

    def _to_host(self):
        if self._host is None:
            r = self.result
            big = r.lists_raw is not None and r.lists_raw.numel() * r.lists_raw.element_size() > self.HOST_CACHE_BYTES
            get = (lambda name, t: t.cpu().numpy()) if self._pool is None else self._pool.fetch
            self._host = dict(self._host_small(),
                              pair_f=None if (r.pair_f is None or big) else get("pair_f", r.pair_f),
                              pair_i=None if (r.pair_i is None or big) else get("pair_i", r.pair_i),
                              # one copy of the raw buffer; the five lists are strided views of it
                              lists=None if (r.lists_raw is None or big) else
                              list_views(get("lists", r.lists_raw), *r.lists_shape))
        return self._host

    def _column(self, m):
        """per-pair outputs of trajectory m as host arrays: pair_f [NPF, A], pair_i [NPI, A], lists [NL, A, T-1].
        Small batches are mirrored on the host once; for large ones (10 000 x 256 full outputs are 3.4 GB) only the
        trajectory's own column crosses PCIe, gathered on the device."""
        h = self._to_host()
        if h["lists"] is not None:
            return h["pair_f"][:, :, m], h["pair_i"][:, :, m], np.stack([v[:, :, m] for v in h["lists"]])
        r = self.result
        return (r.pair_f[:, :, m].cpu().numpy(), r.pair_i[:, :, m].cpu().numpy(),
                torch.stack([v[:, :, m] for v in r.list_views()]).cpu().numpy())

    def result_dict(self, m):
        """the reference's nested dict for trajectory m and its safety flag; needs mode 'full'.

        The dict is LAZY (:class:`LazyMetrics`): it has the reference's keys in the reference's order from the start, but
        a metric's sub-dict is cut from the batch only when it is first read -- a planner that asks trajectory by
        trajectory (interface.py:216-219) and looks at ``['hr']['max_obst_risk_all']`` or the flag pays microseconds per
        call instead of building thirty per-prediction entries it never opens."""
        if self.result.lists_raw is None:
            raise RuntimeError("result_dict needs the batch to be evaluated with mode='full'")
        if self._fast is None:
            # what every per-trajectory call of the step needs, made once: the flags and the cost rows as Python lists (a
            # numpy scalar read costs more than the rest of the call), the key template of the result dict
            h = self._host_small()
            # (ONE flat list of floats, row m at [16 m, 16 m + 16): a list per row would be 2 000 containers for the garbage
            # collector -- allocating them triggered a full collection, 16 ms inside the first call of a step)
            self._fast = (h["safe"].astype(bool).tolist(), h["cost"].ravel().tolist(), dict.fromkeys(self.metric_order, _UNBUILT))
        return LazyMetrics(self, m), self._fast[0][m]

    def _hr_template(self, lazy):
        """(key template of an 'hr' sub-dict, prediction id -> slot): which predictions carry an entry depends on the
        prediction alone (harm_model.py:65-66: none for an empty horizon), not on the trajectory -- made once per batch"""
        if self._hr_tpl is None:
            # (one row of the index block: whether a prediction has a harm model does not depend on the trajectory)
            valid = (self._host["pair_i"][N.PI["hr_valid"], :, 0] if self._host is not None and self._host["pair_i"] is not None
                     else self.result.pair_i[N.PI["hr_valid"], :, 0].cpu().numpy())
            slot = {pid: k for pid, k in self.prediction_slots if valid[k]}
            tpl = dict.fromkeys(slot, _UNBUILT)
            tpl.update(dict.fromkeys(LazyHR.ALL, 0.0))
            self._hr_tpl = (tpl, slot, [N.COST[key] for key in LazyHR.ALL])
        return self._hr_tpl

    def _build_metric(self, m, name, lazy):
        """sub-dict of metric `name` for trajectory m; ``lazy._column()`` = that trajectory's column (see _column), gathered
        on first need and shared by the metrics of one LazyMetrics ('wttc' and the six maxima of 'hr' do not need it)"""
        cost, base = self._fast[1], N.NC * m        # (result_dict made the flat list of cost rows)
        if name == "wttc":
            return cost[base + N.COST["wttc"]]
        if name == "hr":
            return LazyHR(self, base, cost, lazy)
        pf, pi, ls, n_valid, pf_l, pi_l = lazy._column()
        PF, PI, LST = N.PF, N.PI, N.LST
        slots = self.prediction_slots
        if name == "cp":
            row, nv = ls[LST["cp"]], n_valid[LST["cp"]]
            return {pid: row[k, :nv[k]] for pid, k in slots}          # views of this trajectory's own gather
        if name == "dce":
            d, t = pf_l[PF["dce"]], pi_l[PI["time_dce"]]
            return {pid: {"dce": d[k], "time_dce": t[k]} for pid, k in slots}
        if name == "ttc":
            v = pf_l[PF["ttc"]]
            return {pid: v[k] for pid, k in slots}
        if name == "ttce":
            v = pf_l[PF["ttce"]]
            return {pid: v[k] for pid, k in slots}
        if name == "be":
            d, b = pf_l[PF["be_decel"]], pf_l[PF["be_btn"]]
            return {pid: {"required_constant_deceleration": d[k], "break_threat_number": b[k]} for pid, k in slots}
        raise KeyError(name)

    def _hr_entry(self, k, col):
        pf, pi, ls, n_valid, pf_l, pi_l = col
        PF, PI, LST = N.PF, N.PI, N.LST
        i_er, i_or, i_eh, i_oh, i_cp = (LST[k_] for k_ in ("ego_risk", "obst_risk", "ego_harm", "obst_harm", "cp"))
        return {"max_ego_risk": pf_l[PF["max_ego_risk"]][k], "max_obst_risk": pf_l[PF["max_obst_risk"]][k],
                "max_obst_harm_with_cp": pf_l[PF["max_obst_harm_with_cp"]][k],
                "max_obst_risk_index": pi_l[PI["max_obst_risk_index"]][k], "max_ego_harm": pf_l[PF["max_ego_harm"]][k],
                "max_obst_harm": pf_l[PF["max_obst_harm"]][k],
                "ego_risk_traj": ls[i_er, k, :n_valid[i_er, k]].tolist(),      # the two lists the reference returns as lists
                "obst_risk_traj": ls[i_or, k, :n_valid[i_or, k]].tolist(),
                "ego_harm_traj": ls[i_eh, k, :n_valid[i_eh, k]], "obst_harm_traj": ls[i_oh, k, :n_valid[i_oh, k]],
                "collision_probability": ls[i_cp, k, :n_valid[i_cp, k]],
                "max_collision_probability": pf_l[PF["max_collision_probability"]][k]}


_UNBUILT = object()


class _LazyDict(dict):
    """a dict whose values are produced on first access: every key is there from the start (order, ``in`` and ``len`` are
    a plain dict's), unbuilt values are placeholders that every read replaces by the real thing.

    It stays a ``dict`` subclass (the reference returns plain dicts and a planner may test ``isinstance(r, dict)``), so
    the C-level shortcuts CPython takes for dicts must not see the placeholders: ``dict(m)``, ``{**m}`` and
    ``d.update(m)`` copy the raw storage only while ``tp_iter`` is dict's own -- overriding ``__iter__`` / ``keys`` sends
    them through ``keys()`` + ``__getitem__``; ``pop`` / ``popitem`` / ``setdefault`` / ``|`` are overridden one by one"""

    __slots__ = ()          # (subclasses name their few fields: no per-instance __dict__ for the garbage collector to track)

    def _build(self, key):
        raise NotImplementedError

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is _UNBUILT:
            v = self._build(key)
            dict.__setitem__(self, key, v)
        return v

    def __iter__(self):
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        if key in self:
            v = self[key]
            dict.__delitem__(self, key)
            return v
        if default:
            return default[0]
        raise KeyError(key)

    def popitem(self):
        if not len(self):
            raise KeyError("popitem(): dictionary is empty")
        k = next(reversed(self))
        return k, self.pop(k)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        dict.__setitem__(self, key, default)
        return default

    def __or__(self, other):
        if not isinstance(other, dict):
            return NotImplemented
        out = dict(self.materialize())
        out.update(other)
        return out

    def __ror__(self, other):
        if not isinstance(other, dict):
            return NotImplemented
        out = dict(other)
        out.update(self)
        return out

    def __ior__(self, other):
        dict.update(self, other)
        return self

    def materialize(self):
        for k in list(dict.keys(self)):
            v = self[k]
            if isinstance(v, _LazyDict):
                v.materialize()
        return self

    def values(self):
        return dict.values(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other.materialize() if isinstance(other, _LazyDict) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self.materialize())

    def copy(self):
        return dict(self.materialize())

    def __reduce__(self):
        return (dict, (dict(self.materialize()),))


class _Column:
    """the per-pair outputs of ONE trajectory, gathered from the batch on first need and shared by the sub-dicts of its
    result (a plain holder: the dicts point at it, it points at neither of them -- no reference cycle, so a result dict
    is freed when the planner drops it instead of waiting for the garbage collector; 2 000 cyclic results per step made the
    first pass over a batch cost 25 us per call instead of 3)"""
    __slots__ = ("batch", "m", "col")

    def __init__(self, batch, m):
        self.batch, self.m, self.col = batch, m, None

    def _column(self):
        if self.col is None:
            pf, pi, ls = self.batch._column(self.m)          # [NPF, A], [NPI, A], [NL, A, T-1]
            # entries past a list's length are NaN (hr.py:87-98 stops at min(T-1, len(prediction))): lengths per slot
            self.col = (pf, pi, ls, (~np.isnan(ls)).sum(axis=2), pf.tolist(), pi.tolist())
        return self.col


class LazyMetrics(_LazyDict):
    """result dict of one trajectory (keys = the activated metrics in the reference's order, metric.py:125-147)"""
    __slots__ = ("_batch", "_m", "_c")

    def __init__(self, batch, m):
        dict.__init__(self, batch._fast[2])          # the key template: every metric unbuilt
        self._batch, self._m, self._c = batch, m, _Column(batch, m)

    def _column(self):
        return self._c._column()

    def _build(self, key):
        return self._batch._build_metric(self._m, key, self._c)


class LazyHR(_LazyDict):
    """the 'hr' sub-dict: one entry per prediction with a harm model + the six maxima over all of them (hr.py:87-114)"""
    ALL = ("max_ego_risk_all", "max_obst_risk_all", "max_ego_harm_all", "max_obst_harm_all",
           "max_collision_probability_all", "max_obst_harm_with_cp_all")
    __slots__ = ("_slot", "_batch", "_lazy")

    def __init__(self, batch, base, cost, lazy):
        # the keys (one per prediction with a harm model, then the six maxima) come from a template made once per batch; a
        # call that reads the flag and the maxima -- what a planner does per candidate -- costs a dict copy and six stores
        # (cost: the batch's cost rows as one flat list, this trajectory's at [base, base + 16))
        tpl, self._slot, idx = batch._hr_template(lazy)
        dict.__init__(self, tpl)
        for key, i in zip(self.ALL, idx):
            dict.__setitem__(self, key, cost[base + i])
        self._batch, self._lazy = batch, lazy

    def _build(self, key):
        return self._batch._hr_entry(self._slot[key], self._lazy._column())


class Metric:
    def __init__(self, config, vehicle_params, agent_manager, dt=None, harm_coeff=None, device=0, ctx=None,
                 list_storage="f64"):
        self.config = config
        self.list_storage = list_storage     # accelerator.list_storage: element type of the per-timestep lists
        self.metric_thresholds = config["metric_thresholds"]
        self.vehicle_params = vehicle_params
        self.agent_manager = agent_manager
        names = list(config["activated_metrics"] or [])
        for n in names:
            if n not in AVAILABLE:
                raise ValueError(f"unknown metric '{n}'")
        self.metrics = check_required_metrics(names)           # ordered like the reference's dict
        self.dt = float(dt if dt is not None else agent_manager.dt)
        self.sweep = MetricSweep(vehicle_params, self.dt, metrics=self.metrics or ("dce",),
                                 thresholds=self.metric_thresholds, harm_coeff=harm_coeff or load_harm_coeff(),
                                 device=device, ctx=ctx)
        self._agents_version = None
        self._batch = None
        self._batch_ids = {}
        self._batch_objs = []
        self._mirror_pool = HostMirrorPool()      # host memory of the full per-pair mirror, kept across planning steps

    def invalidate(self):
        """call when the phantom set changed (FOInterface.evaluate_scenario does)"""
        self._agents_version = None
        self._batch = None
        self._batch_ids = {}
        self._batch_objs = []

    def agents_uploaded(self):
        """the planning step (fo_step_run) has written the sweep's agent table from the device batch of this step: the next
        evaluate_batch need not upload it again"""
        self._agents_version = 1

    def _upload_agents(self):
        if self._agents_version is not None:
            return
        arrs = self.agent_manager.sweep_arrays()
        # the blocking covariance check only where a caller's matrices can come in: the spawn kernels write diag(var, var)
        am = self.agent_manager
        self.sweep.set_agents(*arrs, check=bool(getattr(am, "_manual", True) or getattr(am, "_external", True)))
        self._agents_version = 1

    def evaluate_batch(self, trajectories, mode="reduced", remember=None, shard=None):
        """one launch for the whole candidate set.  trajectories: list of trajectory objects or dict of [M,T] arrays.

        ``shard`` (BASELINE configs[3]): True / a ``torch.distributed`` process group / a
        :class:`~frenetix_occlusion.distributed.CostGather` -- every rank calls this with the SAME candidate set and the same
        phantom agents, evaluates its contiguous block of the trajectories and contributes the block's cost rows to one
        all-gather; the returned assessment's ``cost`` / ``safe`` cover all M trajectories on every rank, its per-pair
        outputs this rank's rows (``BatchAssessment.rows``)."""
        am = self.agent_manager
        # (a device batch whose live count is still in HBM counts as "may have": asking would stall the step, and a sweep
        # over inactive slots leaves every trajectory safe)
        if not (am.may_have_phantoms() if hasattr(am, "may_have_phantoms") else am.has_phantoms()) or not self.metrics:
            return None                                          # metric.py:44-45: ({}, True) for every trajectory
        arr = None
        if not isinstance(trajectories, dict) and len(trajectories):
            # the planner's objects: packed into the pinned staging buffer by the native helper, one host-to-device copy
            arr = self.sweep.upload_trajectory_objects(trajectories)
        if arr is None:
            arr = trajectories_to_arrays(trajectories)
        self._upload_agents()
        cg = None
        if shard is not None and shard is not False:
            from ..distributed import as_gather
            cg = as_gather(shard, len(arr["x"]), device=self.sweep.device)
        if cg is None:
            res = self.sweep.run(arr["x"], arr["y"], arr["theta"], arr["v"], arr.get("a"), mode=mode, lists=self.list_storage)
        else:
            # this rank's block of the candidates; its cost rows are written straight into the block of the collective
            lo, hi = cg.lo, cg.hi
            loc = {k: (None if arr.get(k) is None else arr[k][lo:hi]) for k in ("x", "y", "theta", "v", "a")}
            T = int(arr["x"].shape[1]) if len(arr["x"]) else 0
            # (a collective over host memory -- gloo, two ranks on one GPU -- takes a copy of the rows instead)
            direct = cg.device == self.sweep.device
            res = self.sweep.alloc_out(hi - lo, T, self.sweep.A, mode, self.list_storage, cost=cg.block() if direct else None)
            if hi > lo:
                res = self.sweep.run(loc["x"], loc["y"], loc["theta"], loc["v"], loc["a"], mode=mode, out=res,
                                     lists=self.list_storage)
            res.cost_all, res.rows = cg.gather(None if direct or hi == lo else res.cost), (lo, hi)
        _ = self.agent_manager.predictions if mode == "full" else None
        slots = getattr(self.agent_manager, "prediction_slots", None) if mode == "full" else None
        ba = BatchAssessment(res, slots, self.metrics, mode)
        ba._pool = self._mirror_pool
        if remember is not None:
            self._batch = ba
            # strong references: id() is only unique among live objects, and the planner may drop its list and
            # build new trajectory objects (re-sampling at a higher density) before asking for single results
            # (a sharded batch remembers this rank's block: the per-trajectory results of the others live on their ranks)
            self._batch_objs = list(remember) if cg is None else list(remember)[cg.lo:cg.hi]
            self._batch_ids = {id(t): i for i, t in enumerate(self._batch_objs)}
        return ba

    def evaluate_metrics(self, trajectory):
        """reference contract (metric.py:35-100): ``(results, safety_check)`` for one trajectory"""
        if not self.agent_manager.has_phantoms() or not self.metrics:
            return {}, True
        m = self._batch_ids.get(id(trajectory)) if self._batch is not None else None
        if m is not None and self._batch_objs[m] is not trajectory:
            m = None
        if m is not None and self._batch.mode == "full":
            return self._batch.result_dict(m)
        ba = self.evaluate_batch([trajectory], mode="full")
        return ba.result_dict(0)
