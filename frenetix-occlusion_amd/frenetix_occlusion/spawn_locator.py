"""Phantom-agent spawn stage (host side of ``fo_scene_spawn`` / ``fo_scene_spawn_rules`` / ``fo_scene_spawn_rule_agents``,
include/fo_hip.h).

Mirrors the reference's ``SpawnLocator.find_spawn_points(ego_pos, ego_orientation, ego_pos_cl, ego_v)``
(ref: spawn_locator.py:80-139) and its ``SpawnPoint`` record (:18-27).  The reference finds <= ~5 spawn points with
three GEOS-based rule families; this build samples the *frontier* of the occluded cell set (occluded cells with a
visible 4-neighbour -- where a hidden road user would emerge from), gated like the reference: at least 3 m ahead of
the ego (:234,381) and not beyond max(4 v_ego, 25) m (:113, constants :65-66).  The kernel also writes the
constant-velocity predictions of the spawned agents straight into the layout ``fo_sweep_set_agents`` reads, so the
phantom set never leaves HBM between sampling and the metric sweep.
"""
import bisect
import math
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _native as N


MIN_AHEAD = 3.0            # spawn_locator.py:234,381 "ahead by >= 3 m"
S_THRESHOLD_TIME = 4.0     # spawn_locator.py:65-66,113: s_threshold = s_ego + max(4 v, 25)
S_THRESHOLD_MIN = 25.0

TYPE_NAME = {0: "Car", 1: "Truck", 3: "Bicycle", 4: "Pedestrian"}
TYPE_CODE = {"car": 0, "truck": 1, "bicycle": 3, "pedestrian": 4}


@dataclass
class SpawnPoint:
    """spawn_locator.py:18-27"""
    position: np.ndarray
    agent_type: str
    cl_pos: Optional[np.ndarray] = None
    source: str = "occluded frontier"
    orientation: Optional[float] = None


def intention_from_curvature(k) -> int:
    """``SpawnLocator._find_ego_intention`` (spawn_locator.py:729-741) behind its curvature call: 1 left turn if the largest
    curvature of the window exceeds 0.10 1/m, else 2 right turn if the smallest is below -0.10, else 0 straight ahead
    (pinned to the reference's own method: tests/golden/relevant_lanelets.npz)"""
    return 1 if k.max() > 0.10 else 2 if k.min() < -0.10 else 0


@dataclass
class PhantomBatch:
    """device-resident phantom predictions, exactly the argument list of ``MetricSweep.set_agents``.

    Agent axis: ``n_cell_agents`` slots of the cell sampler (``fo_scene_spawn``) first, then ``n_rule_points`` slots of the
    reference's rule families (``fo_scene_spawn_rules`` -> ``fo_scene_spawn_rule_agents``); every agent owns ``R``
    consecutive prediction slots (one per candidate route of a phantom vehicle)."""
    n: torch.Tensor          # int32 [1]  number of active cell-sampled agents
    cell: torch.Tensor       # int32 [n_cell_agents] window cell index (-1 = unused slot)
    pos0: torch.Tensor       # [agents,2]     agents = n_cell_agents + n_rule_points
    yaw0: torch.Tensor       # [agents]
    pos: torch.Tensor        # [slots,T,2]    slots = agents * R
    yaw: torch.Tensor        # [slots,T]
    v: torch.Tensor          # [slots,T]
    cov: torch.Tensor        # [slots,T,2,2]
    shape: torch.Tensor      # [slots,2] inflated
    raw_dims: torch.Tensor   # [slots,2]
    type: torch.Tensor       # int32 [slots]
    len: torch.Tensor        # int32 [slots]  (valid samples of the prediction, 0 = unused slot)
    R: int = 1               # prediction slots per agent: slot j * R + r (r = candidate route of a phantom vehicle)
    head: Optional[torch.Tensor] = None   # uint8: one allocation backing pos0 | yaw0 | rule_points | n, rule_n | type
    n_cell_agents: int = 0
    n_rule_points: int = 0
    rule_points: Optional[torch.Tensor] = None   # [n_rule_points, 8] records of fo_scene_spawn_rules
    rule_n: Optional[torch.Tensor] = None        # int32 [1]
    body: Optional[torch.Tensor] = None   # uint8: one allocation backing pos | yaw | v | cov | shape | raw_dims | len
    _host: Optional[dict] = None                 # host copy of the head for the current step (lazy views)
    _host_body: Optional[dict] = None            # host copy of the body (the predictions) for the current step
    _pending: Optional[object] = None            # weak reference to the step's unread LazySpawnPoints
    step: int = 0                                # planning steps queued on this batch (bumped by invalidate)

    def sweep_args(self):
        return self.pos, self.yaw, self.v, self.cov, self.shape, self.raw_dims, self.type, self.len

    def invalidate(self):
        """called before the NEXT step's kernels are queued on these buffers.  A lazily read spawn-point list of the step
        that ends here which somebody still holds (a planner that logs or plots ``spawn_points`` late) is read back now,
        while the buffers still hold its step -- the locator and the interface have dropped their own references by then, so
        a live weak reference means an outside holder; nobody holding one costs nothing."""
        old = self._pending() if self._pending is not None else None
        if old is not None:
            old._get()
        self._pending = None
        self._host = None
        self._host_body = None
        self.step += 1

    def host_body(self):
        """dict(pos [slots,T,2], yaw [slots,T], v [slots,T], cov [slots,T,2,2], shape [slots,2], raw_dims [slots,2], len [slots])
        on the host with ONE device-to-host copy per step (the reference's views of the phantom set -- ``phantom_agents``,
        ``predictions`` -- are cut from it; cached until :meth:`invalidate`)"""
        if self._host_body is None:
            S_, T = self.pos.shape[0], self.pos.shape[1]
            h = self.body.cpu().numpy()
            o, out = 0, {}
            for name, shp in (("pos", (S_, T, 2)), ("yaw", (S_, T)), ("v", (S_, T)), ("cov", (S_, T, 2, 2)), ("shape", (S_, 2)),
                              ("raw_dims", (S_, 2))):
                n = int(np.prod(shp)) * 8
                out[name] = h[o:o + n].view(np.float64).reshape(shp)
                o += n
            out["len"] = h[o:o + 4 * S_].view(np.int32)
            self._host_body = out
        return self._host_body

    def host_head(self):
        """dict(n, rule_n, pos0 [agents,2], yaw0 [agents], rule_points [n_rule_points,8], type [slots]) on the host with ONE
        device-to-host copy per step (cached until :meth:`invalidate`)"""
        if self._host is None:
            A, S_, Rp = self.pos0.shape[0], self.type.shape[0], self.n_rule_points
            h = self.head.cpu().numpy()
            o1, o2 = A * 16, A * 24
            o3 = o2 + Rp * 64
            o4 = o3 + 8
            cnt = h[o3:o4].view(np.int32)
            if Rp and int(cnt[1]) < 0:
                raise RuntimeError("fo_scene_spawn_rules: a rule family ran out of table space at this step (a sampled line of more "
                                   "than 1 024 cells / 8: include/fo_hip.h) -- there is no spawn-point list; use cells of >= 0.32 m")
            self._host = dict(n=int(cnt[0]) if self.n_cell_agents else 0, rule_n=min(int(cnt[1]), Rp) if Rp else 0,
                              pos0=h[:o1].view(np.float64).reshape(A, 2), yaw0=h[o1:o2].view(np.float64),
                              rule_points=h[o2:o3].view(np.float64).reshape(Rp, 8), type=h[o4:o4 + 4 * S_].view(np.int32))
        return self._host

    def live_agents(self):
        """agent indices (rows of pos0 / yaw0; first prediction slot = index * R) that are alive this step, cell-sampled
        agents first, then the rule families' in the reference's order"""
        h = self.host_head()
        return list(range(h["n"])) + [self.n_cell_agents + i for i in range(h["rule_n"])]


class LazySpawnPoints(list):
    """``FOInterface.spawn_points`` / the result of ``find_spawn_points(lazy=True)``: the reference's list of
    :class:`SpawnPoint` (interface.py:186), read back from the device when it is first looked at -- the planning step
    itself never waits for it.

    A ``list`` subclass (the reference returns a plain list: ``isinstance(x, list)``, ``x + [...]``, ``[...] + x``, ``in``,
    slicing, ``==``, ``sorted`` ... keep working): every Python-level access fills the list's own storage once.  C code
    that reads a list's storage directly without going through its methods (``json.dumps``) sees what has been filled so
    far -- call ``len()`` first.  A list that is still unread when the next planning step is queued on the same buffers is
    read back at that moment (``PhantomBatch.invalidate``), so a late reader gets ITS step's points; one that was created
    for a step whose buffers are gone raises."""

    def __init__(self, build, batch=None):
        super().__init__()
        self._build, self._batch = build, batch
        self._step = batch.step if batch is not None else None

    @property
    def materialised(self):
        return self._build is None

    def _get(self):
        if self._build is not None:
            if self._batch is not None and self._batch.step != self._step:
                raise RuntimeError("LazySpawnPoints: the device buffers of this list's planning step have been reused "
                                   "(the list was not alive when the next step was queued)")
            build, self._build, self._batch = self._build, None, None
            list.extend(self, build())
        return self

    def _f(name):     # noqa: N805 -- every read-side list method fills the storage first
        base = getattr(list, name)

        def method(self, *a, **k):
            self._get()
            return base(self, *a, **k)
        method.__name__ = name
        return method

    for _n in ("__len__", "__getitem__", "__iter__", "__contains__", "__reversed__", "__add__", "__mul__", "__rmul__",
               "__iadd__", "__imul__", "__setitem__", "__delitem__", "__lt__", "__le__", "__gt__", "__ge__",
               "append", "extend", "insert", "pop", "remove", "clear", "index", "count", "sort", "reverse", "copy"):
        locals()[_n] = _f(_n)
    del _n, _f

    def __radd__(self, other):
        return list(other) + list(self._get())

    def __eq__(self, other):
        self._get()
        if isinstance(other, LazySpawnPoints):
            other._get()
        return list.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __bool__(self):
        return len(self) > 0

    def __repr__(self):
        return list.__repr__(self._get())

    def __reduce__(self):
        return (list, (list(self._get()),))


class SpawnLocator:
    SOURCE_NAME = {1: "behind_dynamic_obstacle", 2: "behind static obstacle", 3: "left turn", 4: "right turn"}

    def __init__(self, agent_manager, ref_path, config, sensor_model, cosy_cl=None, fo_obstacles=None,
                 visualization=None, debug=False, max_agents=None, pattern=None, dt=0.1, horizon=3.0):
        self.agent_manager = agent_manager
        self.ref_path = np.ascontiguousarray(ref_path, dtype=np.float64)
        self.config = config
        self.cosy_cl = cosy_cl
        self.sensor_model = sensor_model
        self.fo_obstacles = fo_obstacles
        self.visualization = visualization
        self.debug = debug
        self.ctx = sensor_model.ctx
        self.device = sensor_model.device
        self._dev_index = self.device.index or 0
        acc = (config.get("accelerator") or {}).get("spawn", {}) if isinstance(config, dict) else {}
        self.max_agents = int(max_agents if max_agents is not None else acc.get("max_agents", 32))
        self.pattern = list(pattern if pattern is not None else acc.get("pattern",
                                                                         ["Pedestrian", "Pedestrian", "Bicycle", "Car"]))
        if len(self.pattern) != 4:
            raise ValueError("spawn pattern needs exactly 4 agent types")
        self.dt = float(dt)
        self.T = int(horizon / self.dt) + 1                      # agent.py:496
        self.min_ahead = float(acc.get("min_ahead", MIN_AHEAD))
        self.mode = str(acc.get("mode", "rules"))                  # "rules" (the reference's semantics) | "cells" | "both"
        if self.mode not in ("cells", "rules", "both"):
            raise ValueError("accelerator.spawn.mode must be 'cells', 'rules' or 'both'")
        # capacity of the rule families' output: the maxima of the YAML are compared with '>' BEFORE appending and a dynamic
        # obstacle can yield a Car and a Bicycle (Q11, spawn_locator.py:212,304-309,365), so the three families emit up to
        # (max_dynamic + 2) + (max_static + 1) + 1 points -- never less room than that, whatever the YAML says
        sl_cfg = (config.get("spawn_locator") or {}) if isinstance(config, dict) else {}
        self.max_rule_points = max(int(acc.get("max_rule_points", 8)),
                                   int(sl_cfg.get("max_dynamic_spawn_points", 1)) + int(sl_cfg.get("max_static_spawn_points", 1)) + 4)
        self.routes = int(acc.get("routes", 0)) if sensor_model.route_table is not None else 0
        if self.routes == 0 and sensor_model.route_table is not None and self.mode != "cells":
            self.routes = int(sensor_model.route_table.R)       # rule vehicles follow their lanelet's routes (agent.py:283-312)
        self.R = max(self.routes, 1)                               # prediction slots per agent
        self.all_occluded = bool(acc.get("all_occluded", False))   # False: only the visible/occluded frontier
        self.max_dist_override = acc.get("max_dist")               # None: the reference's max(4 v, 25) m
        am = config["agent_manager"]
        pr = am["prediction"]
        self.var0, self.var_factor = 0.1, float(pr["variance_factor"])   # agent.py:416-417,527-528
        t4, s4, rl, rw, il, iw = [], [], [], [], [], []
        for name in self.pattern:
            key = name.lower()
            if key not in TYPE_CODE:
                raise NotImplementedError(f'SpawnLocator: Agent type "{name}" is not implemented!')   # agent.py:120
            c = am[key]
            big = key == "bicycle"                                        # agent.py:402-405
            fl = pr["size_factor_length_l"] if big else pr["size_factor_length_s"]
            fw = pr["size_factor_width_l"] if big else pr["size_factor_width_s"]
            t4.append(TYPE_CODE[key]); s4.append(float(c["default_velocity"]))
            rl.append(float(c["length"])); rw.append(float(c["width"]))
            il.append(float(c["length"]) * fl); iw.append(float(c["width"]) * fw)
        self._t4 = np.array(t4, dtype=np.int32)
        self._s4, self._rl, self._rw = np.array(s4), np.array(rl), np.array(rw)
        self._il, self._iw = np.array(il), np.array(iw)
        # the three agent types the rule families spawn (fo_rule_agent_types_t: 0 Car, 1 Bicycle, 2 Pedestrian)
        self.rule_types = N.RuleAgentTypes()
        for i, key in enumerate(("car", "bicycle", "pedestrian")):
            c = am.get(key)
            if c is None:
                continue
            big = key == "bicycle"
            fl = pr["size_factor_length_l"] if big else pr["size_factor_length_s"]
            fw = pr["size_factor_width_l"] if big else pr["size_factor_width_s"]
            self.rule_types.speed[i] = float(c["default_velocity"])
            self.rule_types.raw_l[i], self.rule_types.raw_w[i] = float(c["length"]), float(c["width"])
            self.rule_types.infl_l[i], self.rule_types.infl_w[i] = float(c["length"]) * fl, float(c["width"]) * fw
        self._d_path = torch.as_tensor(self.ref_path).to(self.device)
        self.batch: Optional[PhantomBatch] = None
        self.spawn_points = []
        self._rule_points, self._n_cell_points = [], 0

    def _settle(self):
        if isinstance(self.spawn_points, LazySpawnPoints):
            self.spawn_points._get()

    @property
    def rule_points(self):
        """the spawn points of the rule families among ``spawn_points`` (reads them back if that has not happened yet)"""
        self._settle()
        return self._rule_points

    @property
    def n_cell_points(self):
        self._settle()
        return self._n_cell_points

    @property
    def n_cell_agents(self):
        return self.max_agents if self.mode in ("cells", "both") else 0

    @property
    def n_rule_points(self):
        return self.max_rule_points if self.mode in ("rules", "both") else 0

    def _alloc(self):
        dev, T = self.device, self.T
        Ac, Rp = self.n_cell_agents, self.n_rule_points
        A = Ac + Rp
        S_ = A * self.R                                            # prediction slots: slot j * R + r
        f = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
        i = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        # what the host views read back lives in one allocation: pos0 | yaw0 | rule points | n, rule_n | type
        o1, o2 = A * 16, A * 24
        o3 = o2 + Rp * 64
        o4 = o3 + 8
        head = torch.zeros(o4 + 4 * S_ + 4, dtype=torch.uint8, device=dev)
        cnt = head[o3:o4].view(torch.int32)
        # the predictions live in one allocation as well: pos | yaw | v | cov | shape | raw_dims (float64) | len (int32) -- the host
        # views of a step (FOAgentManager.phantom_agents / .predictions) cost one copy instead of ten
        sizes = [S_ * T * 2, S_ * T, S_ * T, S_ * T * 4, S_ * 2, S_ * 2]
        body = torch.zeros(8 * sum(sizes) + 4 * S_, dtype=torch.uint8, device=dev)
        offs = np.concatenate(([0], np.cumsum(sizes))) * 8
        fv = lambda k, *shape: body[int(offs[k]):int(offs[k + 1])].view(torch.float64).view(*shape)
        return PhantomBatch(n=cnt[0:1], cell=i(max(Ac, 1)), pos0=head[:o1].view(torch.float64).view(A, 2),
                            yaw0=head[o1:o2].view(torch.float64), pos=fv(0, S_, T, 2), yaw=fv(1, S_, T), v=fv(2, S_, T),
                            cov=fv(3, S_, T, 2, 2), shape=fv(4, S_, 2), raw_dims=fv(5, S_, 2),
                            type=head[o4:o4 + 4 * S_].view(torch.int32), len=body[int(offs[6]):].view(torch.int32), R=self.R,
                            head=head, body=body,
                            n_cell_agents=Ac, n_rule_points=Rp,
                            rule_points=head[o2:o3].view(torch.float64).view(Rp, 8) if Rp else None,
                            rule_n=cnt[1:2] if Rp else None)

    def max_distance(self, ego_v):
        if self.max_dist_override is not None:
            return float(self.max_dist_override)
        return max(S_THRESHOLD_TIME * float(ego_v), S_THRESHOLD_MIN)

    def _batch_for_step(self):
        if self.batch is None:
            self.batch = self._alloc()
        self.batch.invalidate()
        return self.batch

    def sample(self, ego_pos, ego_orientation, ego_v) -> PhantomBatch:
        """device-only path: frontier candidates -> evenly spaced pick -> headings -> predictions (no host sync)"""
        sm = self.sensor_model
        if sm.cell_class is None:
            raise RuntimeError("SpawnLocator: call SensorModel.calc_visible_and_occluded_area first")
        if self.n_cell_agents == 0:
            raise RuntimeError("SpawnLocator.sample: spawn.mode 'rules' has no cell-sampled agents")
        b, w = self._batch_for_step(), sm.window
        c = lambda a: a.ctypes.data
        self.ctx.call("fo_scene_spawn", sm.cell_class.data_ptr(), w.ix0, w.iy0, w.nx, w.ny, float(ego_pos[0]),
                      float(ego_pos[1]), math.cos(ego_orientation), math.sin(ego_orientation), self.min_ahead,
                      self.max_distance(ego_v), 1 if self.all_occluded else 0, self.max_agents, self.routes, c(self._t4), c(self._s4), c(self._rl), c(self._rw),
                      c(self._il), c(self._iw), int(self.ref_path.shape[0]), self._d_path.data_ptr(), self.T, self.dt,
                      self.var0, self.var_factor, b.cell.data_ptr(), b.pos0.data_ptr(), b.yaw0.data_ptr(),
                      b.n.data_ptr(), b.pos.data_ptr(), b.yaw.data_ptr(), b.v.data_ptr(), b.cov.data_ptr(),
                      b.shape.data_ptr(), b.raw_dims.data_ptr(), b.type.data_ptr(), b.len.data_ptr(),
                      N.current_stream(self._dev_index))
        return b

    # ---------------------------------------------------------------- the reference's rule families, on the device
    def _rule_setup(self):
        """one-off: the polyline frame of the reference path as the table fo_scene_spawn_rules reads, the switches and
        maxima of the YAML (the lanelet topology belongs to the static map: SensorModel._set_topology)"""
        from .scenario import lanelets_of
        from .utils.curvilinear import PolylineCS
        sm = self.sensor_model
        self._cs = PolylineCS(self.ref_path)
        self._s_list = self._cs.s.tolist()
        n = len(self.ref_path)
        tab = np.zeros((n, 6))
        tab[:, :2], tab[:, 2] = self._cs.path, self._cs.s
        tab[:-1, 3], tab[:-1, 4:6] = self._cs.seg_len, self._cs.tangent
        self._d_path6 = torch.as_tensor(tab).to(self.device)
        try:
            P = len(lanelets_of(sm.lanelet_network))
        except Exception:
            P = 0
        sl = self.config["spawn_locator"]
        ped = self.config["agent_manager"]["pedestrian"]
        self._rule_cfg = dict(behind_static=int(bool(sl.get("spawn_point_behind_static_obstacle", True))),
                              behind_turn=int(bool(sl.get("spawn_points_behind_turn", True))),
                              behind_dynamic=int(bool(sl.get("spawn_point_behind_dynamic_obstacle", True)) and P > 0),
                              max_static=int(sl.get("max_static_spawn_points", 1)),
                              max_dynamic=int(sl.get("max_dynamic_spawn_points", 1)),
                              ped_width=float(ped["width"]), ped_length=float(ped["length"]))
        # the turn rule samples its 40 m window every cell / 8 and holds 1 024 samples (include/fo_hip.h: a longer line comes back
        # as a refused point count, which only whoever READS the step's list gets to see): say so once, up front
        cs = float(getattr(sm, "cell_size", 0.5))
        if self._rule_cfg["behind_turn"] and 40.0 / (cs / 8.0) + 2.0 > 1024.0:
            import warnings
            warnings.warn(f"SpawnLocator: cells of {cs} m -- the turn rule of the spawn locator samples the 40 m reference window "
                          "every cell / 8 and holds 1 024 samples (cells >= 0.32 m); at a turn the step's spawn-point list will be "
                          "refused (RuntimeError when it is read).  Use larger cells or spawn_points_behind_turn: False.")
        self._rules_ready = True

    def _nearest_vertex(self, s):
        """``np.argmin(np.abs(ref_s - s))`` (spawn_locator.py:684-687: the first index of the smallest distance) by bisection
        on the path's arc lengths -- this runs every planning step on the host"""
        sl = self._s_list
        i = bisect.bisect_left(sl, s)
        if i <= 0:
            return 0
        if i >= len(sl):
            return len(sl) - 1
        return i - 1 if abs(sl[i - 1] - s) <= abs(sl[i] - s) else i

    def rule_params(self, ego_pos, ego_orientation, ego_pos_cl, ego_v):
        """the step's scalars of the rule families (``fo_spawn_rule_params_t``): curvilinear ego position, ``s_threshold``
        (spawn_locator.py:113), the reference window and the ego's intention from the curvature of the next 40 m of the
        reference path (:678-741) -- a few host operations on cached tables, nothing is read from the device"""
        from .utils.curvilinear import curvature
        if not getattr(self, "_rules_ready", False):
            self._rule_setup()
        if ego_pos_cl is None:
            ego_pos_cl = self._cs.convert_to_curvilinear_coords(float(ego_pos[0]), float(ego_pos[1]))
        s_ego = float(ego_pos_cl[0])
        i0, i1 = self._nearest_vertex(s_ego), self._nearest_vertex(s_ego + 40.0)   # :678-693
        intention = 0
        if i1 - i0 >= 3:                                                  # :729-741
            key = (i0, i1)
            if getattr(self, "_intent_key", None) != key:
                k = curvature(self.ref_path[i0:i1])
                self._intent = intention_from_curvature(k)
                self._intent_key = key
            intention = self._intent
        self.last_intention = ("straight ahead", "left turn", "right turn")[intention]
        c = self._rule_cfg
        # how many of this step's obstacles can take the dynamic rule at all (host flags: present, dynamic role, no bicycle /
        # pedestrian): the rule's big workgroups are launched for those only (fo_spawn_rule_params_t::n_dynamic_plus1)
        n_dyn = getattr(self.sensor_model, "_n_dyn_candidates", None)
        return N.SpawnRuleParams(float(ego_pos[0]), float(ego_pos[1]), float(ego_orientation), s_ego, float(ego_pos_cl[1]),
                                 s_ego + max(float(ego_v) * S_THRESHOLD_TIME, S_THRESHOLD_MIN), c["ped_width"], c["ped_length"],
                                 intention, i0, i1, c["behind_static"], c["behind_turn"], c["behind_dynamic"], c["max_static"],
                                 c["max_dynamic"], 0 if n_dyn is None else n_dyn + 1, 0)

    def rule_obstacle_ptrs(self):
        """(O, corners, centres, headings, dimensions, flags, visibility) device pointers of this step's obstacles as the rule
        kernels read them: everything ``SensorModel.upload_obstacles`` put into HBM from ``FOObstacles.arrays_full``"""
        sm = self.sensor_model
        d_corn, d_cen, d_flags, O = getattr(sm, "_obst", (None, None, None, 0))
        if O == 0:
            return (0,) + (None,) * 6
        rl = getattr(sm, "_obst_rule", None)
        if rl is None:
            raise RuntimeError("SpawnLocator: the spawn rule families need the obstacles' headings and dimensions "
                               "(SensorModel.upload_obstacles with an FOObstacles)")
        return O, d_corn.data_ptr(), d_cen.data_ptr(), rl[0].data_ptr(), rl[1].data_ptr(), d_flags.data_ptr(), sm._buf["ovis"].data_ptr()

    def queue_rules(self, ego_pos, ego_orientation, ego_pos_cl, ego_v) -> PhantomBatch:
        """``spawn.mode: rules | both``, device-only: the three rule families of the reference on the cell classes of this
        step (``fo_scene_spawn_rules``), then their spawn points become phantom agents with predictions in the sweep's layout
        (``fo_scene_spawn_rule_agents``) -- the reference's find_spawn_points -> add_agent flow (interface.py:186-198) without a
        read-back.  The host supplies the step's scalars (:meth:`rule_params`)."""
        import ctypes as C
        sm = self.sensor_model
        if sm.cell_class is None:
            raise RuntimeError("SpawnLocator: call SensorModel.calc_visible_and_occluded_area first")
        if self.n_rule_points == 0:
            raise RuntimeError("SpawnLocator.queue_rules: spawn.mode 'cells' has no rule stage")
        pr = self.rule_params(ego_pos, ego_orientation, ego_pos_cl, ego_v)
        b = self.batch if self.batch is not None else self._batch_for_step()
        b.invalidate()
        O, corn, cen, oyaw, odims, ofl, ovis = self.rule_obstacle_ptrs()
        w, st = sm.window, N.current_stream(self._dev_index)
        self.ctx.call("fo_scene_spawn_rules", sm.cell_class.data_ptr(), w.ix0, w.iy0, w.nx, w.ny, int(self._d_path6.shape[0]),
                      self._d_path6.data_ptr(), O, corn, cen, oyaw, odims, ofl, ovis, C.byref(pr), b.n_rule_points,
                      b.rule_points.data_ptr(), b.rule_n.data_ptr(), st)
        a0, s0, T = b.n_cell_agents, b.n_cell_agents * b.R, self.T
        self.ctx.call("fo_scene_spawn_rule_agents", b.n_rule_points, b.rule_points.data_ptr(), b.rule_n.data_ptr(), self.routes,
                      C.byref(self.rule_types), int(self.ref_path.shape[0]), self._d_path.data_ptr(), T, self.dt, self.var0,
                      self.var_factor, b.pos0[a0:].data_ptr(), b.yaw0[a0:].data_ptr(), b.pos[s0:].data_ptr(), b.yaw[s0:].data_ptr(),
                      b.v[s0:].data_ptr(), b.cov[s0:].data_ptr(), b.shape[s0:].data_ptr(), b.raw_dims[s0:].data_ptr(),
                      b.type[s0:].data_ptr(), b.len[s0:].data_ptr(), st)
        return b

    def _points_from_head(self, b):
        """the reference's SpawnPoint list from the batch's host head (one device-to-host copy per step)"""
        h = b.host_head()
        cells, rules = [], []
        typ = h["type"][:: b.R]                # agent j's type = type of its first prediction slot
        for j in range(h["n"]):
            cl = None
            if self.cosy_cl is not None:
                try:
                    cl = np.asarray(self.cosy_cl.convert_to_curvilinear_coords(h["pos0"][j, 0], h["pos0"][j, 1]))
                except Exception:   # out of the projection domain: the reference skips silently (:228-231)
                    cl = None
            cells.append(SpawnPoint(h["pos0"][j].copy(), TYPE_NAME[int(typ[j])], cl, "occluded frontier", float(h["yaw0"][j])))
        obst = list(self.fo_obstacles) if self.fo_obstacles is not None else []
        for q in h["rule_points"][:h["rule_n"]]:
            src = self.SOURCE_NAME[int(q[6])]
            if int(q[6]) == 2:
                src += " " + str(obst[int(q[7])].obstacle_id)             # spawn_locator.py:462
            cl = None if np.isnan(q[4]) else np.array([q[4], q[5]])
            rules.append(SpawnPoint(np.array([q[1], q[2]]), TYPE_NAME[int(q[0])], cl, src, None if np.isnan(q[3]) else float(q[3])))
        return cells, rules

    def _materialize(self, b):
        cells, rules = self._points_from_head(b)
        self._n_cell_points, self._rule_points = len(cells), rules
        return cells + rules

    def lazy_spawn_points(self):
        """the spawn-point list of a step that was queued elsewhere (``PlanningStep.run`` / ``fo_step_run`` on this locator's
        batch): the reference's list, read back from the device when first looked at"""
        b = self.batch
        self._rule_points, self._n_cell_points = [], 0
        self.spawn_points = LazySpawnPoints(lambda: self._materialize(b), b)
        b._pending = weakref.ref(self.spawn_points)
        return self.spawn_points

    def find_spawn_points(self, ego_pos, ego_orientation, ego_pos_cl, ego_v, lazy=False):
        """reference signature (spawn_locator.py:80): returns list[SpawnPoint].  Everything is decided on the device --
        cell-sampled points by ``fo_scene_spawn``, the reference's three rule families (mode 'rules' / 'both') by
        ``fo_scene_spawn_rules`` -- and the agents with their predictions are written there too; the list is the host's view
        of it.  ``lazy=True`` (what FOInterface.evaluate_scenario uses) returns a :class:`LazySpawnPoints` that reads the
        points back when first looked at, so that the step itself never waits for the device."""
        self.spawn_points, self._rule_points, self._n_cell_points = [], [], 0
        b = self._batch_for_step()
        if self.mode in ("cells", "both"):
            self.sample(ego_pos, ego_orientation, ego_v)
        if self.mode in ("rules", "both"):
            self.queue_rules(ego_pos, ego_orientation, ego_pos_cl, ego_v)
        if lazy:
            self.spawn_points = LazySpawnPoints(lambda: self._materialize(b), b)
            b._pending = weakref.ref(self.spawn_points)
        else:
            self.spawn_points = self._materialize(b)
        return self.spawn_points
