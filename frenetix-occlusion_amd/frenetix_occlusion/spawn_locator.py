"""Phantom-agent spawn sampling in the occluded cells (host side of ``fo_scene_spawn``, include/fo_hip.h).

Mirrors the reference's ``SpawnLocator.find_spawn_points(ego_pos, ego_orientation, ego_pos_cl, ego_v)``
(ref: spawn_locator.py:80-139) and its ``SpawnPoint`` record (:18-27).  The reference finds <= ~5 spawn points with
three GEOS-based rule families; this build samples the *frontier* of the occluded cell set (occluded cells with a
visible 4-neighbour -- where a hidden road user would emerge from), gated like the reference: at least 3 m ahead of
the ego (:234,381) and not beyond max(4 v_ego, 25) m (:113, constants :65-66).  The kernel also writes the
constant-velocity predictions of the spawned agents straight into the layout ``fo_sweep_set_agents`` reads, so the
phantom set never leaves HBM between sampling and the metric sweep.
"""
import math
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


@dataclass
class PhantomBatch:
    """device-resident phantom predictions, exactly the argument list of ``MetricSweep.set_agents``"""
    n: torch.Tensor          # int32 [1]  number of active slots
    cell: torch.Tensor       # int32 [max_agents] window cell index (-1 = unused slot)
    pos0: torch.Tensor       # [max_agents,2]
    yaw0: torch.Tensor       # [max_agents]
    pos: torch.Tensor        # [slots,T,2]      slots = max_agents * R
    yaw: torch.Tensor        # [slots,T]
    v: torch.Tensor          # [slots,T]
    cov: torch.Tensor        # [slots,T,2,2]
    shape: torch.Tensor      # [slots,2] inflated
    raw_dims: torch.Tensor   # [slots,2]
    type: torch.Tensor       # int32 [slots]
    len: torch.Tensor        # int32 [slots]  (valid samples of the prediction, 0 = unused slot)
    R: int = 1               # prediction slots per agent: slot j * R + r (r = candidate route of a phantom vehicle)
    head: Optional[torch.Tensor] = None   # uint8: one allocation backing pos0 | yaw0 | n | type (one copy to the host)

    def sweep_args(self):
        return self.pos, self.yaw, self.v, self.cov, self.shape, self.raw_dims, self.type, self.len

    def host_head(self):
        """(n, pos0 [A,2], yaw0 [A], type [slots]) on the host with ONE device-to-host copy"""
        A, S_ = self.pos0.shape[0], self.type.shape[0]
        if self.head is None:
            return int(self.n.item()), self.pos0.cpu().numpy(), self.yaw0.cpu().numpy(), self.type.cpu().numpy()
        h = self.head.cpu().numpy()
        o1, o2, o3 = A * 16, A * 24, A * 24 + 8
        return (int(h[o2:o2 + 4].view(np.int32)[0]), h[:o1].view(np.float64).reshape(A, 2), h[o1:o2].view(np.float64),
                h[o3:o3 + 4 * S_].view(np.int32))


class SpawnLocator:
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
        self.mode = str(acc.get("mode", "cells"))                  # "cells" | "rules" | "both" (fo_scene_spawn_rules)
        if self.mode not in ("cells", "rules", "both"):
            raise ValueError("accelerator.spawn.mode must be 'cells', 'rules' or 'both'")
        self.rule_points = []
        self._rules = None
        self.routes = int(acc.get("routes", 0)) if sensor_model.route_table is not None else 0
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
        self._d_path = torch.as_tensor(self.ref_path).to(self.device)
        self.batch: Optional[PhantomBatch] = None
        self.spawn_points = []

    def _alloc(self):
        dev, A, T = self.device, self.max_agents, self.T
        S_ = A * self.R                                            # prediction slots: slot j * R + r
        f = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
        i = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        # what find_spawn_points reads back lives in one allocation: pos0 | yaw0 | n (+ pad) | type
        o1, o2, o3 = A * 16, A * 24, A * 24 + 8
        head = torch.zeros(o3 + 4 * S_ + 4, dtype=torch.uint8, device=dev)
        return PhantomBatch(n=head[o2:o2 + 4].view(torch.int32), cell=i(A), pos0=head[:o1].view(torch.float64).view(A, 2),
                            yaw0=head[o1:o2].view(torch.float64), pos=f(S_, T, 2), yaw=f(S_, T), v=f(S_, T),
                            cov=f(S_, T, 2, 2), shape=f(S_, 2), raw_dims=f(S_, 2),
                            type=head[o3:o3 + 4 * S_].view(torch.int32), len=i(S_), R=self.R, head=head)

    def max_distance(self, ego_v):
        if self.max_dist_override is not None:
            return float(self.max_dist_override)
        return max(S_THRESHOLD_TIME * float(ego_v), S_THRESHOLD_MIN)

    def sample(self, ego_pos, ego_orientation, ego_v) -> PhantomBatch:
        """device-only path: frontier candidates -> evenly spaced pick -> headings -> predictions (no host sync)"""
        sm = self.sensor_model
        if sm.cell_class is None:
            raise RuntimeError("SpawnLocator: call SensorModel.calc_visible_and_occluded_area first")
        if self.batch is None:
            self.batch = self._alloc()
        b, w = self.batch, sm.window
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
    SOURCE_NAME = {1: "behind_dynamic_obstacle", 2: "behind static obstacle", 3: "left turn", 4: "right turn"}
    MAX_RULE_POINTS = 16

    def _rule_setup(self):
        """one-off: the polyline frame of the reference path as the table fo_scene_spawn_rules reads, the output buffers
        (the lanelet topology belongs to the static map: SensorModel._set_topology)"""
        from .scenario import lanelets_of
        from .utils.curvilinear import PolylineCS
        sm = self.sensor_model
        self._cs = PolylineCS(self.ref_path)
        n = len(self.ref_path)
        tab = np.zeros((n, 6))
        tab[:, :2], tab[:, 2] = self._cs.path, self._cs.s
        tab[:-1, 3], tab[:-1, 4:6] = self._cs.seg_len, self._cs.tangent
        self._d_path6 = torch.as_tensor(tab).to(self.device)
        try:
            lanelets = lanelets_of(sm.lanelet_network)
        except Exception:
            lanelets = []
        self._rule_lanelets = lanelets
        P = len(lanelets)
        self._rule_out = torch.zeros(self.MAX_RULE_POINTS * 8 + 1, dtype=torch.float64, device=self.device)
        self._rule_n = self._rule_out[-1:].view(torch.int32)[:1]
        sl = self.config["spawn_locator"]
        ped = self.config["agent_manager"]["pedestrian"]
        self._rule_cfg = dict(behind_static=int(bool(sl.get("spawn_point_behind_static_obstacle", True))),
                              behind_turn=int(bool(sl.get("spawn_points_behind_turn", True))),
                              behind_dynamic=int(bool(sl.get("spawn_point_behind_dynamic_obstacle", True)) and P > 0),
                              max_static=int(sl.get("max_static_spawn_points", 1)),
                              max_dynamic=int(sl.get("max_dynamic_spawn_points", 1)),
                              ped_width=float(ped["width"]), ped_length=float(ped["length"]))
        self._rules_ready = True

    def _rule_points_device(self, ego_pos, ego_orientation, ego_pos_cl, ego_v):
        """``spawn.mode: rules``: the three rule families of the reference evaluated by fo_scene_spawn_rules on the cell
        classes of this step, in HBM; the host supplies the step's scalars (curvilinear ego position, s_threshold,
        the ego's intention from the curvature of the next 40 m of the reference path, spawn_locator.py:113,678-741)
        and reads the handful of spawn points back."""
        import ctypes as C
        from .utils.curvilinear import curvature
        if not getattr(self, "_rules_ready", False):
            self._rule_setup()
        sm = self.sensor_model
        if ego_pos_cl is None:
            ego_pos_cl = self._cs.convert_to_curvilinear_coords(float(ego_pos[0]), float(ego_pos[1]))
        s_ego = float(ego_pos_cl[0])
        ref_s = self._cs.s
        i0 = int(np.argmin(np.abs(ref_s - s_ego)))                       # :678-693
        i1 = int(np.argmin(np.abs(ref_s - (s_ego + 40.0))))
        intention = 0
        if i1 - i0 >= 3:                                                  # :729-741
            key = (i0, i1)
            if getattr(self, "_intent_key", None) != key:
                k = curvature(self.ref_path[i0:i1])
                self._intent = 1 if k.max() > 0.10 else 2 if k.min() < -0.10 else 0
                self._intent_key = key
            intention = self._intent
        self.last_intention = ("straight ahead", "left turn", "right turn")[intention]
        # obstacle attributes the rules read beyond the corner points (one small upload per step)
        obst = list(self.fo_obstacles) if self.fo_obstacles is not None else []
        O = len(obst)
        d_corn, d_cen, d_flags, O_dev = getattr(sm, "_obst", (None, None, None, 0))
        if O != O_dev:
            raise RuntimeError("SpawnLocator: the obstacles of this step have not been uploaded (SensorModel.upload_obstacles)")
        p = lambda t: t.data_ptr() if t is not None else None
        if O:
            host = np.zeros(O * 25, dtype=np.uint8)
            yd = host[:O * 24].view(np.float64).reshape(O, 3)
            fl = host[O * 24:]
            for i, o in enumerate(obst):
                if o.current_pos is None:
                    continue
                yd[i] = (o.current_orientation, o.length, o.width)
                t = str(o.obstacle_type).lower()
                fl[i] = 1 | (2 if o.occludes else 0) | (4 if o.obstacle_role == "dynamic" else 0) | (8 if t in ("bicycle", "pedestrian") else 0)
            # (headings, dimensions and flags -- not the positions -- so the block rarely changes from step to step:
            # uploaded only when it does)
            key = host.tobytes()
            if getattr(self, "_rule_obst_key", None) != key:
                d = torch.as_tensor(host).to(self.device)
                ydev = d[:O * 24].view(torch.float64).view(O, 3)
                self._rule_obst_dev = (ydev[:, 0].contiguous(), ydev[:, 1:3].contiguous(), d[O * 24:])
                self._rule_obst_key = key
            d_yaw, d_dims, d_fl = self._rule_obst_dev
            d_vis = sm._buf["ovis"]
        else:
            d_yaw = d_dims = d_fl = d_vis = None
        pr = N.SpawnRuleParams(float(ego_pos[0]), float(ego_pos[1]), float(ego_orientation), s_ego, float(ego_pos_cl[1]),
                               s_ego + max(float(ego_v) * S_THRESHOLD_TIME, S_THRESHOLD_MIN),
                               self._rule_cfg["ped_width"], self._rule_cfg["ped_length"], intention, i0, i1,
                               self._rule_cfg["behind_static"], self._rule_cfg["behind_turn"], self._rule_cfg["behind_dynamic"],
                               self._rule_cfg["max_static"], self._rule_cfg["max_dynamic"])
        w = sm.window
        self.ctx.call("fo_scene_spawn_rules", sm.cell_class.data_ptr(), w.ix0, w.iy0, w.nx, w.ny, int(self._d_path6.shape[0]),
                      self._d_path6.data_ptr(), O, p(d_corn), p(d_cen), p(d_yaw), p(d_dims), p(d_fl), p(d_vis),
                      C.byref(pr), self.MAX_RULE_POINTS, self._rule_out.data_ptr(), self._rule_n.data_ptr(),
                      N.current_stream(self._dev_index))
        # the one read-back of the rule path: into a pinned buffer, then wait for the stream
        if getattr(self, "_rule_host", None) is None:
            self._rule_host = torch.empty(self._rule_out.shape, dtype=self._rule_out.dtype).pin_memory()
        self._rule_host.copy_(self._rule_out, non_blocking=True)
        torch.cuda.current_stream(self._dev_index).synchronize()
        h = self._rule_host.numpy()
        n = int(h[-1:].view(np.int32)[0])
        pts = []
        for q in h[:n * 8].reshape(n, 8):
            src = self.SOURCE_NAME[int(q[6])]
            if int(q[6]) == 2:
                src += " " + str(obst[int(q[7])].obstacle_id)             # spawn_locator.py:462
            cl = None if np.isnan(q[4]) else np.array([q[4], q[5]])
            pts.append(SpawnPoint(np.array([q[1], q[2]]), TYPE_NAME[int(q[0])], cl, src, None if np.isnan(q[3]) else float(q[3])))
        return pts

    def find_spawn_points(self, ego_pos, ego_orientation, ego_pos_cl, ego_v):
        """reference signature (spawn_locator.py:80): returns list[SpawnPoint].  Cell-sampled points come from the
        device (one small D2H copy); rule-based points (mode 'rules' / 'both') come from ``fo_scene_spawn_rules`` -- the
        reference's three rule families on the device -- and are turned into agents by the caller through
        ``FOAgentManager.add_agent``."""
        self.spawn_points, self.rule_points = [], []
        if self.mode in ("cells", "both"):
            b = self.sample(ego_pos, ego_orientation, ego_v)
            n, pos0, yaw0, typ_slots = b.host_head()
            typ = typ_slots[:: b.R]                # agent j's type = type of its first prediction slot
            for j in range(n):
                cl = None
                if self.cosy_cl is not None:
                    try:
                        cl = np.asarray(self.cosy_cl.convert_to_curvilinear_coords(pos0[j, 0], pos0[j, 1]))
                    except Exception:   # out of the projection domain: the reference skips silently (:228-231)
                        cl = None
                self.spawn_points.append(SpawnPoint(pos0[j].copy(), TYPE_NAME[int(typ[j])], cl, "occluded frontier",
                                                    float(yaw0[j])))
        else:
            self.batch = None
        self.n_cell_points = len(self.spawn_points)
        if self.mode in ("rules", "both"):
            self.rule_points = self._rule_points_device(ego_pos, ego_orientation, ego_pos_cl, ego_v)
            self.spawn_points = self.spawn_points + self.rule_points
        return self.spawn_points
