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
        self.mode = str(acc.get("mode", "cells"))                  # "cells" | "rules" | "both" (spawn_rules.py)
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

    def _rule_engine(self):
        if self._rules is None:
            from .scenario import lanelets_of, points_in_polygon
            from .spawn_rules import SpawnRules
            from .utils.curvilinear import PolylineCS
            sm = self.sensor_model
            if self.cosy_cl is None:
                self.cosy_cl = PolylineCS(self.ref_path)     # interface.py docstring: "initialized if not provided"
            try:
                lanelets = lanelets_of(sm.lanelet_network)
            except Exception:
                lanelets = []

            def lane_yaw_at(xy):
                if sm.lane_yaw is None:
                    return None
                (x0, y0), (nx, ny) = sm.raster_origin, sm.raster_dims
                ix, iy = int(math.floor((xy[0] - x0) / sm.cell_size)), int(math.floor((xy[1] - y0) / sm.cell_size))
                if not (0 <= ix < nx and 0 <= iy < ny) or np.isnan(sm.lane_yaw[iy, ix]):
                    return None
                return float(sm.lane_yaw[iy, ix])

            def lanelet_of(xy):
                q = np.asarray(xy, dtype=np.float64).reshape(1, 2)
                for ll in lanelets:
                    if points_in_polygon(q, ll.polygon)[0]:
                        return ll
                return None
            inters = getattr(getattr(self.agent_manager, "scenario", None), "intersections", None) or []
            self._rules = SpawnRules(self.config, self.ref_path, self.cosy_cl, lane_yaw_at, lanelet_of, self.fo_obstacles,
                                     self.debug, lanelets=lanelets, intersections=inters)
        self._rules.cosy_cl = self.cosy_cl or self._rules.cosy_cl
        return self._rules

    def find_spawn_points(self, ego_pos, ego_orientation, ego_pos_cl, ego_v):
        """reference signature (spawn_locator.py:80): returns list[SpawnPoint].  Cell-sampled points come from the
        device (one small D2H copy); rule-based points (mode 'rules' / 'both') are evaluated on a host copy of the
        cell classes and are turned into agents by the caller through ``FOAgentManager.add_agent``."""
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
            from .spawn_rules import CellView
            rules = self._rule_engine()
            sm = self.sensor_model
            view = CellView(sm.cell_class.cpu().numpy(), sm.window)
            if ego_pos_cl is None:
                ego_pos_cl = rules.cosy_cl.convert_to_curvilinear_coords(float(ego_pos[0]), float(ego_pos[1]))
            self.rule_points = rules.find(view, ego_pos, ego_pos_cl, ego_v, ego_orientation)
            self.spawn_points = self.spawn_points + self.rule_points
        return self.spawn_points
