"""Phantom-agent registry (mirrors ref: agent.py:28-199 ``FOAgentManager``).

Holds the phantom agents of the current step and their predictions.  The predictions live on the GPU in the
structure-of-arrays form the metric sweep consumes; ``predictions`` (the dict the reference exposes,
agent.py:179-183) is materialised lazily for callers that want the reference's view.  Manually added agents
(``add_agent``, scripted in the YAML ``agents:`` list) get constant-velocity straight-line predictions
(agent.py:451-536); lane-following Frenet predictions for phantom vehicles need the un-vendored frenetix sampler
and are out of scope -- vehicles are predicted along their lane heading at constant speed.
"""
from random import randint

import numpy as np
import torch

from .spawn_locator import TYPE_CODE, TYPE_NAME, PhantomBatch


class PhantomAgent:
    def __init__(self, agent_id, agent_type, position, orientation, velocity, length, width):
        self.agent_id, self.agent_type = agent_id, agent_type
        self.initial_position, self.initial_orientation, self.initial_velocity = position, orientation, velocity
        self.length, self.width = length, width
        self.predictions = None


class FOAgentManager:
    def __init__(self, scenario, reference_path, config, timestep, visualization=None, dt=0.1, fo_obstacles=None,
                 debug=False, device=None):
        self.scenario = scenario
        self.timestep = timestep
        self.reference_path = np.asarray(reference_path, dtype=np.float64)
        self.config = config
        self.visualization = visualization
        self.fo_obstacles = fo_obstacles
        self.dt = dt
        self.debug = debug
        self.device = device if device is not None else torch.device("cuda", 0)
        self.real_agents = []
        self._batch = None            # PhantomBatch from the spawn kernels (device)
        self._live = None             # agent indices of the batch that are alive this step (None: not read back yet)
        self._batch_agents = None     # PhantomAgent objects of those slots (built on first access)
        self._manual = []             # agents added through add_agent() (host-side predictions)
        self._external = []           # (obstacle id, prediction dict, type name) of REAL agents (extension, see below)
        self._pred_cache = None
        self.all_obstacle_id = [getattr(o, "obstacle_id", None) for o in getattr(scenario, "obstacles", [])]
        self._step_ids = []           # ids minted for this step's phantoms; handed back by reset()

    # ---- reference API ------------------------------------------------------------------------------------
    def reset(self):
        self._batch, self._live, self._batch_agents = None, None, None
        self._manual = []
        self._external = []
        self._pred_cache = None
        self._release_step_ids()

    def _release_step_ids(self):
        """hand back the ids of phantoms that no longer exist; ids of the scenario's obstacles and of agents added
        to the scenario (``real_agents``) stay taken.  The reference never releases (agent.py:189-199 only appends),
        which is harmless at its ~5 phantoms per step but would drain the 1001-value range in 4 steps at 256."""
        if self._step_ids:
            keep = {a.agent_id for a in self.real_agents}
            drop = {i for i in self._step_ids if i not in keep}
            self.all_obstacle_id = [i for i in self.all_obstacle_id if i not in drop]
            self._step_ids = []

    ID_RANGE = (10000, 11000)        # agent.py:191
    ID_RANGE_WIDE = (11001, 99999)   # still five digits, so agent_by_prediction_id's str(pid)[:5] keeps working

    def _create_id(self):
        """agent.py:189-199: unique random id in [10000, 11000].  Bounded: a few random draws, then the first free
        value of the range, then the wider five-digit range; raises when every five-digit id is taken instead of
        spinning."""
        taken = set(self.all_obstacle_id)
        lo, hi = self.ID_RANGE
        i = None
        for _ in range(8):
            c = randint(lo, hi)
            if c not in taken:
                i = c
                break
        if i is None:
            for rng in (self.ID_RANGE, self.ID_RANGE_WIDE):
                i = next((c for c in range(rng[0], rng[1] + 1) if c not in taken), None)
                if i is not None:
                    break
        if i is None:
            raise RuntimeError("FOAgentManager: no free five-digit agent id left")
        self.all_obstacle_id.append(i)
        self._step_ids.append(i)
        return i

    def _velocity(self, velocity, conf, allow_lanelet):
        if velocity == "default":
            return float(conf["default_velocity"])
        if velocity == "lanelet":
            if not allow_lanelet:
                raise NotImplementedError
            return 30 / 3.6                                     # agent.py:326-328 (urban); other lanelet types [ext]
        if not isinstance(velocity, (int, float)):
            raise ValueError('Only "default", "lanelet", int or float is allowed!')   # agent.py:86,100,114,131
        return float(velocity)

    def add_agent(self, pos, velocity="default", agent_type="Car", add_to_scenario=False, timestep=0, horizon=3.0,
                  mode="ref_path", orientation=None):
        if self.timestep != timestep:                            # agent.py:69-70
            return None
        key = agent_type.lower()
        if key not in ("bicycle", "car", "truck", "pedestrian"):
            raise NotImplementedError(f'OAPManager: Agent type "{agent_type}" is not implemented!')
        if mode not in ("ref_path", "lane_center"):
            raise NotImplementedError(f'Selected mode "{mode}" is not implemented: use "ref_path" or "lane_center"!')
        conf = self.config[key]
        v = self._velocity(velocity, conf, key in ("car", "truck"))
        pos = np.asarray(pos, dtype=np.float64)
        routes = self._routes_at(pos) if (key != "pedestrian" and orientation is None) else []
        if routes:                                               # vehicles follow their lanelet (agent.py:283-312)
            seg = routes[0][1:2] - routes[0][0:1]
            orientation = float(np.arctan2(seg[0, 1], seg[0, 0]))
        elif orientation is None:
            curve = self.reference_path
            if mode == "lane_center":                            # agent.py:459-467; off-lanelet -> reference path (Q12)
                curve = self._lane_center_at(pos)
            orientation = self._heading_towards_path(pos, curve)
        agent = PhantomAgent(self._create_id(), agent_type, pos, float(orientation), v, float(conf["length"]),
                             float(conf["width"]))
        agent.predictions = [self._route_prediction(agent, horizon, r) for r in routes] or [self._cv_prediction(agent, horizon)]
        agent.predictions = [p for p in agent.predictions if p is not None] or [self._cv_prediction(agent, horizon)]
        if add_to_scenario:
            self.real_agents.append(agent)
            from .scenario import Obstacle
            p0 = agent.predictions[0]
            st = np.column_stack((p0["pos_list"][1:], p0["orientation_list"][1:], p0["v_list"][1:]))
            ob = Obstacle(agent.agent_id, "dynamic", key, agent.length, agent.width, int(timestep),
                          np.array([pos[0], pos[1], agent.initial_orientation, v]), st)
            if hasattr(self.scenario, "add_objects"):
                self.scenario.add_objects(ob)
            if self.fo_obstacles is not None:
                self.fo_obstacles.add(ob)
        else:
            self._manual.append(agent)
            self._pred_cache = None
        return agent

    def _routes_at(self, pos, R=3):
        """candidate route polylines of the lanelet containing pos (route_planner.py:31-90), cached per scenario"""
        from .scenario import enumerate_routes, lanelets_of, points_in_polygon, route_polyline
        try:
            lanelets = lanelets_of(self.scenario.lanelet_network if hasattr(self.scenario, "lanelet_network") else self.scenario)
        except Exception:
            return []
        if getattr(self, "_route_cache", None) is None:
            self._route_cache = (enumerate_routes(lanelets), {ll.lanelet_id: ll for ll in lanelets})
        routes, by = self._route_cache
        q = np.asarray(pos, dtype=np.float64).reshape(1, 2)
        for ll in lanelets:
            if points_in_polygon(q, ll.polygon)[0]:
                polys, seen = [], set()
                for rt in routes.get(ll.lanelet_id, []):
                    pl = route_polyline(by, rt)
                    if len(pl) >= 2 and pl.tobytes() not in seen:
                        seen.add(pl.tobytes())
                        polys.append(pl)
                return polys[:R]
        return []

    def _route_prediction(self, agent, horizon, poly):
        """the reference's min-var(v) Frenet sample along a route polyline: speed held, quintic lateral move to the nearest
        of d1 in {-0.5, 0, 0.5} (same construction as fo_spawn_predict_kernel; replaces the frenetix sampler of
        agent.py:283-426, frenetix_handler.py:82-105)"""
        pr = self.config["prediction"]
        big = agent.agent_type.lower() == "bicycle"
        fl = pr["size_factor_length_l"] if big else pr["size_factor_length_s"]
        fw = pr["size_factor_width_l"] if big else pr["size_factor_width_s"]
        p = agent.initial_position
        a, e = poly[:-1], poly[1:] - poly[:-1]
        l = np.hypot(e[:, 0], e[:, 1])
        t = np.clip(np.sum((p[None] - a) * e, axis=1) / (l * l), 0.0, 1.0)
        foot = a + t[:, None] * e
        k = int(np.argmin(np.sum((p[None] - foot) ** 2, axis=1)))
        sarr = np.concatenate(([0.0], np.cumsum(l)))
        s0 = sarr[k] + t[k] * l[k]
        d0 = float(((p - foot[k])[0] * (-e[k, 1]) + (p - foot[k])[1] * e[k, 0]) / l[k])
        T = int(horizon / self.dt) + 1
        sk = s0 + agent.initial_velocity * (np.arange(T) * self.dt)
        sk = sk[sk <= sarr[-1]]
        if len(sk) == 0:
            return None
        m = np.clip(np.searchsorted(sarr, sk, side="right") - 1, 0, len(l) - 1)
        u = e[m] / l[m][:, None]
        # the Frenet sample the reference keeps (agent.py:349-379; utils/frenet_sampling.py): speed held, quintic to d1
        from .utils.frenet_sampling import lateral_profile, nearest_lateral_target
        tk = np.arange(len(sk)) * self.dt
        dk, dd = lateral_profile(d0, nearest_lateral_target(d0), tk)
        pos = a[m] + (sk - sarr[m])[:, None] * u + dk[:, None] * np.stack((-u[:, 1], u[:, 0]), -1)
        L = len(sk)
        var = 0.1 * np.power(pr["variance_factor"], np.arange(L))
        cov = np.zeros((L, 2, 2))
        cov[:, 0, 0] = var
        cov[:, 1, 1] = var
        v0 = agent.initial_velocity
        return {"orientation_list": np.arctan2(u[:, 1], u[:, 0]) + np.arctan2(dd, v0), "v_list": np.sqrt(v0 * v0 + dd * dd),
                "pos_list": pos, "shape": {"length": agent.length * fl, "width": agent.width * fw}, "cov_list": cov}

    def _lane_center_at(self, pos):
        from .scenario import lanelets_of, points_in_polygon
        try:
            lanelets = lanelets_of(self.scenario.lanelet_network if hasattr(self.scenario, "lanelet_network") else self.scenario)
        except Exception:
            return self.reference_path
        q = np.asarray(pos, dtype=np.float64).reshape(1, 2)
        for ll in lanelets:
            if points_in_polygon(q, ll.polygon)[0]:
                return ll.center
        return self.reference_path

    def _heading_towards_path(self, pos, curve=None):
        """agent.py:475-481 + helper_functions.py:38-76: unit normal towards the curve, angle in [0, 2 pi)"""
        p = self.reference_path if curve is None else np.asarray(curve, dtype=np.float64)
        a, b = p[:-1], p[1:]
        e = b - a
        l2 = np.maximum(np.sum(e * e, axis=1), 1e-300)
        t = np.clip(np.sum((pos[None] - a) * e, axis=1) / l2, 0.0, 1.0)
        q = a + t[:, None] * e
        k = int(np.argmin(np.sum((q - pos[None]) ** 2, axis=1)))
        d = q[k] - pos
        n = float(np.hypot(d[0], d[1]))
        ang = float(np.arctan2(d[1], d[0])) if n > 0 else 0.0
        return ang + 2.0 * np.pi if ang < 0 else ang

    def _cv_prediction(self, agent, horizon):
        pr = self.config["prediction"]
        big = agent.agent_type.lower() == "bicycle"
        fl = pr["size_factor_length_l"] if big else pr["size_factor_length_s"]
        fw = pr["size_factor_width_l"] if big else pr["size_factor_width_s"]
        vx = round(agent.initial_velocity * np.cos(agent.initial_orientation), 3)     # agent.py:492-493 (Q12)
        vy = round(agent.initial_velocity * np.sin(agent.initial_orientation), 3)
        T = int(horizon / self.dt) + 1
        t = np.arange(T)[:, None] * self.dt
        pos = agent.initial_position[None] + t * np.array([[vx, vy]])
        var = 0.1 * np.power(pr["variance_factor"], np.arange(T))
        cov = np.zeros((T, 2, 2))
        cov[:, 0, 0] = var
        cov[:, 1, 1] = var
        return {"orientation_list": np.full(T, agent.initial_orientation), "v_list": np.full(T, agent.initial_velocity),
                "pos_list": pos, "shape": {"length": agent.length * fl, "width": agent.width * fw}, "cov_list": cov}

    def agent_by_prediction_id(self, prediction_id):
        agent_id = int(str(prediction_id)[:5])
        for a in self.phantom_agents:
            if a.agent_id == agent_id:
                return a
        return None

    def update_real_agents(self, cr_scenario_predictions):
        """agent.py:171-177: scripted real pedestrians overwrite their entry in the caller's prediction dict"""
        if cr_scenario_predictions is None:
            return
        for a in self.real_agents:
            if a.agent_type.lower() == "pedestrian" and a.agent_id in cr_scenario_predictions:
                cr_scenario_predictions[a.agent_id] = a.predictions[0]

    def set_external_predictions(self, predictions, types=None):
        """EXTENSION, not in the reference (whose metrics see the phantom agents only, interface.py:216-219): the
        predictions of REAL agents as the planner's prediction module delivers them -- ``{obstacle_id: {'pos_list'
        [L,2], 'v_list' [L], 'orientation_list' [L], 'cov_list' [L,2,2], 'shape': {'length', 'width'}}}`` -- take part
        in the sweep as further agents, keyed by their obstacle id in the results.  Their covariances may carry
        correlation.  ``types``: ``{obstacle_id: type name}``, default 'car'.  Call after reset()."""
        self._external = []
        for oid in sorted(predictions or {}):
            p = predictions[oid]
            L = min(len(p["pos_list"]), len(p["v_list"]), len(p["orientation_list"]), len(p["cov_list"]))
            if L > 0:
                name = str((types or {}).get(oid, "car")).lower()
                self._external.append((int(oid), p, name if name in TYPE_CODE else "unknown"))
        self._pred_cache = None

    # ---- device side ------------------------------------------------------------------------------------------
    def attach_batch(self, batch: PhantomBatch, n_active=None):
        """take over the spawn kernels' output.  Which of its agents are alive is read from the device when a host view
        (``phantom_agents``, ``predictions``, ``has_phantoms``) first needs it; ``n_active`` (the number of live
        cell-sampled agents, for a batch without rule agents) spares that copy when the caller knows it already."""
        if self._batch_agents:                       # ids of the batch being replaced
            gone = {a.agent_id for a in self._batch_agents}
            self.all_obstacle_id = [i for i in self.all_obstacle_id if i not in gone]
            self._step_ids = [i for i in self._step_ids if i not in gone]
        self._batch, self._batch_agents = batch, None
        self._live = list(range(int(n_active))) if (n_active is not None and not batch.n_rule_points) else None
        self._pred_cache = None

    def _live_agents(self):
        if self._batch is None:
            return []
        if self._live is None:
            self._live = self._batch.live_agents()
        return self._live

    @property
    def _n_batch(self):
        return len(self._live_agents())

    def may_have_phantoms(self):
        """True when a sweep has something to evaluate or MAY have: a device batch is attached (its live count stays in HBM
        -- asking for it would stall the step; a batch without live agents leaves every trajectory safe), or host-side
        agents exist"""
        if self._batch is not None and self._live is None:
            return True
        return self.has_phantoms()              # the live agents are known on the host already: the exact answer

    def has_phantoms(self):
        return self._n_batch > 0 or bool(self._manual) or bool(self._external)

    def n_slots(self):
        """length of the agent axis of the sweep outputs"""
        return (self._batch.pos.shape[0] if self._batch is not None else 0) + \
            sum(len(a.predictions) for a in self._manual) + len(self._external)

    @property
    def phantom_agents(self):
        """agent.py:39: list of phantom agents of this step (objects are created on first access)"""
        if self._batch_agents is None:
            self._batch_agents = []
            live = self._live_agents()
            if live:
                b = self._batch
                h = b.host_head()
                R = b.R
                idx = np.asarray(live, dtype=np.int64)
                if b.body is not None:           # one device-to-host copy of the step's predictions, shared with .predictions
                    hb = b.host_body()
                    raw, ln, v0 = hb["raw_dims"][idx * R], hb["len"].reshape(-1, R)[idx], hb["v"][:, 0].reshape(-1, R)[idx]
                else:
                    ti = torch.as_tensor(live, device=b.len.device)
                    raw = b.raw_dims[ti * R].cpu().numpy()
                    ln = b.len.view(-1, R)[ti].cpu().numpy()
                    v0 = b.v[:, 0].reshape(-1, R)[ti].cpu().numpy()
                for q, j in enumerate(live):
                    r0 = int(np.argmax(ln[q] > 0))     # first prediction that exists carries the speed
                    self._batch_agents.append(PhantomAgent(self._create_id(), TYPE_NAME[int(h["type"][j * R])], h["pos0"][j].copy(),
                                                           float(h["yaw0"][j]), float(v0[q, r0]), float(raw[q, 0]),
                                                           float(raw[q, 1])))
        return self._batch_agents + self._manual

    def sweep_arrays(self):
        """tensors for MetricSweep.set_agents: spawn-kernel slots first, then manually added agents, then the real
        agents' predictions (set_external_predictions)"""
        parts = []
        if self._batch is not None:
            parts.append(self._batch.sweep_args())
        if self._manual or self._external:
            dev = self.device
            # one sweep slot per prediction: (type name, un-inflated length, width, prediction)
            preds = [(a.agent_type, a.length, a.width, p) for a in self._manual for p in a.predictions]
            preds += [(t, p["shape"]["length"], p["shape"]["width"], p) for _, p, t in self._external]
            T = max(len(p["pos_list"]) for *_, p in preds)
            n = len(preds)
            pos, yaw, v = np.zeros((n, T, 2)), np.zeros((n, T)), np.zeros((n, T))
            cov, shape, raw = np.zeros((n, T, 2, 2)), np.zeros((n, 2)), np.zeros((n, 2))
            typ, ln = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
            for i, (tname, rl, rw, p) in enumerate(preds):
                L = min(len(p["pos_list"]), len(p["v_list"]), len(p["orientation_list"]), len(p["cov_list"]))
                pos[i, :L], yaw[i, :L], v[i, :L] = p["pos_list"][:L], p["orientation_list"][:L], p["v_list"][:L]
                cov[i, :L] = p["cov_list"][:L]
                shape[i] = (p["shape"]["length"], p["shape"]["width"])
                raw[i] = (rl, rw)
                typ[i], ln[i] = TYPE_CODE[tname.lower()], L
            d = lambda x, dt=torch.float64: torch.as_tensor(x).to(dev, dt)
            parts.append((d(pos), d(yaw), d(v), d(cov), d(shape), d(raw), d(typ, torch.int32), d(ln, torch.int32)))
        if not parts:
            return None
        if len(parts) == 1:
            return parts[0]
        T = max(p[0].shape[1] for p in parts)

        def pad_t(t):
            if t.shape[1] == T:
                return t
            pad = torch.zeros((t.shape[0], T - t.shape[1]) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            return torch.cat((t, pad), dim=1)
        out = []
        for k in range(8):
            ts = [p[k] for p in parts]
            if k < 4:
                ts = [pad_t(t) for t in ts]
            out.append(torch.cat(ts, dim=0).contiguous())
        return tuple(out)

    @property
    def predictions(self):
        """dict prediction_id -> {'pos_list','v_list','orientation_list','shape','cov_list'} (agent.py:179-183),
        key = int(str(agent_id) + '0'); ``prediction_slots`` maps each id to its index on the sweep's agent axis."""
        if self._pred_cache is not None:
            return self._pred_cache
        out, order = {}, []
        agents = self.phantom_agents
        live = self._live_agents()
        if live:
            b = self._batch
            R = b.R
            if b.body is not None:
                hb = b.host_body()
                sl = (np.asarray(live, dtype=np.int64)[:, None] * R + np.arange(R)[None]).reshape(-1)
                pos, yaw, v, cov, shape, ln = (hb[k_][sl] for k_ in ("pos", "yaw", "v", "cov", "shape", "len"))
            else:
                sl = (torch.as_tensor(live, device=b.len.device)[:, None] * R + torch.arange(R, device=b.len.device)[None]).reshape(-1)
                pos, yaw, v = b.pos[sl].cpu().numpy(), b.yaw[sl].cpu().numpy(), b.v[sl].cpu().numpy()
                cov, shape, ln = b.cov[sl].cpu().numpy(), b.shape[sl].cpu().numpy(), b.len[sl].cpu().numpy()
            for q, j in enumerate(live):
                a = agents[q]
                a.predictions = []
                for r in range(R):                      # one prediction per candidate route (agent.py:410-424)
                    k, slot, L = q * R + r, j * R + r, int(ln[q * R + r])
                    if L <= 0:
                        continue
                    pred = {"orientation_list": yaw[k, :L], "v_list": v[k, :L], "pos_list": pos[k, :L],
                            "shape": {"length": float(shape[k, 0]), "width": float(shape[k, 1])},
                            "cov_list": cov[k, :L]}
                    pid = int(str(a.agent_id) + str(len(a.predictions)))       # agent.py:179-183
                    a.predictions.append(pred)
                    out[pid] = pred
                    order.append((pid, slot))
        base = self._batch.pos.shape[0] if self._batch is not None else 0
        slot = base
        for a in self._manual:
            for i, p in enumerate(a.predictions):
                pid = int(str(a.agent_id) + str(i))                  # agent.py:179-183
                out[pid] = p
                order.append((pid, slot))
                slot += 1
        for oid, p, _ in self._external:                             # real agents: keyed by their obstacle id
            out[oid] = p
            order.append((oid, slot))
            slot += 1
        self._pred_cache = out
        self.prediction_slots = order
        return out
