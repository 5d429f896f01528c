"""Per-timestep obstacle state cache (mirrors ref: utils/fo_obstacle.py:14-116, helper_functions.py:99-112).

Wraps each scenario obstacle -- a :class:`~frenetix_occlusion.scenario.Obstacle` or a duck-typed CommonRoad obstacle --
and caches pose / corner points / visibility for the current global time step.  ``arrays()`` packs what
``fo_scene_visibility`` consumes.  Pure host bookkeeping (a handful of obstacles); no polygons are built.
"""
import math

import numpy as np

from ..scenario import Obstacle, obstacle_from_commonroad, _CVX, _CVY


def _pyhost():
    from .. import _native
    return _native.pyhost()


class FOObstacle:
    def __init__(self, obst):
        self.cr_obstacle = obst
        self._o = obst if isinstance(obst, Obstacle) else obstacle_from_commonroad(obst)
        self.obstacle_id = self._o.obstacle_id
        self.obstacle_type = self._o.obstacle_type
        self.obstacle_role = self._o.role
        self.initial_timestep = self._o.initial_time_step
        self.global_timestep = None
        self.relative_time_step = None
        self.current_pos = None
        self.current_orientation = None
        self.current_corner_points = None
        self._owner = None               # the FOObstacles this one belongs to (pending visibility of a one-call step)
        self._current_visible = False
        self._last_visible_at_ts = None

    # visibility of the current step: after a one-call planning step these are read from the step's mirror when somebody
    # looks (SensorModel.defer_visible_objects)
    @property
    def current_visible(self):
        o = self._owner
        if o is not None and o._pending is not None:
            o._pending()
        return self._current_visible

    @current_visible.setter
    def current_visible(self, v):
        self._current_visible = v

    @property
    def last_visible_at_ts(self):
        o = self._owner
        if o is not None and o._pending is not None:
            o._pending()
        return self._last_visible_at_ts

    @last_visible_at_ts.setter
    def last_visible_at_ts(self, v):
        self._last_visible_at_ts = v

    @property
    def length(self):
        return self._o.length

    @property
    def width(self):
        return self._o.width

    def update_at_timestep(self, timestep):
        """fo_obstacle.py:79-116: rel = step - initial; 0 -> initial state, >= 1 -> state_list[rel-1], else absent"""
        if self._owner is not None:
            # the last step's visibility first (as FOObstacles.update does): a deferred one-call step must stamp
            # last_visible_at_ts with ITS time step before this obstacle moves on, not overwrite the moved state later
            if self._owner._pending is not None:
                self._owner._pending()
            self._owner._packed = None       # (a single obstacle moved by hand: the owner's packed rows are stale)
        self.global_timestep = timestep
        self.relative_time_step = timestep - self.initial_timestep
        self._current_visible = False
        pose = self._o.pose_at(timestep)
        if pose is None:
            self.current_pos = self.current_orientation = self.current_corner_points = None
            return
        self.current_pos = np.array(pose[:2], dtype=np.float64)
        self.current_orientation = float(pose[2])
        self.current_corner_points = self._o.corners(pose)

    @property
    def occludes(self):
        return str(self.obstacle_type).lower() != "bicycle"  # sensor_model.py:177 (Q10)


class FOObstacles:
    def __init__(self, cr_obstacles):
        self.cr_obstacles = list(cr_obstacles)
        self.fo_obstacles = [FOObstacle(o) for o in self.cr_obstacles]
        self._pending = None                        # SensorModel.resolve_visible_objects while a step's visibility is unread
        self._table = None                          # per-obstacle state table for update() (built on first use)
        self._packed = None                         # the current step's rows in the layout of packed()
        for o in self.fo_obstacles:
            o._owner = self
        self._visible_obstacle_multipolygon = None  # list of [4,2] corner arrays of the visible obstacles

    @property
    def visible_obstacle_multipolygon(self):
        if self._pending is not None:
            self._pending()
        return self._visible_obstacle_multipolygon

    @visible_obstacle_multipolygon.setter
    def visible_obstacle_multipolygon(self, v):
        self._visible_obstacle_multipolygon = v

    def __iter__(self):
        return iter(self.fo_obstacles)

    def __len__(self):
        return len(self.fo_obstacles)

    def add(self, cr_obstacle):
        self.cr_obstacles.append(cr_obstacle)
        self.fo_obstacles.append(FOObstacle(cr_obstacle))
        self.fo_obstacles[-1]._owner = self
        self._table = self._packed = None

    def update(self, timestep):
        if self._pending is not None:     # the last step's visibility first: last_visible_at_ts outlives the step
            self._pending()
        O = len(self.fo_obstacles)
        if O == 0:
            return
        # All obstacles of the step in a handful of array expressions (the per-obstacle form -- FOObstacle.update_at_timestep,
        # kept for callers that move one obstacle -- costs ~10 us each): poses from a table [O, steps, 3] made once, corner
        # points by the formula of scenario.Obstacle.corners, written straight into the rows a one-call step uploads.
        H = _pyhost()
        if H is None and O < 4:          # (without the C helper the array form only pays from a handful of obstacles on)
            for o in self.fo_obstacles:
                o.update_at_timestep(timestep)
            return
        tb = self._table
        if tb is None:
            tb = self._table = self._build_table()
        host = np.empty(O * 105, dtype=np.uint8)                       # fresh per step: last step's views stay what they were
        corn = host[:O * 64].view(np.float64).reshape(O, 4, 2)
        cen = host[O * 64:O * 80].view(np.float64).reshape(O, 2)
        yaw = host[O * 80:O * 88].view(np.float64)
        dims = host[O * 88:O * 104].view(np.float64).reshape(O, 2)
        flags = host[O * 104:]
        if H is not None:
            # csrc/fo_pyhost.c obstacle_rows: the same operations in the same order, in C (the array expressions below cost
            # ~40 us of interpreter time whatever O is)
            present = np.empty(O, dtype=np.uint8)
            H.obstacle_rows(tb["P"], tb["L"], tb["t0"], tb["static_u8"], tb["vx"], tb["vy"], tb["dims"], tb["flags"], int(timestep),
                            host, present)
            ya = yaw
        else:
            rel = np.where(tb["static"], 0, timestep - tb["t0"])
            present = (rel >= 0) & (rel < tb["L"])
            pose = tb["P"][tb["ar"], np.clip(rel, 0, tb["L"] - 1)]          # [O, 3]
            ya = pose[:, 2]
            c = np.array([math.cos(v) for v in ya])[:, None]                # (math.cos, as Obstacle.corners: numpy's differs in the last bit)
            s = np.array([math.sin(v) for v in ya])[:, None]
            corn[:, :, 0] = pose[:, 0:1] + (c * tb["vx"] - s * tb["vy"])
            corn[:, :, 1] = pose[:, 1:2] + (s * tb["vx"] + c * tb["vy"])
            cen[:] = pose[:, :2]
            yaw[:] = ya
            dims[:] = tb["dims"]
            flags[:] = tb["flags"]
            if not present.all():
                gone = ~present
                corn[gone] = 0.0; cen[gone] = 0.0; yaw[gone] = 0.0; dims[gone] = 0.0; flags[gone] = 0
        t0 = tb["t0_list"]
        for i, o in enumerate(self.fo_obstacles):
            o.global_timestep = timestep
            o.relative_time_step = timestep - t0[i]
            o._current_visible = False
            if present[i]:
                o.current_pos, o.current_orientation, o.current_corner_points = cen[i], float(ya[i]), corn[i]
            else:
                o.current_pos = o.current_orientation = o.current_corner_points = None
        self._packed = (host, corn, cen, flags, yaw, dims)

    def _build_table(self):
        obs = [o._o for o in self.fo_obstacles]
        O = len(obs)
        static = np.array([o.role == "static" for o in obs])
        L = np.array([1 if st else 1 + len(o.states) for o, st in zip(obs, static)])
        P = np.zeros((O, int(L.max()), 3))
        for i, o in enumerate(obs):
            P[i, 0] = o.initial[:3]
            if L[i] > 1:
                P[i, 1:L[i]] = np.asarray(o.states)[:, :3]
        l2 = np.array([o.length / 2.0 for o in obs])[:, None]
        w2 = np.array([o.width / 2.0 for o in obs])[:, None]
        flags = np.array([1 | (2 if f.occludes else 0) | (4 if f.obstacle_role == "dynamic" else 0)
                          | (8 if str(f.obstacle_type).lower() in ("bicycle", "pedestrian") else 0) for f in self.fo_obstacles],
                         dtype=np.uint8)
        return dict(static=static, static_u8=static.astype(np.uint8), t0=np.array([o.initial_time_step for o in obs], dtype=np.int64),
                    t0_list=[f.initial_timestep for f in self.fo_obstacles], L=L.astype(np.int64), P=np.ascontiguousarray(P),
                    ar=np.arange(O), vx=np.ascontiguousarray(_CVX[None, :] * l2), vy=np.ascontiguousarray(_CVY[None, :] * w2),
                    dims=np.array([(float(f.length), float(f.width)) for f in self.fo_obstacles]).reshape(O, 2), flags=flags)

    def update_multipolygon(self):
        self._visible_obstacle_multipolygon = [o.current_corner_points for o in self.fo_obstacles if o._current_visible]

    def packed(self):
        """the rows ``SensorModel.stage_obstacles`` hands to a one-call step, one host buffer in the layout of
        ``SensorModel.upload_obstacles``: corners [O,4,2] | centres [O,2] | headings [O] | dimensions [O,2] | flags [O]"""
        if self._packed is not None:
            return self._packed[0]
        corn, cen, flags, yaw, dims = self.arrays_full()
        O = len(flags)
        host = np.empty(O * 105, dtype=np.uint8)
        host[:O * 64].view(np.float64)[:] = corn.reshape(-1)
        host[O * 64:O * 80].view(np.float64)[:] = cen.reshape(-1)
        host[O * 80:O * 88].view(np.float64)[:] = yaw
        host[O * 88:O * 104].view(np.float64)[:] = dims.reshape(-1)
        host[O * 104:] = flags
        return host

    def arrays(self):
        """corner points [O,4,2], centres [O,2], flags uint8 [O] (bit0 present at this step, bit1 occludes)"""
        if self._packed is not None:
            _, corn, cen, flags, _, _ = self._packed
            return corn, cen, flags & 3
        O = len(self.fo_obstacles)
        corn, cen, flags = np.zeros((O, 4, 2)), np.zeros((O, 2)), np.zeros(O, dtype=np.uint8)
        for i, o in enumerate(self.fo_obstacles):
            if o.current_pos is None:
                continue
            corn[i], cen[i] = o.current_corner_points, o.current_pos
            flags[i] = 1 | (2 if o.occludes else 0)
        return corn, cen, flags

    def arrays_full(self):
        """``arrays()`` plus what the spawn rule families read (fo_scene_spawn_rules): headings [O], dimensions [O,2]
        (length, width), and two more flag bits -- bit2 dynamic role, bit3 type bicycle or pedestrian
        (spawn_locator.py:209-210).  The visibility kernels test bits 0 and 1 only, so one flag array serves both."""
        if self._packed is not None:
            _, corn, cen, flags, yaw, dims = self._packed
            return corn, cen, flags, yaw, dims
        corn, cen, flags = self.arrays()
        O = len(self.fo_obstacles)
        yaw, dims = np.zeros(O), np.zeros((O, 2))
        for i, o in enumerate(self.fo_obstacles):
            if o.current_pos is None:
                continue
            yaw[i], dims[i] = o.current_orientation, (o.length, o.width)
            t = str(o.obstacle_type).lower()
            flags[i] |= (4 if o.obstacle_role == "dynamic" else 0) | (8 if t in ("bicycle", "pedestrian") else 0)
        return corn, cen, flags, yaw, dims
