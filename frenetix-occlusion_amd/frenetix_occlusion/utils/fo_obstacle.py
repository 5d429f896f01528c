"""Per-timestep obstacle state cache (mirrors ref: utils/fo_obstacle.py:14-116, helper_functions.py:99-112).

Wraps each scenario obstacle -- a :class:`~frenetix_occlusion.scenario.Obstacle` or a duck-typed CommonRoad obstacle --
and caches pose / corner points / visibility for the current global time step.  ``arrays()`` packs what
``fo_scene_visibility`` consumes.  Pure host bookkeeping (a handful of obstacles); no polygons are built.
"""
import numpy as np

from ..scenario import Obstacle, obstacle_from_commonroad


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

    def update(self, timestep):
        if self._pending is not None:     # the last step's visibility first: last_visible_at_ts outlives the step
            self._pending()
        for o in self.fo_obstacles:
            o.update_at_timestep(timestep)

    def update_multipolygon(self):
        self._visible_obstacle_multipolygon = [o.current_corner_points for o in self.fo_obstacles if o._current_visible]

    def packed(self):
        """the rows ``SensorModel.stage_obstacles`` hands to a one-call step, one host buffer in the layout of
        ``SensorModel.upload_obstacles``: corners [O,4,2] | centres [O,2] | headings [O] | dimensions [O,2] | flags [O]"""
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
