"""``FOInterface`` -- the plugin surface the Frenetix-Motion-Planner loop talks to (ref: interface.py:18-238).

Same constructor, same ``evaluate_scenario`` / ``trajectory_safety_assessment`` / ``set_coordinate_system``
signatures and the same instance attributes as the reference, so the planner sees a drop-in; behind it every
per-step stage runs in libfo_hip.so on one MI355X:

    evaluate_scenario           obstacle poses -> ray fan + cell classes (fo_scene_visibility) -> phantom sampling and
                                predictions in the occluded cells (fo_scene_spawn) -> agent table (fo_sweep_set_agents)
    trajectory_safety_assessment_batch   all candidate trajectories x all phantom predictions in one launch
    trajectory_safety_assessment         the reference's per-trajectory call, served from the cached batch

Deviations from the reference (documented in DESIGN.md): ``visible_area`` is a ring polygon + cell mask
(:class:`~frenetix_occlusion.sensor_model.VisibleArea`) instead of a shapely geometry; spawn points come from the
reference's three rule families evaluated on the cell classes (``accelerator.spawn.mode: rules``, the default:
fo_scene_spawn_rules -> fo_scene_spawn_rule_agents, device resident -- the spawn points never leave HBM on their way into
the sweep, ``spawn_points`` is a list that reads them back when first looked at) and / or from the occluded-cell frontier
(``cells`` / ``both``: the sampler of the BASELINE configurations, not what the reference spawns); no matplotlib
(``plot`` is accepted and ignored); metric ``'be'`` implies ``'ttc'`` (the reference raises KeyError when ``'be'`` is
activated without it).
"""
import os
import time

import numpy as np
import torch
import yaml

from . import _native as N
from .agent import FOAgentManager
from .metrics.metric import Metric
from .sensor_model import SensorModel
from .spawn_locator import SpawnLocator
from .utils.fo_obstacle import FOObstacles


class FOInterface:
    def __init__(self, scenario, reference_path, vehicle_params, dt, config_path=None, cosy_cl=None, share_map_with=None):
        """Signature of the reference (interface.py:69) plus ``share_map_with``: another FOInterface of the same scenario
        on the same GPU (the planner of another ego) whose static map this one reads instead of building and uploading
        its own (BASELINE configs[4])."""
        self.config = self._load_config(config_path)
        acc = self.config.get("accelerator") or {}
        if not torch.cuda.is_available():
            raise RuntimeError("FOInterface needs a ROCm GPU: this build has no CPU path")
        self.device = torch.device("cuda", int(acc.get("device", 0)))

        # things that never change (interface.py:74-82)
        self.cr_scenario = scenario
        self.lanelet_network = scenario.lanelet_network
        self.ego_reference_path = np.asarray(reference_path, dtype=np.float64)
        self.cosy_cl = cosy_cl
        self.vehicle_params = vehicle_params
        self.dt = dt
        self.plot = self.config.get("plot", False)
        self.debug = self.config.get("debug", False)

        # per-step state (interface.py:85-90)
        self.predictions = None
        self.ego_pos = None
        self.ego_orientation = None
        self.ego_pos_cl = None
        self.timestep = None
        self.spawn_points = []

        self.sensor_radius = self.config["sensor_model"]["sensor_radius"]
        self.sensor_angle = self.config["sensor_model"]["sensor_angle"]
        self.visualization = None   # debug drawing is out of scope (forces TkAgg upstream)
        self.include_real_agents = bool(acc.get("include_real_agents", False))
        self._timing_sync = str(acc.get("timing", "issue")) == "device"
        self.step_timing = {}
        self._one_call = bool(acc.get("one_call", True))
        self._scene_step = None

        self.ctx = N.Context(self.device.index)
        self.fo_obstacles = FOObstacles(self.cr_scenario.obstacles)
        spawn_acc = acc.get("spawn") or {}
        routes = int(spawn_acc.get("routes", 0))
        if routes == 0 and str(spawn_acc.get("mode", "rules")) != "cells":
            routes = 3     # the rule families' vehicles follow the routes of their lanelet (agent.py:283-312)
        self.sensor_model = SensorModel(lanelet_network=self.lanelet_network, ref_path=self.ego_reference_path,
                                        sensor_radius=self.sensor_radius, sensor_angle=self.sensor_angle,
                                        visualization=None, debug=self.debug, ctx=self.ctx,
                                        n_rays=int(acc.get("rays", 720)), cell_size=float(acc.get("cell_size", 0.5)),
                                        device=self.device.index, routes=routes,
                                        footprint=str(acc.get("footprint", "polygon")),
                                        enclosed_holes=str(acc.get("enclosed_holes", "transparent")),
                                        cell_visibility=str(acc.get("cell_visibility", "exact")),
                                        shadow_length=float(acc.get("shadow_length", 100.0)),
                                        share_map_with=share_map_with.sensor_model if share_map_with is not None else None,
                                        intersections=getattr(self.cr_scenario, "intersections", None) or
                                        getattr(self.lanelet_network, "intersections", None))
        self.agent_manager = FOAgentManager(scenario=self.cr_scenario, reference_path=self.ego_reference_path,
                                            config=self.config["agent_manager"], visualization=None,
                                            timestep=self.timestep, dt=self.dt, debug=self.debug,
                                            fo_obstacles=self.fo_obstacles, device=self.device)
        self.spawn_locator = SpawnLocator(agent_manager=self.agent_manager, ref_path=self.ego_reference_path,
                                          config=self.config, cosy_cl=self.cosy_cl, sensor_model=self.sensor_model,
                                          fo_obstacles=self.fo_obstacles, visualization=None, debug=self.debug,
                                          dt=self.dt)
        self.metrics = Metric(self.config["metrics"], self.vehicle_params, self.agent_manager, dt=self.dt,
                              device=self.device.index, ctx=self.ctx, list_storage=str(acc.get("list_storage", "f64")))

    # ---------------------------------------------------------------------------------------- reference API
    def set_coordinate_system(self, cosy_cl):
        self.cosy_cl = cosy_cl
        self.spawn_locator.cosy_cl = cosy_cl

    def _add_real_agents(self):
        if self.config.get("agents") is None:
            return
        for agent in self.config["agents"]:
            self.agent_manager.add_agent(pos=agent["position"], velocity=agent["velocity"],
                                         agent_type=agent["agent_type"], add_to_scenario=True,
                                         timestep=agent["timestep"], horizon=agent["horizon"])

    def _tick(self, name, t0):
        """stage timer of ``step_timing``: host wall time since t0; with ``accelerator.timing: device`` the stream is
        drained first, so the figure is the stage's own GPU + host time instead of the time to issue it"""
        if self._timing_sync:
            torch.cuda.current_stream(self.device).synchronize()
        t1 = time.perf_counter()
        self.step_timing[name] = (t1 - t0) * 1e3
        return t1

    def evaluate_scenario(self, predictions, ego_pos, ego_orientation, ego_pos_cl, ego_v, timestep, cosy_cl=None):
        """ref: interface.py:148-214.  Besides the visible area (the return value) the call leaves
        ``self.step_timing``: milliseconds per stage of this planning step -- ``obstacles_ms`` (state cache),
        ``visibility_ms`` (ray fan, cell classes), ``spawn_ms`` (phantom sampling / rule families + predictions),
        ``agents_ms`` (registry), ``evaluate_scenario_ms`` (all of it), later ``assessment_batch_ms`` /
        ``assessment_single_ms`` (metric sweep calls; the latter accumulates over the per-trajectory calls of the step
        and counts them in ``assessment_single_calls``).  Host wall times of asynchronous work = the time to ISSUE it,
        unless ``accelerator.timing: device`` (see :meth:`_tick`)."""
        t_all = t0 = time.perf_counter()
        self.step_timing = {"timestep": int(timestep), "mode": "device" if self._timing_sync else "issue"}
        self.set_coordinate_system(cosy_cl)
        self._update_time_step(timestep)
        self._add_real_agents()
        self.predictions = predictions
        self.ego_pos = np.asarray(ego_pos, dtype=np.float64)
        self.ego_orientation = float(ego_orientation)
        self.ego_pos_cl = ego_pos_cl
        self.agent_manager.reset()
        self.metrics.invalidate()
        self.spawn_points = []

        self.fo_obstacles.update(self.timestep)
        t0 = self._tick("obstacles_ms", t0)
        # The GPU side of the step -- ray fan, cell classes, visible objects, phantom sampling and / or the reference's rule
        # families, their agents and predictions, the sweep's agent table -- is ONE native call (fo_step_run through a
        # PlanningStep without candidate trajectories; ``accelerator.one_call: False`` queues the stage calls instead).
        # Nothing of interface.py:186-198's find_spawn_points -> add_agent loop runs on the host; the spawn-point list is the
        # reference's host view, read back when somebody looks at it.
        sm, sl = self.sensor_model, self.spawn_locator
        if self._one_call:
            if self._scene_step is None:
                from .step import PlanningStep
                empty = torch.empty((0, sl.T), dtype=torch.float64, device=self.device)
                self._scene_step = PlanningStep(sm, sl, self.metrics.sweep, empty, empty, empty, empty, empty, mode="reduced",
                                                mirror=True)
            sm.timestep = self.timestep
            sl.spawn_points, sl._rule_points, sl._n_cell_points = [], [], 0   # (last step's list: only an outside holder keeps it alive)
            # (the obstacle rows travel with the step, and so does the copy of the hit ids / visibility flags back: the
            # reference's visible-object bookkeeping -- visible_objects_timestep, current_visible, obstacle_occlusions, the
            # visible multipolygon -- is applied when somebody looks at it, or when the obstacles move on to the next step)
            sm.stage_obstacles(self.fo_obstacles)
            self._scene_step.run(self.ego_pos, self.ego_orientation, ego_v, self.ego_pos_cl)
            sm.adopt_step(self.ego_pos, self.ego_orientation)
            sm.defer_visible_objects(self.timestep, self.fo_obstacles)
            t0 = self._tick("visibility_ms", t0)
            self.spawn_points = sl.lazy_spawn_points()
        else:
            sm.calc_visible_and_occluded_area(timestep=self.timestep, ego_pos=self.ego_pos,
                                              ego_orientation=self.ego_orientation, obstacles=self.fo_obstacles)
            self.fo_obstacles.update_multipolygon()
            t0 = self._tick("visibility_ms", t0)
            self.spawn_points = sl.find_spawn_points(self.ego_pos, self.ego_orientation, self.ego_pos_cl, ego_v, lazy=True)
        t0 = self._tick("spawn_ms", t0)
        self.agent_manager.attach_batch(self.spawn_locator.batch)
        if self.debug:
            for sp in self.spawn_points:
                print("Phantom agent of type {} added to scenario at position {}".format(sp.agent_type, sp.position))
        self.agent_manager.update_real_agents(self.predictions)
        if self.include_real_agents and self.predictions:
            # EXTENSION (accelerator.include_real_agents, off by default): the real agents' predictions join the sweep
            types = {}
            for oid in self.predictions:
                ob = self.cr_scenario.obstacle_by_id(oid) if hasattr(self.cr_scenario, "obstacle_by_id") else None
                t = getattr(ob, "obstacle_type", None)
                types[oid] = getattr(t, "value", t) or "car"
            self.agent_manager.set_external_predictions(self.predictions, types)
        if self._one_call and not (self.agent_manager._manual or self.agent_manager._external):
            self.metrics.agents_uploaded()        # fo_step_run wrote the sweep's agent table from the device batch
        self._tick("agents_ms", t0)
        if self._timing_sync:          # (counting them reads the device)
            self.step_timing["n_spawn_points"] = len(self.spawn_points)
        self._tick("evaluate_scenario_ms", t_all)
        return self.sensor_model.visible_area

    def trajectory_safety_assessment(self, trajectory):
        t0 = time.perf_counter()
        metrics, safety_assessment = self.metrics.evaluate_metrics(trajectory)
        st = self.step_timing
        st["assessment_single_ms"] = st.get("assessment_single_ms", 0.0) + (time.perf_counter() - t0) * 1e3
        st["assessment_single_calls"] = st.get("assessment_single_calls", 0) + 1
        return metrics, safety_assessment

    # ---------------------------------------------------------------------------------------- batched entry (new)
    def trajectory_safety_assessment_batch(self, trajectories, mode="reduced", shard=None):
        """All candidates of a planning step in one launch.  ``trajectories``: list of trajectory objects
        (``.cartesian.{x,y,theta,v,a}``) or a dict of [M,T] arrays / device tensors.  Returns a
        :class:`~frenetix_occlusion.metrics.metric.BatchAssessment` (``.cost [M,16]``, ``.safe [M]`` on the device), or
        None when the host knows that there are no phantom agents (every trajectory is then safe, metric.py:44-45;
        while the phantom set of the step has not been looked at from the host, its size stays in HBM and the sweep runs
        over a possibly empty set -- every trajectory safe as well).  With
        ``mode='full'`` and a list input, later ``trajectory_safety_assessment(t)`` calls for the same objects are
        served from this batch.

        ``shard`` (one process per GPU, BASELINE configs[3]): True (the default ``torch.distributed`` group), a process
        group or a :class:`~frenetix_occlusion.distributed.CostGather` -- every rank runs the same planning step
        (``evaluate_scenario`` is replicated: cheaper than a broadcast) and calls this with the same candidates; each
        evaluates its block of them and ONE all-gather of the cost rows gives every rank ``cost [M,16]`` / ``safe [M]`` of
        all candidates (``distributed.select_trajectory`` then picks the same trajectory everywhere)."""
        remember = trajectories if (mode == "full" and not isinstance(trajectories, dict)) else None
        t0 = time.perf_counter()
        ba = self.metrics.evaluate_batch(trajectories, mode=mode, remember=remember, shard=shard)
        self._tick("assessment_batch_ms", t0)
        self.step_timing["assessment_batch_size"] = 0 if ba is None else len(ba)
        return ba

    def future_visibility_batch(self, trajectories, t_stride=5, n_rays=192):
        """EXTENSION, not part of the reference: per candidate trajectory and every ``t_stride``-th sample, the number
        of currently occluded cells the ego would see from there and the visible polygon's area
        (:meth:`SensorModel.future_visibility`); a planner can turn it into the ``occ_*`` cost terms the reference
        leaves unused.  Returns device tensors ``(revealed [M,K] int32, area [M,K] float64)``."""
        from .metrics.metric import trajectories_to_arrays
        arr = trajectories_to_arrays(trajectories)
        return self.sensor_model.future_visibility(arr["x"], arr["y"], t_stride=t_stride, n_rays=n_rays)

    def _update_time_step(self, timestep):
        self.timestep = timestep
        self.sensor_model.timestep = timestep
        self.agent_manager.timestep = timestep

    @staticmethod
    def _load_config(filepath=None):
        """interface.py:227-238 (there the default path is computed but not used; here it is)"""
        if not filepath:
            filepath = os.path.join(os.path.dirname(__file__), "config", "config.yaml")
        with open(filepath, "r") as f:
            return yaml.safe_load(f)
