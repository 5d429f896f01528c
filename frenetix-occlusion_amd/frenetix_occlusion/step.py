"""One planning step in one native call (``fo_step_run``, include/fo_hip.h): ray fan -> cell classes -> phantom sampling
and / or the reference's spawn rule families + predictions -> agent table -> metric sweep -> threshold reduction.

``PlanningStep`` binds a :class:`SensorModel`, a :class:`SpawnLocator`, a :class:`MetricSweep` (one ego, one context) and
the candidate-trajectory tensors of a planning loop; :meth:`run` updates the handful of per-step scalars in a
structure that was filled once and crosses the FFI once -- the results of the stage-by-stage calls bit for bit, in eight
kernel launches instead of their twelve and without their five ctypes crossings (~90 converted arguments, ~90 us of host
time per step; a step of the reference's own size needs ~90 us of GPU time).  Nothing is read back: the cost vectors
and flags stay in HBM.
"""
import ctypes as C
import math

import torch

from . import _native as N
from .sweep import SweepResult


class PlanningStep:
    def __init__(self, sensor_model, spawn_locator, sweep, x, y, theta, v, a=None, mode="reduced", lists="f64", shard=None,
                 mirror=False):
        """``shard`` (BASELINE configs[3]: one process per GPU, every rank builds the same step): True / a process group / a
        :class:`~frenetix_occlusion.distributed.CostGather` for the batch size -- this rank's step covers its contiguous
        block of the candidates (scene stage and phantoms replicated), writes the block's cost rows into the collective's
        block and :meth:`run` ends with the ONE all-gather of the path: ``out.cost_all`` = cost [M_total, 16] on every rank,
        ``out.rows`` = this rank's rows; every other output holds those rows only.

        ``mirror``: the step also copies its hit ids and visibility flags into the sensor model's pinned host mirror
        (``fo_step_t::h_mirror``) -- what ``SensorModel.defer_visible_objects`` reads the reference's visible-object side
        effects from, without a copy call of its own."""
        if not (sensor_model.ctx is spawn_locator.ctx is sweep.ctx):
            raise ValueError("PlanningStep: the three stages must share one context (one ego, one GPU)")
        if mode not in ("reduced", "pair", "full"):
            raise ValueError(f"unknown output mode '{mode}'")
        self.sm, self.sl, self.sw = sensor_model, spawn_locator, sweep
        self.ctx = sweep.ctx
        self.mode, self.lists, self.mirror = mode, lists, bool(mirror)
        dev = sweep.device
        t = lambda q: None if q is None else torch.as_tensor(q).to(device=dev, dtype=torch.float64).contiguous()
        self.traj = [t(x), t(y), t(theta), t(v), t(a)]
        self.gather = None
        if shard is not None and shard is not False:
            from .distributed import as_gather
            self.gather = as_gather(shard, int(self.traj[0].shape[0]), device=dev)
            lo, hi = self.gather.lo, self.gather.hi
            self.traj = [None if q is None else q[lo:hi] for q in self.traj]      # (row blocks of [M,T]: still contiguous)
        self.M, self.T = int(self.traj[0].shape[0]), int(self.traj[0].shape[1])
        if spawn_locator.batch is None:
            spawn_locator.batch = spawn_locator._alloc()
        self.batch = spawn_locator.batch
        A = int(self.batch.pos.shape[0])
        if lists not in N.LIST_FORMAT:
            raise ValueError(f"unknown list format '{lists}'")
        ldt = torch.float64 if lists == "f64" else torch.float32
        # (the cost rows go straight into the collective's block when that lives on this GPU -- RCCL; a collective over host
        # memory -- gloo -- takes a copy of them in run())
        self._direct = self.gather is not None and self.gather.device == dev
        self.out = SweepResult(cost=(self.gather.block() if self._direct else
                                     torch.empty((self.M, N.NC), dtype=torch.float64, device=dev)),
                               safe=torch.empty((self.M,), dtype=torch.uint8, device=dev))
        if self.gather is not None:
            self.out.rows = (self.gather.lo, self.gather.hi)
        if mode in ("pair", "full"):
            self.out.pair_f = torch.empty((N.NPF, A, self.M), dtype=torch.float64, device=dev)
            self.out.pair_i = torch.empty((N.NPI, A, self.M), dtype=torch.int32, device=dev)
        if mode == "full":
            self.out.lists_raw = torch.empty((N.NL * A * max(self.T - 1, 0) * self.M,), dtype=ldt, device=dev)
        self.out.lists_shape = (A, max(self.T - 1, 0), self.M)
        # (the list format is a member of the step structure: every run re-asserts it, whatever other callers of the
        # context have set in between)
        sweep.reserve(self.M, self.T, A, spawn_locator.T)
        self._s = None
        self._key = None

    def _fill(self, w, O):
        """everything that does not change from step to step"""
        sm, sl, b = self.sm, self.sl, self.batch
        s = N.Step()
        p = lambda q: None if q is None else q.data_ptr()
        dirs, rmax, half = sm._fan_buffers()
        poly = sm.footprint == "polygon"
        s.n_rays, s.polygon_footprint, s.fov_deg, s.r = sm.n_rays, 1 if poly else 0, sm.sensor_angle, sm.sensor_radius
        s.d_dirs, s.d_rmax, s.d_half = p(dirs), p(rmax) if poly else None, p(half) if poly else None
        s.full_circle = 1 if sm.sensor_angle >= 359.9 else 0
        s.exact_cells = 1 if sm.cell_visibility == "exact" else 0
        s.O = O
        buf = sm._buffers(w, O)
        s.win_nx, s.win_ny = w.nx, w.ny
        s.d_range, s.d_hit_id, s.d_ring, s.d_obst_vis = p(buf["rng"]), p(buf["hit"]), p(buf["ring"]), p(buf["ovis"])
        s.d_cls, s.d_occ_idx, s.d_n_occ = p(buf["cls"]), p(buf["occ"]), p(buf["n_occ"])
        s.min_ahead, s.all_occluded, s.max_agents, s.routes = sl.min_ahead, 1 if sl.all_occluded else 0, sl.max_agents, sl.routes
        s.n_path, s.d_path, s.T_agents, s.dt, s.var0, s.var_factor = int(sl.ref_path.shape[0]), p(sl._d_path), sl.T, sl.dt, sl.var0, sl.var_factor
        for i in range(4):
            s.type4[i], s.speed4[i], s.raw_l4[i], s.raw_w4[i] = int(sl._t4[i]), sl._s4[i], sl._rl[i], sl._rw[i]
            s.infl_l4[i], s.infl_w4[i] = sl._il[i], sl._iw[i]
        s.d_cell, s.d_pos0, s.d_yaw0, s.d_n = p(b.cell), p(b.pos0), p(b.yaw0), p(b.n)
        s.d_pos, s.d_yaw, s.d_v, s.d_cov, s.d_shape, s.d_raw_dims = p(b.pos), p(b.yaw), p(b.v), p(b.cov), p(b.shape), p(b.raw_dims)
        s.d_type, s.d_len = p(b.type), p(b.len)
        s.M, s.T = self.M, self.T
        s.d_x, s.d_y, s.d_theta, s.d_vel, s.d_acc = (p(q) for q in self.traj)
        o = self.out
        s.d_cost, s.d_safe, s.d_pair_f, s.d_pair_i, s.d_lists = p(o.cost), p(o.safe), p(o.pair_f), p(o.pair_i), p(o.lists_raw)
        s.list_format = N.LIST_FORMAT[self.lists]
        s.spawn_mode = N.SPAWN_MODE[sl.mode]
        if sl.mode != "cells":       # the reference's rule families, device resident
            if not getattr(sl, "_rules_ready", False):
                sl._rule_setup()
            s.n_path6, s.d_path6 = int(sl._d_path6.shape[0]), p(sl._d_path6)
            s.max_rule_points, s.d_rule_points, s.d_n_rule_points = b.n_rule_points, p(b.rule_points), p(b.rule_n)
            s.rule_types = sl.rule_types
        if self.mirror:
            s.h_mirror, s.d_mirror, s.mirror_bytes = buf["hv_host"].data_ptr(), p(buf["hv"]), int(buf["hv"].numel())
        self._buf = buf
        return s

    def run(self, ego_pos, ego_orientation, ego_v, ego_pos_cl=None) -> SweepResult:
        """queue one planning step on the current stream; returns the (reused) device outputs.  ``ego_pos_cl``: the ego's
        curvilinear position (s, d) for the spawn rule families (default: its projection on the reference path)"""
        sm, sl = self.sm, self.sl
        yaw = float(ego_orientation)
        w = sm._window_for(ego_pos)
        O = getattr(sm, "_obst", (None, None, None, 0))[3]
        key = (w.nx, w.ny, O)
        # the sensor model owns ONE buffer set, keyed by (window size, obstacle count): another caller between two runs of
        # this step (a direct calc_visible_and_occluded_area, a second PlanningStep, an upload with another obstacle count)
        # may have re-keyed it.  _buffers() re-keys it back for this step; a set that is not the one the structure points
        # at means the structure is stale -- adopt_step / defer_visible_objects read sm._buf, so both must be the same set
        if self._s is None or key != self._key or sm._buffers(w, O) is not self._buf:
            self._s, self._key = self._fill(w, O), key
        s = self._s
        # the obstacle tensors are re-allocated by every upload_obstacles: their pointers are per-step members (and the
        # tensors stay referenced here for as long as the structure points at them)
        self._obst_ref = (getattr(sm, "_obst", (None, None, None, 0)), getattr(sm, "_obst_rule", None))
        d_corn, d_cen, d_flags, _ = self._obst_ref[0]
        q = lambda t: None if t is None else t.data_ptr()
        s.d_ocorn, s.d_ocen, s.d_oflags = q(d_corn), q(d_cen), q(d_flags)
        host = getattr(sm, "_obst_host", None)       # rows staged by SensorModel.stage_obstacles: the native call copies them
        if host is not None:
            s.h_obstacles, s.d_obstacles, s.obstacles_bytes = host.ctypes.data, sm._obst_dev.data_ptr(), host.nbytes
        else:
            s.h_obstacles, s.d_obstacles, s.obstacles_bytes = None, None, 0
        if sl.mode != "cells":
            rl = self._obst_ref[1]
            if O and rl is None:
                raise RuntimeError("PlanningStep: the spawn rule families need the obstacles' headings and dimensions "
                                   "(SensorModel.upload_obstacles with an FOObstacles)")
            s.d_oyaw, s.d_odims = (q(rl[0]), q(rl[1])) if O else (None, None)
            s.rule = sl.rule_params(ego_pos, yaw, ego_pos_cl, ego_v)
        self.batch.invalidate()
        skip = sm._edge_skip_for(sm.enclosed_hole_rings(ego_pos, yaw))
        s.d_edge_skip = None if skip is None else skip.data_ptr()
        s.ego_yaw, s.ego_x, s.ego_y, s.head_x, s.head_y = yaw, float(ego_pos[0]), float(ego_pos[1]), math.cos(yaw), math.sin(yaw)
        s.win_ix0, s.win_iy0 = w.ix0, w.iy0
        s.max_dist = sl.max_distance(ego_v)
        self.ctx._check(self.ctx._lib.fo_step_run(self.ctx._h, C.byref(s), N.current_stream(sm._dev_index)))
        sm._obst_host = None                         # (consumed: the rows are in the context's pinned ring)
        # the stage objects see the step as if they had queued it themselves
        sm.window, sm.ego_pos, sm.ego_orientation, sm.edge_skip = w, ego_pos, yaw, skip
        b = self._buf
        sm.range, sm.hit_id, sm.cell_class = b["rng"], b["hit"], b["cls"]
        sm.occluded_idx_buffer, sm.n_occluded = b["occ"], b["n_occ"]
        self.sw.A, self.sw.Ta = int(self.batch.pos.shape[0]), sl.T
        self.ctx.list_format = self.lists
        if self.gather is not None:
            # the one collective of the path, queued behind the step
            self.out.cost_all = self.gather.gather(None if self._direct or self.M == 0 else self.out.cost)
        return self.out
