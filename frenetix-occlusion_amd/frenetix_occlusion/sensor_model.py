"""Visibility / occlusion of one ego pose on the GPU (host side of ``fo_scene_visibility``, include/fo_hip.h).

Mirrors the reference's ``SensorModel`` (ref: sensor_model.py:16-101): same constructor arguments (plus the
discretisation parameters), same ``calc_visible_and_occluded_area(timestep, ego_pos, ego_orientation, obstacles)``
entry, same attributes afterwards (``visible_area``, ``occluded_area``, ``visible_objects_timestep``,
``obstacle_occlusions``).  The reference builds the areas with GEOS polygon algebra; here they are a polar ray fan
(first hit per ray) plus a class per raster cell -- the discretisation DESIGN.md specifies.  ``visible_area`` is
therefore a :class:`VisibleArea` (ring polygon + cell mask) instead of a shapely geometry (documented deviation,
SURVEY 8b).  All arithmetic happens in libfo_hip.so; numpy here only packs small inputs.
"""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _native as N
from .scenario import MapGeometry

ROAD, VISIBLE, OCCLUDED = 1, 2, 4  # cell class bits (include/fo_hip.h)


def ray_dirs(n_rays, ego_yaw=0.0, fov_deg=360.0):
    """Unit ray directions, counter-clockwise.  Full circle (sensor_angle >= 359.9, sensor_model.py:121-122):
    angle_i = yaw + 2 pi i / n; open fan: n rays from yaw - fov/2 to yaw + fov/2 inclusive (:115-124)."""
    if fov_deg >= 359.9:
        ang = ego_yaw + 2.0 * np.pi * np.arange(n_rays) / n_rays
    else:
        half = np.radians(fov_deg) / 2.0
        ang = ego_yaw + np.linspace(-half, half, n_rays)
    return np.stack((np.cos(ang), np.sin(ang)), -1)


def footprint_ranges(n_rays, ego_yaw, fov_deg, r):
    """Range of the reference's sensor footprint along each ray of ``ray_dirs``.  The footprint is a polygon
    inscribed in the circle: ``Point(ego).buffer(r)`` = regular 64-gon with a vertex at world angle 0
    (sensor_model.py:121-122; shapely's default of 16 segments per quarter circle [ext]) for a full-circle sensor,
    else ego + 100 arc points (``_calc_relevant_sector``, :201-209).  Along direction phi the chord between two
    neighbouring arc points (angular pitch d) is met at r cos(d/2) / cos((phi - a0) mod d - d/2)."""
    if fov_deg >= 359.9:
        d = 2.0 * np.pi / 64.0
        rel = ego_yaw + 2.0 * np.pi * np.arange(n_rays) / n_rays
    else:
        d = np.radians(fov_deg) / 99.0
        rel = np.linspace(0.0, np.radians(fov_deg), n_rays)
    m = np.mod(rel, d)
    return r * np.cos(0.5 * d) / np.cos(m - 0.5 * d)


def half_fan_dirs(ego_yaw):
    """unit directions of the 100-point half fan about the heading; times 1.5 r they are the arc points of the polygon
    that bounds the reference's occluded area (sensor_model.py:85-87, 201-209)"""
    ang = np.linspace(ego_yaw - 0.5 * np.pi, ego_yaw + 0.5 * np.pi, 100)
    return np.stack((np.cos(ang), np.sin(ang)), -1)


def footprint_polygon(ego, yaw, fov_deg, r):
    """vertices of that polygon [n,2]"""
    if fov_deg >= 359.9:
        ang = np.arange(64) * (2.0 * np.pi / 64.0)
        return np.stack((ego[0] + r * np.cos(ang), ego[1] + r * np.sin(ang)), -1)
    half = np.radians(fov_deg) / 2.0
    ang = np.linspace(yaw - half, yaw + half, 100)
    arc = np.stack((ego[0] + r * np.cos(ang), ego[1] + r * np.sin(ang)), -1)
    return np.concatenate((np.asarray(ego, dtype=np.float64)[None], arc), axis=0)


class HoleIndex:
    """Interior rings (holes) of the road union's boundary and the per-step test "does the sensor footprint enclose
    this ring".  An enclosed ring is an interior ring of road ∩ footprint and casts no shadow in the reference, which
    walks `visible_area.exterior` only (sensor_model.py:126-131); a ring the footprint cuts open lies on the exterior
    of road ∩ footprint and does."""

    def __init__(self, geo: MapGeometry):
        self.edge_ring = np.asarray(geo.edge_ring)
        self.holes = []
        for k, pts in geo.holes():
            self.holes.append((k, pts))
        self.centres = np.array([pts.mean(axis=0) for _, pts in self.holes]).reshape(-1, 2)
        self.radii = np.array([float(np.sqrt(((pts - pts.mean(axis=0)) ** 2).sum(axis=1).max())) for _, pts in self.holes])

    def enclosed(self, ego_pos, ego_yaw, fov_deg, r, footprint="polygon"):
        """ids of the enclosed rings (all end points inside the footprint).  Cheap rejection first: the farthest
        vertex of a ring is at least as far from the ego as the ring's centroid."""
        if not self.holes:
            return ()
        ex, ey = float(ego_pos[0]), float(ego_pos[1])
        dx, dy = self.centres[:, 0] - ex, self.centres[:, 1] - ey
        near = dx * dx + dy * dy <= r * r
        if not near.any():                      # the usual case: one comparison per ring, no further work
            return ()
        ego = np.array([ex, ey])
        cand = np.nonzero(near & (self.radii <= 2.0 * r))[0]
        if len(cand) == 0:
            return ()
        from .scenario import points_in_polygon
        full = fov_deg >= 359.9
        foot = footprint_polygon(ego, ego_yaw, fov_deg, r) if not full else None
        out = []
        for i in cand:
            k, pts = self.holes[i]
            rho = np.hypot(pts[:, 0] - ego[0], pts[:, 1] - ego[1])
            inside = (rho <= r).all()
            if inside and full and footprint == "polygon":
                # inside the regular 64-gon Point(ego).buffer(r) (a vertex at world angle 0): the distance along the
                # normal of the sector's side stays below the apothem -- a handful of vector operations instead of a
                # crossing-number pass over 64 edges
                delta = 2.0 * np.pi / 64.0
                phi = np.mod(np.arctan2(pts[:, 1] - ego[1], pts[:, 0] - ego[0]), delta)
                near_side = rho * np.cos(phi - 0.5 * delta) > r * np.cos(0.5 * delta) * (1.0 - 1e-9)
                if near_side.any():             # within a hair of the polygon's side: let the exact test decide
                    inside = points_in_polygon(pts, footprint_polygon(ego, ego_yaw, fov_deg, r)).all()
            elif inside and foot is not None:
                inside = points_in_polygon(pts, foot).all()
            if inside:
                out.append(k)
        return tuple(out)

    def edge_skip(self, rings):
        """byte per boundary piece: 1 = on one of `rings`"""
        return np.isin(self.edge_ring, np.array(rings, dtype=np.int64)).astype(np.uint8)


@dataclass
class CellWindow:
    """raster window the per-step cell classes live in: window cell (ix, iy) = world raster cell (ix0+ix, iy0+iy)"""
    x0: float
    y0: float
    cs: float
    ix0: int
    iy0: int
    nx: int
    ny: int

    def centers(self, idx):
        idx = np.asarray(idx)
        ix, iy = idx % self.nx, idx // self.nx
        return np.stack((self.x0 + (self.ix0 + ix + 0.5) * self.cs, self.y0 + (self.iy0 + iy + 0.5) * self.cs), -1)

    def cell_of(self, xy):
        xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
        ix = np.floor((xy[:, 0] - self.x0) / self.cs).astype(np.int64) - self.ix0
        iy = np.floor((xy[:, 1] - self.y0) / self.cs).astype(np.int64) - self.iy0
        ok = (ix >= 0) & (ix < self.nx) & (iy >= 0) & (iy < self.ny)
        return np.where(ok, iy * self.nx + ix, -1)


class VisibleArea:
    """What ``evaluate_scenario`` hands back instead of a shapely geometry: the visible polygon's ring (device
    tensor [n_rays, 2]; for an open fan the ego position closes the polygon) and the per-cell classes."""

    def __init__(self, ego_pos, ring, rng, hit_id, cls, window: CellWindow, full_circle, bit=VISIBLE):
        self.ego_pos = np.asarray(ego_pos, dtype=np.float64)
        self.ring, self.range, self.hit_id, self.cls, self.window = ring, rng, hit_id, cls, window
        self.full_circle = full_circle
        self._bit = bit
        self._ring_h = None

    @property
    def exterior(self):
        """polygon vertices as numpy [n,2] (host copy, lazily)"""
        if self._ring_h is None:
            r = self.ring.cpu().numpy()
            self._ring_h = r if self.full_circle else np.concatenate((self.ego_pos[None], r), axis=0)
        return self._ring_h

    @property
    def area(self):
        """area of the cells of this class (cell count x cell area)"""
        return float(((self.cls & self._bit) != 0).sum().item()) * self.window.cs ** 2

    @property
    def ring_area(self):
        p = self.exterior
        x, y = p[:, 0], p[:, 1]
        return 0.5 * abs(float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y)))

    @property
    def is_empty(self):
        return not bool(((self.cls & self._bit) != 0).any().item())

    def mask(self):
        """bool device tensor [ny, nx]"""
        return (self.cls & self._bit) != 0

    def contains(self, xy):
        """class bit of the cell containing each point (bool numpy [n])"""
        idx = self.window.cell_of(xy)
        flat = self.cls.reshape(-1).cpu().numpy()
        out = np.zeros(len(idx), dtype=bool)
        ok = idx >= 0
        out[ok] = (flat[idx[ok]] & self._bit) != 0
        return out


class SensorModel:
    def __init__(self, lanelet_network, ref_path, sensor_radius=30, sensor_angle=90, debug=True, visualization=None,
                 ctx: Optional[N.Context] = None, n_rays=720, cell_size=0.5, device=0, routes=0,
                 footprint="polygon", enclosed_holes="transparent", cell_visibility="exact", share_map_with=None,
                 intersections=None, shadow_length=100.0):
        """lanelet_network: a :class:`~frenetix_occlusion.scenario.MapGeometry`, a list of
        :class:`~frenetix_occlusion.scenario.Lanelet`, or an object with ``.lanelets`` (duck-typed CommonRoad).

        footprint: "polygon" = the reference's inscribed polygon (see :func:`footprint_ranges`), "circle" = exact
        radius.  enclosed_holes: "transparent" = an interior ring of the road union that the footprint encloses casts
        no shadow (the reference walks exterior rings only, sensor_model.py:126-131), "occlude" = every boundary piece
        occludes.  cell_visibility: "exact" = cells the ray fan cannot decide (their two enclosing rays stop at
        different occluders) are settled by the reference's set algebra at the cell centre, "fan" = chord rule only.
        shadow_length: where an obstacle's occlusion polygon ends, metres along its two silhouette sight lines (100 in the
        reference, helper_functions.py:145-146; it matters for an obstacle seen from so close by that the chord between the
        two end points passes inside the sensor radius); ``math.inf`` = the physical shadow, which never ends.  Read by the
        exact cell settlement only.
        share_map_with: another SensorModel of the same scenario on the same GPU (an ego of a multi-ego run): the static
        map is neither recomputed on the host nor uploaded again, both contexts read one copy in HBM
        (``fo_scene_share_map``); ``lanelet_network`` is then ignored."""
        if (footprint not in ("polygon", "circle") or enclosed_holes not in ("transparent", "occlude")
                or cell_visibility not in ("exact", "fan")):
            raise ValueError("footprint: 'polygon' | 'circle'; enclosed_holes: 'transparent' | 'occlude'; "
                             "cell_visibility: 'exact' | 'fan'")
        self.footprint, self.enclosed_holes, self.cell_visibility = footprint, enclosed_holes, cell_visibility
        self.shadow_length = float(shadow_length)
        if not self.shadow_length > 0.0:
            raise ValueError("shadow_length: a positive length in metres, or inf")
        if not torch.cuda.is_available():
            raise RuntimeError("SensorModel needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device("cuda", int(device))
        self._dev_index = int(device)
        self.ctx = ctx or N.Context(self.device.index)
        self.ctx.call("fo_scene_set_shadow_length", self.shadow_length)
        self.lanelet_network = lanelet_network
        self.ref_path = ref_path
        self.visualization = visualization
        self.sensor_radius = float(sensor_radius)
        self.sensor_angle = float(sensor_angle)
        self.debug = debug
        self.n_rays = int(n_rays)
        self.cell_size = float(cell_size)
        self.routes = int(routes)   # candidate routes per lanelet uploaded for phantom vehicle predictions (0 = none)
        # intersections of the scenario (spawn rule "behind a dynamic obstacle"); default: the network's own, if it has any
        self.intersections = intersections if intersections is not None else getattr(lanelet_network, "intersections", None)
        # state of the last step (names of sensor_model.py:26-35)
        self.visible_area = None
        self.occluded_area = None
        self._obstacle_occlusions = {}
        self._visible_objects_timestep = None
        self._vis_pending = None          # (timestep, obstacles, pinned mirror) of a one-call step nobody has looked at yet
        self._obst_host = None            # obstacle rows staged for the next PlanningStep.run (stage_obstacles)
        self._obst_dev = None
        self.timestep = None
        self.ego_pos = None
        self.ego_orientation = None
        self.window = None
        self.cell_class = None
        if share_map_with is not None:
            self._share_map(share_map_with)
        else:
            self._set_map(lanelet_network)

    # the reference's per-step side effects (sensor_model.py:58-101, 183) -- after a one-call step they are host VIEWS of the
    # step's mirror (hit ids + visibility flags, copied by the step itself into pinned memory): read when somebody looks
    @property
    def visible_objects_timestep(self):
        self.resolve_visible_objects()
        return self._visible_objects_timestep

    @visible_objects_timestep.setter
    def visible_objects_timestep(self, v):
        self._visible_objects_timestep = v

    @property
    def obstacle_occlusions(self):
        self.resolve_visible_objects()
        return self._obstacle_occlusions

    @obstacle_occlusions.setter
    def obstacle_occlusions(self, v):
        self._obstacle_occlusions = v

    def _share_map(self, other):
        """take over another SensorModel's static map: host-side geometry by reference, device-side by fo_scene_share_map"""
        if other.device != self.device or other.cell_size != self.cell_size or (self.routes > 0 and other.routes != self.routes):
            raise ValueError("share_map_with: the other SensorModel must live on the same GPU with the same cell size "
                             "(and route count)")
        self.lanelet_network = other.lanelet_network
        self.map_geometry, self.hole_index = other.map_geometry, other.hole_index
        self._skip_key, self._edge_skip = (), None
        self.raster_origin, self.raster_dims, self.lane_yaw = other.raster_origin, other.raster_dims, other.lane_yaw
        self.route_table, self.lanelet_raster = other.route_table, other.lanelet_raster
        self.ctx.call("fo_scene_share_map", other.ctx._h)

    # ---- one-off: replaces _convert_lanelet_network (sensor_model.py:195-199)
    def _set_map(self, net):
        from .scenario import lane_yaw_raster, lanelets_of
        if isinstance(net, MapGeometry):
            geo, lanelets = net, None
        else:
            lanelets = lanelets_of(net)
            geo = MapGeometry.from_lanelets(lanelets)
        self.map_geometry = geo
        self.hole_index = HoleIndex(geo)
        self._skip_key, self._edge_skip = (), None
        margin = 2.0 * self.cell_size
        xy = geo.poly_xy
        cs = self.cell_size
        x0 = math.floor((xy[:, 0].min() - margin) / cs) * cs
        y0 = math.floor((xy[:, 1].min() - margin) / cs) * cs
        nx = int(math.ceil((xy[:, 0].max() + margin - x0) / cs))
        ny = int(math.ceil((xy[:, 1].max() + margin - y0) / cs))
        self.raster_origin, self.raster_dims = (x0, y0), (nx, ny)
        lane_yaw = lane_yaw_raster(lanelets, x0, y0, cs, nx, ny) if lanelets is not None else None
        self.lane_yaw = lane_yaw
        off = np.ascontiguousarray(geo.poly_off, dtype=np.int32)
        pxy = np.ascontiguousarray(geo.poly_xy, dtype=np.float64)
        edges = np.ascontiguousarray(geo.edges, dtype=np.float64)
        org = np.array([x0, y0], dtype=np.float64)
        dims = np.array([nx, ny], dtype=np.int32)
        ly = np.ascontiguousarray(lane_yaw, dtype=np.float64) if lane_yaw is not None else None
        self.ctx.call("fo_scene_set_map", len(off) - 1, off.ctypes.data, pxy.ctypes.data, len(edges),
                      edges.ctypes.data if len(edges) else None, cs, margin,
                      ly.ctypes.data if ly is not None else None, org.ctypes.data, dims.ctypes.data)
        if len(edges):
            line = np.ascontiguousarray(geo.edge_line, dtype=np.int32)
            self.ctx.call("fo_scene_set_edge_lines", len(edges), line.ctypes.data)
        if lanelets:
            self._set_topology(lanelets)
            # centre lines: a pedestrian spawned behind a turn heads for the centre of its lanelet (agent.py:459-467)
            coff = np.zeros(len(lanelets) + 1, dtype=np.int32)
            coff[1:] = np.cumsum([len(ll.center) for ll in lanelets])
            cxy = np.ascontiguousarray(np.concatenate([np.asarray(ll.center, dtype=np.float64).reshape(-1, 2) for ll in lanelets]))
            self.ctx.call("fo_scene_set_centerlines", len(lanelets), coff.ctypes.data, cxy.ctypes.data)
        self.route_table = self.lanelet_raster = None
        if self.routes > 0 and lanelets is not None:
            from .scenario import RouteTable, lanelet_index_raster
            tab = RouteTable.from_lanelets(lanelets, R=self.routes)
            ras = np.ascontiguousarray(lanelet_index_raster(lanelets, x0, y0, cs, nx, ny), dtype=np.int32)
            first, count = np.ascontiguousarray(tab.first, np.int32), np.ascontiguousarray(tab.count, np.int32)
            rxy, rs = np.ascontiguousarray(tab.xy, np.float64), np.ascontiguousarray(tab.s, np.float64)
            self.ctx.call("fo_scene_set_routes", len(lanelets), tab.R, first.ctypes.data, count.ctypes.data, len(rs),
                          rxy.ctypes.data if len(rs) else None, rs.ctypes.data if len(rs) else None, ras.ctypes.data)
            self.route_table, self.lanelet_raster = tab, ras

    def _set_topology(self, lanelets):
        """lanelet topology the spawn rule families read on the device (fo_scene_set_topology): first vertex of the left
        bounds, first predecessor, left neighbour, the intersections' incoming / inner lanelets (spawn_locator.py:171-202,
        249-252, 424).  Part of the static map: uploaded by its owner, shared with it."""
        from .scenario import normalize_intersections
        index = {ll.lanelet_id: i for i, ll in enumerate(lanelets)}
        P = len(lanelets)
        left0 = np.ascontiguousarray([ll.left[0] for ll in lanelets], dtype=np.float64)
        pred0 = np.array([index.get(ll.predecessors[0], -1) if ll.predecessors else -1 for ll in lanelets], dtype=np.int32)
        adjl = np.array([index.get(ll.adj_left, -1) if ll.adj_left is not None else -1 for ll in lanelets], dtype=np.int32)
        inters = normalize_intersections(self.intersections)
        off, lan, kind = [0], [], []
        for it in inters:
            for e in it["incomings"]:
                for lid in e["incoming"]:
                    if lid in index:
                        lan.append(index[lid]); kind.append(0)
                for key in ("left", "right", "straight"):
                    for lid in e[key]:
                        if lid in index:
                            lan.append(index[lid]); kind.append(1)
            off.append(len(lan))
        off = np.array(off, dtype=np.int32)
        lan, kind = np.array(lan or [0], dtype=np.int32), np.array(kind or [0], dtype=np.uint8)
        self.ctx.call("fo_scene_set_topology", P, left0.ctypes.data, pred0.ctypes.data, adjl.ctypes.data, len(inters),
                      off.ctypes.data, lan.ctypes.data, kind.ctypes.data)

    def road_raster(self):
        nx, ny = self.raster_dims
        out = np.zeros((ny, nx), dtype=np.uint8)
        self.ctx.call("fo_scene_copy_raster", out.ctypes.data)
        return out

    def _window_for(self, ego_pos):
        """cells covering the 1.5 r disc about the ego (the occluded area reaches 1.5 r, sensor_model.py:85-90)"""
        cs, (x0, y0) = self.cell_size, self.raster_origin
        reach = 1.5 * self.sensor_radius
        ix0 = int(math.floor((ego_pos[0] - reach - x0) / cs))
        iy0 = int(math.floor((ego_pos[1] - reach - y0) / cs))
        n = int(math.ceil(2.0 * reach / cs)) + 1
        return CellWindow(x0, y0, cs, ix0, iy0, n, n)

    # ---- per step: replaces calc_visible_and_occluded_area (sensor_model.py:41-101)
    def upload_obstacles(self, obstacles):
        """obstacle corner points / centres / flags of the current step -> HBM (a few hundred bytes)"""
        dev = self.device
        self._obst_host = None
        if obstacles is not None and len(obstacles) > 0:
            yaw = dims = None
            if hasattr(obstacles, "arrays_full"):      # + what the spawn rule families read (headings, dimensions, role bits)
                corn, cen, flags, yaw, dims = obstacles.arrays_full()
            else:
                corn, cen, flags = obstacles.arrays() if hasattr(obstacles, "arrays") else obstacles
            O = len(flags)
            # one host buffer, one copy: corners [O,4,2] | centres [O,2] | headings [O] | dimensions [O,2] | flags [O]
            host = np.zeros(O * 105, dtype=np.uint8)
            host[:O * 64].view(np.float64)[:] = np.asarray(corn, dtype=np.float64).reshape(-1)
            host[O * 64:O * 80].view(np.float64)[:] = np.asarray(cen, dtype=np.float64).reshape(-1)
            if yaw is not None:
                host[O * 80:O * 88].view(np.float64)[:] = yaw
                host[O * 88:O * 104].view(np.float64)[:] = np.asarray(dims, dtype=np.float64).reshape(-1)
            host[O * 104:] = np.asarray(flags, dtype=np.uint8)
            # obstacles whose flags allow the dynamic-obstacle rule (present, dynamic role, no bicycle / pedestrian): a count the
            # rule launch wants from the host (fo_spawn_rule_params_t::n_dynamic_plus1); None = flags without role bits
            self._n_dyn_candidates = int(((host[O * 104:] & 13) == 5).sum()) if yaw is not None else None
            d = torch.as_tensor(host).to(dev)
            self._obst = (d[:O * 64].view(torch.float64).view(O, 4, 2), d[O * 64:O * 80].view(torch.float64).view(O, 2),
                          d[O * 104:], O)
            self._obst_rule = (d[O * 80:O * 88].view(torch.float64), d[O * 88:O * 104].view(torch.float64).view(O, 2)) \
                if yaw is not None else None
        else:
            self._obst = (None, None, None, 0)
            self._obst_rule = None
            self._n_dyn_candidates = 0
        return self._obst

    STAGE_BYTES = 64 << 10      # fo_step_t::h_obstacles: one slot of the context's pinned ring (624 obstacles)

    def stage_obstacles(self, obstacles):
        """:meth:`upload_obstacles` without the copy: the rows are packed on the host and travel with the next
        ``PlanningStep.run`` (``fo_step_t::h_obstacles``: the native call stages them through pinned memory in front of its
        first launch); their HBM home is allocated once per obstacle count, so the step structure's pointers stay put."""
        if obstacles is None or len(obstacles) == 0:
            self._obst, self._obst_rule, self._n_dyn_candidates, self._obst_host = (None, None, None, 0), None, 0, None
            return self._obst
        host = obstacles.packed() if hasattr(obstacles, "packed") else None
        if host is None or host.nbytes > self.STAGE_BYTES:      # (beyond the native call's staging slot: the plain copy)
            return self.upload_obstacles(obstacles)
        O = host.size // 105
        d = self._obst_dev
        if d is None or d.numel() != host.size:
            d = self._obst_dev = torch.empty(host.size, dtype=torch.uint8, device=self.device)
            self._obst_views = ((d[:O * 64].view(torch.float64).view(O, 4, 2), d[O * 64:O * 80].view(torch.float64).view(O, 2),
                                 d[O * 104:], O),
                                (d[O * 80:O * 88].view(torch.float64), d[O * 88:O * 104].view(torch.float64).view(O, 2)))
        self._obst, self._obst_rule = self._obst_views
        self._n_dyn_candidates = int(((host[O * 104:] & 13) == 5).sum())
        self._obst_host = host
        return self._obst

    def _buffers(self, w, O):
        """per-step outputs, allocated once per (window size, obstacle count) and reused"""
        key = (w.nx, w.ny, O, self.n_rays)
        if getattr(self, "_buf_key", None) != key:
            dev, n = self.device, self.n_rays
            if getattr(self, "_buf", None) is not None:
                # (the window or the obstacle count changed -- rare: a step still in flight may be writing the old pinned
                # mirror, which the allocator would hand to the next pinned request the moment it is dropped)
                torch.cuda.current_stream(dev).synchronize()
            # hit ids and visibility flags are read back together after a step: one allocation, one copy
            hv = torch.zeros(4 * n + max(O, 1), dtype=torch.uint8, device=dev)
            self._buf = dict(rng=torch.empty(n, dtype=torch.float64, device=dev),
                             hit=hv[:4 * n].view(torch.int32), hv=hv,
                             ring=torch.empty((n, 2), dtype=torch.float64, device=dev),
                             ovis=hv[4 * n:],
                             # whole 32-bit words: the settle kernel clears class bits with word atomics
                             cls=torch.empty((w.ny * w.nx + 3) // 4 * 4, dtype=torch.uint8,
                                             device=dev)[:w.ny * w.nx].view(w.ny, w.nx),
                             occ=torch.empty(w.nx * w.ny, dtype=torch.int32, device=dev),
                             n_occ=torch.zeros(1, dtype=torch.int32, device=dev),
                             # pinned host mirror of hv, written by a one-call step (fo_step_t::h_mirror)
                             hv_host=torch.empty(4 * n + max(O, 1), dtype=torch.uint8, pin_memory=True))
            self._buf_key = key
        return self._buf

    def fan(self, ego_orientation):
        """(dirs [n_rays,2], rmax [n_rays] or None, half [100,2] or None): the ray fan about ``ego_orientation``, the
        footprint range along each ray and the half fan that bounds the occluded area, written by the device
        (``fo_scene_fan``) into buffers owned by this object -- valid until the next call.  :func:`ray_dirs`,
        :func:`footprint_ranges` and :func:`half_fan_dirs` are the host statement of the same definitions."""
        dirs, rmax, half = self._fan_buffers()
        poly = self.footprint == "polygon"
        self.ctx.call("fo_scene_fan", self.n_rays, float(ego_orientation), self.sensor_angle, self.sensor_radius,
                      1 if poly else 0, dirs.data_ptr(), rmax.data_ptr() if poly else None,
                      half.data_ptr() if poly else None, N.current_stream(self._dev_index))
        return dirs, (rmax if poly else None), (half if poly else None)

    def _fan_buffers(self):
        if getattr(self, "_fan_buf", None) is None:
            self._fan_buf = (torch.empty((self.n_rays, 2), dtype=torch.float64, device=self.device),
                             torch.empty(self.n_rays, dtype=torch.float64, device=self.device),
                             torch.empty((100, 2), dtype=torch.float64, device=self.device))
        return self._fan_buf

    def enclosed_hole_rings(self, ego_pos, ego_orientation):
        if self.enclosed_holes != "transparent":
            return ()
        return self.hole_index.enclosed(ego_pos, ego_orientation, self.sensor_angle, self.sensor_radius, self.footprint)

    def _edge_skip_for(self, rings):
        """device byte per boundary piece: 1 = casts no shadow this step; None when nothing is skipped.  Re-uploaded
        only when the set of enclosed rings changes."""
        if not rings:
            return None
        if rings != self._skip_key:
            self._edge_skip = torch.as_tensor(self.hole_index.edge_skip(rings)).to(self.device)
            self._skip_key = rings
        return self._edge_skip

    def launch(self, ego_pos, ego_orientation, dirs=None, rmax=None, half=None):
        """queue the visibility kernels for one ego pose on the current stream; no host synchronisation.
        Obstacles are the ones of the last ``upload_obstacles``; ``dirs`` / ``rmax`` / ``half`` (device tensors, see
        :meth:`fan`) default to the fan about ``ego_orientation``."""
        self.ego_pos = np.asarray(ego_pos, dtype=np.float64)
        self.ego_orientation = float(ego_orientation)
        full = self.sensor_angle >= 359.9
        if dirs is None:
            dirs, rmax, half = self.fan(self.ego_orientation)
        skip = self._edge_skip_for(self.enclosed_hole_rings(self.ego_pos, self.ego_orientation))
        d_corn, d_cen, d_flags, O = getattr(self, "_obst", (None, None, None, 0))
        w = self._window_for(self.ego_pos)
        b = self._buffers(w, O)
        p = lambda t: t.data_ptr() if t is not None else None
        hx, hy = math.cos(self.ego_orientation), math.sin(self.ego_orientation)
        self.ctx.call("fo_scene_visibility", float(self.ego_pos[0]), float(self.ego_pos[1]), hx, hy,
                      self.sensor_radius, 1 if full else 0, 1 if self.cell_visibility == "exact" else 0, self.n_rays,
                      p(dirs), p(rmax), p(half), p(skip), O, p(d_corn),
                      p(d_cen), p(d_flags),
                      w.ix0, w.iy0, w.nx, w.ny, p(b["rng"]), p(b["hit"]), p(b["ring"]), p(b["ovis"]), p(b["cls"]),
                      p(b["occ"]), p(b["n_occ"]), N.current_stream(self._dev_index))
        self.window = w
        self.dirs, self.rmax, self.half_dirs, self.edge_skip = dirs, rmax, half, skip
        self.range, self.hit_id, self.cell_class = b["rng"], b["hit"], b["cls"]
        self.occluded_idx_buffer, self.n_occluded = b["occ"], b["n_occ"]
        self.visible_area = VisibleArea(self.ego_pos, b["ring"], b["rng"], b["hit"], b["cls"], w, full, VISIBLE)
        self.occluded_area = VisibleArea(self.ego_pos, b["ring"], b["rng"], b["hit"], b["cls"], w, full, OCCLUDED)
        return self.visible_area

    # ---- extension, not part of the reference
    def future_visibility(self, x, y, t_stride=5, n_rays=192, radius=None):
        """How much of the currently occluded area each candidate trajectory will come to see.

        x, y: [M, T] trajectory samples (numpy or device tensors).  From every ``t_stride``-th sample a world-aligned
        full fan of ``n_rays`` rays (<= 768) of length ``radius`` (default: the sensor radius) is cast against the
        static map and the obstacles of the last ``upload_obstacles``; returns device tensors
        ``revealed [M, K]`` (int32: cells of the occluded set of the last ``launch`` that lie inside that fan) and
        ``area [M, K]`` (float64: area of the polygon of hit points), K = ceil(T / t_stride).  Queued on the current
        stream, no host synchronisation.  Uses the plain chord rule of the ray fan (no exact settlement, no
        enclosed-hole or footprint-polygon refinements): it is a cost term, not a reproduction of reference output."""
        if self.window is None:
            raise RuntimeError("future_visibility needs the occluded set of a previous launch()")
        dev = self.device
        tx = x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64))
        ty = y if torch.is_tensor(y) else torch.as_tensor(np.ascontiguousarray(y, dtype=np.float64))
        tx, ty = tx.to(dev, torch.float64).contiguous(), ty.to(dev, torch.float64).contiguous()
        M, T = tx.shape
        K = (T + t_stride - 1) // t_stride
        r = float(self.sensor_radius if radius is None else radius)
        key = int(n_rays)
        if getattr(self, "_fv_dirs_key", None) != key:
            self._fv_dirs = torch.empty((key, 2), dtype=torch.float64, device=dev)
            self.ctx.call("fo_scene_fan", key, 0.0, 360.0, r, 0, self._fv_dirs.data_ptr(), None, None,
                          N.current_stream(self._dev_index))
            self._fv_dirs_key = key
        revealed = torch.empty((M, K), dtype=torch.int32, device=dev)
        area = torch.empty((M, K), dtype=torch.float64, device=dev)
        d_corn, _, d_flags, O = getattr(self, "_obst", (None, None, None, 0))
        w = self.window
        p = lambda t: t.data_ptr() if t is not None else None
        self.ctx.call("fo_scene_future_visibility", M, T, tx.data_ptr(), ty.data_ptr(), int(t_stride), key,
                      self._fv_dirs.data_ptr(), r, O, p(d_corn), p(d_flags), self.occluded_idx_buffer.data_ptr(),
                      self.n_occluded.data_ptr(), w.ix0, w.iy0, w.nx, revealed.data_ptr(), area.data_ptr(),
                      N.current_stream(self._dev_index))
        return revealed, area

    def calc_visible_and_occluded_area(self, timestep, ego_pos, ego_orientation, obstacles):
        """reference entry point.  obstacles: an FOObstacles (already updated to `timestep`) or None."""
        self.timestep = timestep
        _, _, _, O = self.upload_obstacles(obstacles)
        self.launch(ego_pos, ego_orientation)
        return self.read_visible_objects(timestep, obstacles)

    def adopt_step(self, ego_pos, ego_orientation):
        """after a one-call planning step (``PlanningStep.run`` / ``fo_step_run``) queued this object's kernels: what
        :meth:`launch` leaves behind -- the visible / occluded area objects over the step's buffers"""
        b, w = self._buf, self.window
        full = self.sensor_angle >= 359.9
        self.ego_pos, self.ego_orientation = np.asarray(ego_pos, dtype=np.float64), float(ego_orientation)
        dirs, rmax, half = self._fan_buffers()
        poly = self.footprint == "polygon"
        self.dirs, self.rmax, self.half_dirs = dirs, (rmax if poly else None), (half if poly else None)
        self.visible_area = VisibleArea(self.ego_pos, b["ring"], b["rng"], b["hit"], b["cls"], w, full, VISIBLE)
        self.occluded_area = VisibleArea(self.ego_pos, b["ring"], b["rng"], b["hit"], b["cls"], w, full, OCCLUDED)
        return self.visible_area

    def read_visible_objects(self, timestep, obstacles):
        """the side effects the reference's sensor model leaves on its obstacles (sensor_model.py:58-101, 183):
        ``visible_objects_timestep``, ``current_visible`` / ``last_visible_at_ts`` per obstacle, ``obstacle_occlusions`` --
        one device-to-host copy of the step's hit ids and visibility flags (the only point of a step where the host waits
        for the device, and only when there are obstacles)"""
        self._vis_pending = None
        if getattr(obstacles, "_pending", None) is not None:     # (a one-call step's view nobody looked at: superseded)
            obstacles._pending = None
        O = getattr(self, "_obst", (None, None, None, 0))[3]
        self._apply_visible(self._buf["hv"].cpu().numpy() if O else None, timestep, obstacles)
        return self.visible_area

    def _apply_visible(self, hv, timestep, obstacles):
        self._visible_objects_timestep = []
        self._obstacle_occlusions.clear()
        O = getattr(self, "_obst", (None, None, None, 0))[3]
        if O and hv is not None:
            hit_h = hv[:4 * self.n_rays].view(np.int32)
            vis = hv[4 * self.n_rays:4 * self.n_rays + O].astype(bool)
            E = len(self.map_geometry.edges)
            for i, obst in enumerate(obstacles):
                obst.current_visible = bool(vis[i])
                if vis[i]:
                    self._visible_objects_timestep.append(obst.obstacle_id)
                    obst.last_visible_at_ts = timestep
                    # rays stopped by this obstacle = its shadow wedge (sensor_model.py:183: obstacle_occlusions[id])
                    self._obstacle_occlusions[obst.obstacle_id] = np.nonzero(hit_h == E + i)[0]

    def defer_visible_objects(self, timestep, obstacles):
        """after a one-call step that mirrors its hit ids and visibility flags itself (``PlanningStep(mirror=True)``): the
        side effects of :meth:`read_visible_objects` become pending -- applied when one of them is looked at (this object's
        ``visible_objects_timestep`` / ``obstacle_occlusions``, an obstacle's ``current_visible`` / ``last_visible_at_ts``,
        the obstacles' ``visible_obstacle_multipolygon``) or, at the latest, when the obstacles move on to the next step"""
        O = getattr(self, "_obst", (None, None, None, 0))[3]
        if not O or obstacles is None:
            self._vis_pending = None
            self._apply_visible(None, timestep, obstacles)
            return
        self._vis_pending = (timestep, obstacles, self._buf["hv_host"])
        if hasattr(obstacles, "_pending"):
            obstacles._pending = self.resolve_visible_objects
        else:
            self.resolve_visible_objects()

    def resolve_visible_objects(self):
        pend = self._vis_pending
        if pend is None:
            return
        self._vis_pending = None
        timestep, obstacles, mirror = pend
        if hasattr(obstacles, "_pending"):
            obstacles._pending = None
        self.ctx.call("fo_step_mirror_wait")
        self._apply_visible(mirror.numpy(), timestep, obstacles)
        if hasattr(obstacles, "update_multipolygon"):
            obstacles.update_multipolygon()

    def occluded_cells(self):
        """ascending window indices of the occluded cells (device tensor view; synchronises to read the count)"""
        k = int(self.n_occluded.item())
        return self.occluded_idx_buffer[:k]
