"""The reference's spawn rule families restated on the ray/cell discretisation (host side, a handful of queries per
step; ref: spawn_locator.py:80-139, 323-476, 481-578, 678-741).

The reference evaluates these rules with shapely/GEOS on the visible / occluded polygons; here the same predicates are
asked of the per-step cell classes (fo_scene_visibility):
    line.intersects(area)                -> some sample of the line (step cs/8) lies in a cell of that class
    point.buffer(r).intersects(area)     -> the disc touches a cell square of that class
    point.buffer(r).within(road)         -> every cell square the disc touches is road
    area.buffer(b).exterior & line       -> samples where "disc of radius b touches the area" flips along the line
Rules restated: ego intention from the curvature of the next 40 m of the reference path (:729-741), pedestrian behind a
visible static obstacle (:323-476), pedestrian behind a turn (:481-578).  Not restated: Car / Bicycle behind a dynamic
obstacle (:145-317; needs the intersection topology and the Jaccard rectangle fit) -- the occluded-cell sampling of
fo_scene_spawn covers vehicles.  PARITY UNPINNED (no GEOS here, no reference test): pinned by tests/test_spawn_rules.py.
"""
import math
from typing import List, Optional

import numpy as np

from .sensor_model import OCCLUDED, ROAD, VISIBLE
from .spawn_locator import SpawnPoint
from .utils.curvilinear import curvature, pathlength

S_THRESHOLD_TIME, MIN_S_THRESHOLD = 4.0, 25.0              # spawn_locator.py:65-66
MAX_DISTANCE_TO_OTHER_OBSTACLE = 30.0                      # :69
MIN_DISTANCE_BETWEEN_PEDESTRIANS = 5.0                     # :73
OFFSET_REF_PATH = {"left turn": 3.0, "right turn": 0.0}    # :76-78
PHANTOM_OFFSET_S = {"left turn": -0.5, "right turn": 0.0}
PHANTOM_OFFSET_D = {"left turn": 1.0, "right turn": -1.0}


class CellView:
    """host copy of one step's cell classes + the geometric predicates the rules need"""

    def __init__(self, cls, window):
        self.cls = np.asarray(cls, dtype=np.uint8)
        self.w = window

    def _cell(self, xy):
        ix = int(math.floor((xy[0] - self.w.x0) / self.w.cs)) - self.w.ix0
        iy = int(math.floor((xy[1] - self.w.y0) / self.w.cs)) - self.w.iy0
        return ix, iy

    def class_at(self, xy):
        ix, iy = self._cell(xy)
        if 0 <= ix < self.w.nx and 0 <= iy < self.w.ny:
            return int(self.cls[iy, ix])
        return 0

    def _cells_touching_disc(self, xy, rad):
        cs = self.w.cs
        ix0, iy0 = self._cell((xy[0] - rad, xy[1] - rad))
        ix1, iy1 = self._cell((xy[0] + rad, xy[1] + rad))
        for iy in range(iy0, iy1 + 1):
            for ix in range(ix0, ix1 + 1):
                x_lo = self.w.x0 + (self.w.ix0 + ix) * cs
                y_lo = self.w.y0 + (self.w.iy0 + iy) * cs
                qx = min(max(xy[0], x_lo), x_lo + cs)
                qy = min(max(xy[1], y_lo), y_lo + cs)
                if (qx - xy[0]) ** 2 + (qy - xy[1]) ** 2 <= rad * rad:
                    inside = 0 <= ix < self.w.nx and 0 <= iy < self.w.ny
                    yield int(self.cls[iy, ix]) if inside else 0

    def disc_touches(self, xy, rad, bit):
        return any(c & bit for c in self._cells_touching_disc(xy, rad))

    def disc_within(self, xy, rad, bit):
        return all(c & bit for c in self._cells_touching_disc(xy, rad))

    def sample_polyline(self, pts, step=None):
        """points along a polyline every `step` metres (default cs/8), with their arc length"""
        pts = np.asarray(pts, dtype=np.float64)
        step = step or self.w.cs / 8.0
        s = pathlength(pts)
        if s[-1] <= 0.0:
            return pts[:1], np.zeros(1)
        q = np.arange(0.0, s[-1] + 0.5 * step, step)
        q[-1] = min(q[-1], s[-1])
        return np.stack((np.interp(q, s, pts[:, 0]), np.interp(q, s, pts[:, 1])), -1), q

    def polyline_touches(self, pts, bit):
        p, _ = self.sample_polyline(pts)
        return any(self.class_at(q) & bit for q in p)

    def runs_inside(self, pts, bit):
        """maximal runs of consecutive samples inside cells of class `bit`: list of (first point, last point)"""
        p, _ = self.sample_polyline(pts)
        inside = np.array([bool(self.class_at(q) & bit) for q in p])
        runs, start = [], None
        for i, f in enumerate(inside):
            if f and start is None:
                start = i
            if not f and start is not None:
                runs.append((p[start], p[i - 1]))
                start = None
        if start is not None:
            runs.append((p[start], p[-1]))
        return runs


def segment_rect_distance(a, b, corners):
    """distance between segment ab and a convex quadrilateral (0 if they touch or the segment is inside)"""
    a, b, c = np.asarray(a, float), np.asarray(b, float), np.asarray(corners, float)

    def inside(p):
        s = 0
        for i in range(4):
            e, w = c[(i + 1) % 4] - c[i], p - c[i]
            cr = e[0] * w[1] - e[1] * w[0]
            if abs(cr) > 1e-12:
                if s == 0:
                    s = 1 if cr > 0 else -1
                elif (cr > 0) != (s > 0):
                    return False
        return True

    def seg_seg(p1, p2, p3, p4):
        def pt_seg(p, q1, q2):
            d = q2 - q1
            l2 = float(np.dot(d, d))
            t = 0.0 if l2 == 0 else min(1.0, max(0.0, float(np.dot(p - q1, d)) / l2))
            return float(np.linalg.norm(p - (q1 + t * d)))
        d1, d2 = p2 - p1, p4 - p3
        den = d1[0] * d2[1] - d1[1] * d2[0]
        if abs(den) > 1e-14:
            w = p3 - p1
            t = (w[0] * d2[1] - w[1] * d2[0]) / den
            u = (w[0] * d1[1] - w[1] * d1[0]) / den
            if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
                return 0.0
        return min(pt_seg(p1, p3, p4), pt_seg(p2, p3, p4), pt_seg(p3, p1, p2), pt_seg(p4, p1, p2))
    if inside(a) or inside(b):
        return 0.0
    return min(seg_seg(a, b, c[i], c[(i + 1) % 4]) for i in range(4))


class SpawnRules:
    def __init__(self, config, ref_path, cosy_cl, lane_yaw_at, lanelet_of, fo_obstacles, debug=False):
        """lane_yaw_at(xy) -> lanelet heading or None; lanelet_of(xy) -> Lanelet or None"""
        sl = config["spawn_locator"]
        self.behind_turn = bool(sl.get("spawn_points_behind_turn", True))
        self.behind_static = bool(sl.get("spawn_point_behind_static_obstacle", True))
        self.max_static = int(sl.get("max_static_spawn_points", 1))
        ped = config["agent_manager"]["pedestrian"]
        self.ped_width, self.ped_length = float(ped["width"]), float(ped["length"])
        self.ref_path = np.asarray(ref_path, dtype=np.float64)
        self.ref_s = pathlength(self.ref_path)
        self.cosy_cl = cosy_cl
        self.lane_yaw_at, self.lanelet_of = lane_yaw_at, lanelet_of
        self.fo_obstacles = fo_obstacles
        self.debug = debug

    # ---- spawn_locator.py:678-693, 729-741
    def reference_window(self, ego_cl, distance=40.0):
        i0 = int(np.argmin(np.abs(self.ref_s - ego_cl[0])))
        i1 = int(np.argmin(np.abs(self.ref_s - (ego_cl[0] + distance))))
        return self.ref_path[i0:i1], self.ref_s[i0:i1]

    @staticmethod
    def ego_intention(reference):
        if len(reference) < 3:
            return "straight ahead"
        k = curvature(reference)
        if k.max() > 0.10:
            return "left turn"
        if k.min() < -0.10:
            return "right turn"
        return "straight ahead"

    def find(self, view: CellView, ego_pos, ego_cl, ego_v) -> List[SpawnPoint]:
        self.ego_pos, self.ego_cl = np.asarray(ego_pos, dtype=np.float64), np.asarray(ego_cl, dtype=np.float64)
        self.s_threshold = self.ego_cl[0] + max(float(ego_v) * S_THRESHOLD_TIME, MIN_S_THRESHOLD)     # :113
        self.reference, self.reference_s = self.reference_window(self.ego_cl)
        intention = self.ego_intention(self.reference)
        out: List[SpawnPoint] = []
        if self.behind_static:
            out += self.behind_static_obstacle(view)
        if self.behind_turn and intention in ("left turn", "right turn"):
            sp = self.behind_turn_point(view, intention)
            if sp is not None:
                out.append(sp)
        self.last_intention = intention
        return out

    # ---- spawn_locator.py:323-476
    def behind_static_obstacle(self, view: CellView) -> List[SpawnPoint]:
        pts, s_positions = [], []
        vis = [o for o in self.fo_obstacles if o.current_visible and o.obstacle_role == "static"]
        vis.sort(key=lambda o: float(np.linalg.norm(self.ego_pos - o.current_pos)))
        visible_polys = [o.current_corner_points for o in self.fo_obstacles if o.current_visible]
        for ob in vis:
            if len(pts) > self.max_static:                                           # :365 (Q11: '>' before appending)
                break
            if np.linalg.norm(self.ego_pos - ob.current_pos) > MAX_DISTANCE_TO_OTHER_OBSTACLE:
                continue
            try:
                ob_cl = self.cosy_cl.convert_to_curvilinear_coords(ob.current_pos[0], ob.current_pos[1])
            except Exception:
                continue
            # :380 compares against ego s + s_threshold although s_threshold already contains ego s (kept as is)
            if self.ego_cl[0] + self.s_threshold < ob_cl[0] or ob_cl[0] < self.ego_cl[0] + 3.0:
                continue
            try:
                ccl = np.array(self.cosy_cl.convert_list_of_points_to_curvilinear_coords(
                    [np.array([[x], [y]]) for x, y in ob.current_corner_points], 4))
            except Exception:
                continue
            off = 0.8
            s_min, s_max = ccl[:, 0].min() - off, ccl[:, 0].max() + off
            d_min, d_max = ccl[:, 1].min() - off, ccl[:, 1].max() + off
            for s_line in (s_min, s_max):
                try:
                    line = np.array([self.cosy_cl.convert_to_cartesian_coords(s_line, d_min),
                                     self.cosy_cl.convert_to_cartesian_coords(s_line, d_max)])
                except Exception:
                    continue
                if not view.polyline_touches(line, OCCLUDED) or not view.polyline_touches(line, VISIBLE):
                    continue
                if any(segment_rect_distance(line[0], line[1], c) <= self.ped_width / 2.0 for c in visible_polys):
                    continue
                # boundary of visible_area.buffer(ped_length / 2 * 1.3) along the line (:414-415)
                b = self.ped_length / 2.0 * 1.3
                p, _ = view.sample_polyline(line)
                near = np.array([view.disc_touches(q, b, VISIBLE) for q in p])
                flips = np.nonzero(near[1:] != near[:-1])[0]
                cand = [p[i + 1] if near[i] else p[i] for i in flips]       # the sample just outside the buffered area
                if not cand:
                    continue
                if len(cand) == 1:
                    spawn = cand[0]
                else:                                                        # MultiPoint branch (:419-433)
                    ll = self.lanelet_of(ob.current_pos)
                    anchor = ll.left[0] if ll is not None else ob.current_pos
                    cand.sort(key=lambda q: float(np.linalg.norm(anchor - q)))
                    spawn = next((q for q in cand if view.class_at(q) & OCCLUDED), None)
                if spawn is None:
                    continue
                if view.disc_touches(spawn, 0.15, VISIBLE):                  # :440
                    continue
                if not view.disc_within(spawn, 0.15, ROAD):                  # :444
                    continue
                try:
                    spawn_cl = self.cosy_cl.convert_to_curvilinear_coords(spawn[0], spawn[1])
                except Exception:
                    continue
                if any(abs(s - spawn_cl[0]) <= MIN_DISTANCE_BETWEEN_PEDESTRIANS for s in s_positions):
                    continue
                yaw = self.lane_yaw_at(ob.current_pos)
                if yaw is None:
                    continue
                pts.append(SpawnPoint(np.array(spawn), "Pedestrian", np.asarray(spawn_cl),
                                      "behind static obstacle " + str(ob.obstacle_id), float(yaw) + math.pi / 2.0))
                s_positions.append(float(spawn_cl[0]))
                break                                                        # one spawn point per obstacle
        return pts

    # ---- spawn_locator.py:481-578
    def behind_turn_point(self, view: CellView, intention) -> Optional[SpawnPoint]:
        if len(self.reference) < 2:
            return None
        if intention == "left turn":
            try:
                line = np.array([self.cosy_cl.convert_to_cartesian_coords(s, OFFSET_REF_PATH["left turn"])
                                 for s in self.reference_s])
            except Exception:
                return None
        else:
            line = self.reference
        runs = view.runs_inside(line, OCCLUDED)
        if not runs:
            return None
        first = runs[0][0] if len(runs) == 1 else runs[-1][0]      # MultiLineString: the LAST part's first point (:528)
        try:
            s_int = self.cosy_cl.convert_to_curvilinear_coords(first[0], first[1])[0]
        except Exception:
            return None
        s_ph = s_int + PHANTOM_OFFSET_S[intention]
        if s_ph > self.s_threshold or s_ph < self.ego_cl[0] + 3.0:
            return None
        d_off = PHANTOM_OFFSET_D[intention] + OFFSET_REF_PATH[intention]
        try:
            pos = self.cosy_cl.convert_to_cartesian_coords(s_ph, d_off)
            while view.disc_touches(pos, 0.5, VISIBLE):
                s_ph += 0.5
                pos = self.cosy_cl.convert_to_cartesian_coords(s_ph, d_off)
        except Exception:
            return None
        for o in self.fo_obstacles:
            if o.current_visible and segment_rect_distance(pos, pos, o.current_corner_points) <= 0.5:
                return None
        yaw_e, yaw_p = self.lane_yaw_at(self.ego_pos), self.lane_yaw_at(pos)
        if yaw_e is None or yaw_p is None:
            return None
        if abs(yaw_p - yaw_e) % (2.0 * math.pi) < math.radians(45.0):
            return None
        return SpawnPoint(np.array(pos), "Pedestrian", np.array([s_ph, PHANTOM_OFFSET_D[intention]]), intention, None)
