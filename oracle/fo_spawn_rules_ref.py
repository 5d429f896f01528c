"""The reference's spawn rule families restated on the ray/cell discretisation in NumPy / Python
(ref: spawn_locator.py:80-139, 323-476, 481-578, 678-741).

TEST INFRASTRUCTURE ONLY: this is the checker of the device implementation (csrc/fo_spawn_rules.hpp,
fo_scene_spawn_rules) -- an independent statement of the same definitions, with plain loops, scipy's connected
components and convex hull.  tests/ import it; the product package never does (it was the product's host path for
`spawn.mode: rules` until round 3: 0.9 ms per step in the interpreter).

The reference evaluates these rules with shapely/GEOS on the visible / occluded polygons; here the same predicates are
asked of the per-step cell classes (fo_scene_visibility):
    line.intersects(area)                -> some sample of the line (step cs/8) lies in a cell of that class
    point.buffer(r).intersects(area)     -> the disc touches a cell square of that class
    point.buffer(r).within(road)         -> every cell square the disc touches is road
    area.buffer(b).exterior & line       -> samples where "disc of radius b touches the area" flips along the line
Rules restated: ego intention from the curvature of the next 40 m of the reference path (:729-741), pedestrian behind a
visible static obstacle (:323-476), pedestrian behind a turn (:481-578), Car / Bicycle behind a visible dynamic
obstacle (:145-317; the candidate region is sampled on a 0.25 m lattice with exact lanelet / wedge / distance tests
and the cell classes for "occluded", the rectangle fit on a 0.1 m lattice).
PARITY: the GEOMETRY is unpinned (no GEOS here, no reference test; known-answer scenes in tests/test_spawn_rules.py).  Pinned
to the reference's own, unmodified SpawnLocator code (tests/golden/relevant_lanelets.npz, gen_golden.py relevant): find_spawn_points'
orchestration -- s_threshold, the 40 m window, which family runs under which intention and switch, the order of the list -- the
intention thresholds, the nearest-vertex rule with ties, and the dynamic rule's topology side (which intersection is the ego's,
which lanelets are relevant with and without one).
"""
import math
from typing import List, Optional

import numpy as np

from frenetix_occlusion.spawn_locator import SpawnPoint
from frenetix_occlusion.utils.curvilinear import curvature, pathlength

ROAD, VISIBLE, OCCLUDED = 1, 2, 4      # class bits of fo_scene_visibility (include/fo_hip.h)

S_THRESHOLD_TIME, MIN_S_THRESHOLD = 4.0, 25.0              # spawn_locator.py:65-66
MAX_DISTANCE_TO_OTHER_OBSTACLE = 30.0                      # :69
MIN_DISTANCE_BETWEEN_PEDESTRIANS = 5.0                     # :73
TOLERANCE_SAME_DIRECTION = math.radians(20.0)              # :68
BUFFER_AROUND_VEHICLE_FROM_SIDE = 12.0                     # :70
MIN_AREA_THRESHOLD = 10.0                                  # :71
AGENT_AREA_LIMITS = {"Car": 9.0, "Bicycle": 1.7}           # :72
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
    def __init__(self, config, ref_path, cosy_cl, lane_yaw_at, lanelet_of, fo_obstacles, debug=False, lanelets=None,
                 intersections=None):
        """lane_yaw_at(xy) -> lanelet heading or None; lanelet_of(xy) -> Lanelet or None; lanelets / intersections
        (scenario.Lanelet list, list of {'incomings': [{'incoming','right','straight','left'}]}) feed the
        dynamic-obstacle rule"""
        sl = config["spawn_locator"]
        self.behind_dynamic = bool(sl.get("spawn_point_behind_dynamic_obstacle", True)) and lanelets is not None
        self.max_dynamic = int(sl.get("max_dynamic_spawn_points", 1))
        self.lanelets = list(lanelets or [])
        self.intersections = list(intersections or [])
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
        return SpawnRules.intention_of(curvature(reference))

    @staticmethod
    def intention_of(k):
        """the thresholds of spawn_locator.py:735-741 on a curvature array (pinned: tests/golden/relevant_lanelets.npz)"""
        if k.max() > 0.10:
            return "left turn"
        if k.min() < -0.10:
            return "right turn"
        return "straight ahead"

    def find(self, view: CellView, ego_pos, ego_cl, ego_v, ego_orientation=0.0) -> List[SpawnPoint]:
        self.ego_orientation = float(ego_orientation)
        self.ego_pos, self.ego_cl = np.asarray(ego_pos, dtype=np.float64), np.asarray(ego_cl, dtype=np.float64)
        self.s_threshold = self.ego_cl[0] + max(float(ego_v) * S_THRESHOLD_TIME, MIN_S_THRESHOLD)     # :113
        self.reference, self.reference_s = self.reference_window(self.ego_cl)
        intention = self.ego_intention(self.reference)
        out: List[SpawnPoint] = []
        if self.behind_dynamic and intention in ("straight ahead", "left turn"):          # :124-126
            out += self.behind_dynamic_obstacle(view)
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

    # ---- spawn_locator.py:145-317
    def _lanelets_at(self, xy):
        from frenetix_occlusion.scenario import points_in_polygon
        q = np.asarray(xy, dtype=np.float64).reshape(1, 2)
        return [ll for ll in self.lanelets if points_in_polygon(q, ll.polygon)[0]]

    def _relevant_lanelet_ids(self):
        """other incomings / inner lanelets of the intersection the ego approaches, or -- without an intersection --
        the oncoming neighbours of the lanelets along the reference path (:171-202)"""
        ego_ll = self.lanelet_of(self.ego_pos)
        if ego_ll is None:
            return None, set(), set()
        for it in self.intersections:
            inc, inner = set(), set()
            for e in it["incomings"]:
                inc.update(e["incoming"])
                inner.update(e["left"]); inner.update(e["right"]); inner.update(e["straight"])
            if ego_ll.lanelet_id in inc or ego_ll.lanelet_id in inner:
                rel = (inc | inner) - {ego_ll.lanelet_id}
                return it, rel, inner
        rel = set()
        for i in range(0, len(self.reference), 5):
            ll = self.lanelet_of(self.reference[i])
            if ll is not None and ll.adj_left is not None:
                rel.add(ll.adj_left)
        return None, rel, set()

    def behind_dynamic_obstacle(self, view: CellView) -> List[SpawnPoint]:
        from scipy import ndimage
        from frenetix_occlusion.scenario import points_in_polygon
        pts: List[SpawnPoint] = []
        vis = [o for o in self.fo_obstacles if o.current_visible and o.obstacle_role == "dynamic"]
        vis.sort(key=lambda o: float(np.linalg.norm(self.ego_pos - o.current_pos)))
        if not vis:
            return pts
        intersection, relevant, inner = self._relevant_lanelet_ids()
        by_id = {ll.lanelet_id: ll for ll in self.lanelets}
        for ob in vis:
            if str(ob.obstacle_type).lower() in ("bicycle", "pedestrian"):
                continue
            if len(pts) > self.max_dynamic:                                            # :212 (Q11)
                break
            if np.linalg.norm(self.ego_pos - ob.current_pos) > MAX_DISTANCE_TO_OTHER_OBSTACLE:
                continue
            ob_lls = self._lanelets_at(ob.current_pos)
            if not any(ll.lanelet_id in relevant for ll in ob_lls):
                continue
            try:
                ob_cl = self.cosy_cl.convert_to_curvilinear_coords(ob.current_pos[0], ob.current_pos[1])
            except Exception:
                continue
            if ob_cl[0] < self.ego_cl[0] + 3.0 or abs(ob_cl[1]) > 15.0:                # :234
                continue
            polys = [ll.polygon for ll in ob_lls if ll.lanelet_id in relevant]
            if intersection is not None and all(ll.lanelet_id in inner for ll in ob_lls):
                first = next(ll for ll in ob_lls if ll.lanelet_id in relevant)
                if first.predecessors and first.predecessors[0] in by_id:
                    polys.append(by_id[first.predecessors[0]].polygon)                # :249-252
            # candidate region on a 0.25 m lattice around the obstacle
            h = 0.25
            rad = BUFFER_AROUND_VEHICLE_FROM_SIDE
            ax = np.arange(-rad, rad + 0.5 * h, h)
            gx, gy = np.meshgrid(ob.current_pos[0] + ax, ob.current_pos[1] + ax)
            q = np.stack((gx.ravel(), gy.ravel()), -1)

            def member(q):
                ok = np.zeros(len(q), dtype=bool)
                for pl in polys:
                    ok |= points_in_polygon(q, pl)                                      # possible_polygon (:255)
                diff = abs(ob.current_orientation - self.ego_orientation) % (2.0 * math.pi)
                if math.pi - TOLERANCE_SAME_DIRECTION <= diff <= math.pi + TOLERANCE_SAME_DIRECTION:
                    ok &= _behind_rect(self.ego_pos, q, ob.current_corner_points)       # the obstacle's own shadow (:264)
                else:
                    ok &= np.array([bool(view.class_at(p) & OCCLUDED) for p in q])      # global occluded area (:272)
                ok &= np.hypot(q[:, 0] - ob.current_pos[0], q[:, 1] - ob.current_pos[1]) <= rad
                ok &= _rect_distance_points(q, ob.current_pos, ob.current_orientation, ob.length, ob.width) > 1.0
                return ok
            inside = member(q).reshape(gx.shape)
            lab, n_lab = ndimage.label(inside)                                           # largest part (:279-281)
            if n_lab == 0:
                continue
            sizes = ndimage.sum(inside, lab, index=np.arange(1, n_lab + 1))
            k = int(np.argmax(sizes)) + 1
            region = lab == k
            if sizes[k - 1] * h * h < MIN_AREA_THRESHOLD:
                continue
            center = np.array([gx[region].mean(), gy[region].mean()])
            if not any(ll.lanelet_id in relevant for ll in self._lanelets_at(center)):
                continue

            def in_region(p):
                """membership of arbitrary points: predicate true and nearest lattice node belongs to the chosen part"""
                p = np.asarray(p, dtype=np.float64).reshape(-1, 2)
                ix = np.rint((p[:, 0] - gx[0, 0]) / h).astype(int)
                iy = np.rint((p[:, 1] - gy[0, 0]) / h).astype(int)
                okb = (ix >= 0) & (ix < gx.shape[1]) & (iy >= 0) & (iy < gx.shape[0])
                out = np.zeros(len(p), dtype=bool)
                out[okb] = region[iy[okb], ix[okb]]
                return out & member(p)
            ahead = ob.current_pos + 4.0 * np.array([math.cos(ob.current_orientation), math.sin(ob.current_orientation)])
            if in_region(ahead)[0]:                                                       # region in front of it (:297)
                continue
            yaw = self.lane_yaw_at(center)
            if yaw is None:
                continue
            car = _fit_rectangle(center, 5.5, 2.5, yaw, in_region)                       # :695-711
            if car is None:
                continue
            bike = _fit_rectangle(car["centroid"], 2.0, 1.0, yaw, in_region)
            for key, fit in (("Car", car), ("Bicycle", bike)):
                if fit is not None and fit["area"] >= AGENT_AREA_LIMITS[key] and fit["jaccard"] > 0.98:
                    pts.append(SpawnPoint(np.array(fit["centroid"]), key, None, "behind_dynamic_obstacle", None))
        return pts


def _behind_rect(ego, q, corners):
    """points whose sight line from `ego` crosses the convex quadrilateral (the obstacle's shadow wedge)"""
    c = np.asarray(corners, dtype=np.float64)
    d = q - ego[None]
    hit = np.zeros(len(q), dtype=bool)
    for i in range(4):
        a, e = c[i], c[(i + 1) % 4] - c[i]
        den = d[:, 0] * e[1] - d[:, 1] * e[0]
        w = a - ego
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (w[0] * e[1] - w[1] * e[0]) / den          # along the sight line, 1 = the point itself
            u = (w[0] * d[:, 1] - w[1] * d[:, 0]) / den
        hit |= (np.abs(den) > 1e-14) & (t >= 0.0) & (t <= 1.0) & (u >= 0.0) & (u <= 1.0)
    return hit


def _rect_distance_points(q, center, yaw, length, width):
    c, s_ = math.cos(yaw), math.sin(yaw)
    dx, dy = q[:, 0] - center[0], q[:, 1] - center[1]
    lx, ly = c * dx + s_ * dy, -s_ * dx + c * dy
    ex, ey = np.maximum(np.abs(lx) - length / 2.0, 0.0), np.maximum(np.abs(ly) - width / 2.0, 0.0)
    return np.hypot(ex, ey)


def _fit_rectangle(center, length, width, yaw, in_region, h=0.1):
    """oriented rectangle clipped to the region, on a 0.1 m lattice: area, centroid and the Jaccard similarity of the
    clipped shape with its minimum rotated rectangle (spawn_locator.py:695-726)"""
    nx_, ny_ = int(round(length / h)), int(round(width / h))
    u = (np.arange(nx_) + 0.5) * h - length / 2.0
    v = (np.arange(ny_) + 0.5) * h - width / 2.0
    uu, vv = np.meshgrid(u, v)
    c, s_ = math.cos(yaw), math.sin(yaw)
    p = np.stack((center[0] + c * uu.ravel() - s_ * vv.ravel(), center[1] + s_ * uu.ravel() + c * vv.ravel()), -1)
    ok = in_region(p)
    if not ok.any():
        return None
    area = float(ok.sum()) * h * h
    pts = p[ok]
    centroid = pts.mean(axis=0)
    if ok.all():
        return {"area": area, "centroid": centroid, "jaccard": 1.0}
    # minimum rotated rectangle of the clipped cells (edge directions of the convex hull), cells have size h
    from scipy.spatial import ConvexHull
    try:
        hull = pts[ConvexHull(pts).vertices]
    except Exception:
        return {"area": area, "centroid": centroid, "jaccard": 0.0}
    best = np.inf
    for i in range(len(hull)):
        e = hull[(i + 1) % len(hull)] - hull[i]
        n = np.linalg.norm(e)
        if n == 0:
            continue
        e = e / n
        a1 = hull @ e
        a2 = hull @ np.array([-e[1], e[0]])
        best = min(best, (a1.max() - a1.min() + h) * (a2.max() - a2.min() + h))
    return {"area": area, "centroid": centroid, "jaccard": min(1.0, area / best)}
