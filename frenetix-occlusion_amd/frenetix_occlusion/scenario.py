"""Scene description without commonroad-io: CommonRoad 2020a XML -> plain arrays, plus the synthetic urban grid of
BASELINE config 3.  (SURVEY §8f rank 1 "CommonRoad XML -> device map loader" and the harness row a18.)

What the hot path needs from a scenario (ref: sensor_model.py:195-199, fo_obstacle.py:79-116, spawn_locator.py):
lanelet polygons, the occluding boundary of their union, obstacle rectangles per time step, lanelet headings.
Everything here is one-off host work (numpy); the per-step work happens in libfo_hip.so.
"""
import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class Lanelet:
    lanelet_id: int
    left: np.ndarray            # [n,2]
    right: np.ndarray           # [n,2]
    successors: List[int] = field(default_factory=list)
    predecessors: List[int] = field(default_factory=list)
    adj_left: Optional[int] = None
    adj_left_same_direction: Optional[bool] = None
    adj_right: Optional[int] = None
    adj_right_same_direction: Optional[bool] = None
    lanelet_type: str = "urban"

    @property
    def center(self):
        return 0.5 * (self.left + self.right)

    @property
    def polygon(self):
        """left bound followed by the reversed right bound (CommonRoad lanelet polygon [ext])"""
        return np.concatenate((self.left, self.right[::-1]), axis=0)


_CVX = np.array([-1.0, -1.0, 1.0, 1.0])     # Rectangle vertices in units of the half dimensions (helper_functions.py:99-112)
_CVY = np.array([-1.0, 1.0, 1.0, -1.0])


@dataclass
class Obstacle:
    obstacle_id: int
    role: str                   # "static" | "dynamic"
    obstacle_type: str          # commonroad ObstacleType value, e.g. "car", "bicycle"
    length: float
    width: float
    initial_time_step: int
    initial: np.ndarray         # x, y, yaw, v
    states: np.ndarray          # [n,4] x, y, yaw, v for time steps initial+1 ...

    def pose_at(self, timestep):
        """fo_obstacle.py:79-93: rel = step - initial; 0 -> initial state, >= 1 -> state_list[rel-1], else absent."""
        if self.role == "static":
            return self.initial
        rel = timestep - self.initial_time_step
        if rel == 0:
            return self.initial
        if rel >= 1 and rel - 1 < len(self.states):
            return self.states[rel - 1]
        return None

    def corners(self, pose):
        """helper_functions.py:99-112: Rectangle vertices (-l/2,-w/2), (-l/2,w/2), (l/2,w/2), (l/2,-w/2) rotated + shifted"""
        # (written out per coordinate -- x = px + (c vx - s vy), y = py + (s vx + c vy) -- so that FOObstacles.update, which
        # does all obstacles of a step in one array expression, produces the same bits)
        l2, w2 = self.length / 2.0, self.width / 2.0
        c, s = math.cos(pose[2]), math.sin(pose[2])
        out = np.empty((4, 2))
        out[:, 0] = pose[0] + (c * (_CVX * l2) - s * (_CVY * w2))
        out[:, 1] = pose[1] + (s * (_CVX * l2) + c * (_CVY * w2))
        return out


@dataclass
class Scenario:
    dt: float
    lanelets: List[Lanelet]
    obstacles: List[Obstacle]
    intersections: List[dict] = field(default_factory=list)
    ego_initial: Optional[np.ndarray] = None  # x, y, yaw, v
    benchmark_id: str = ""

    def lanelet_by_id(self, lid) -> Lanelet:
        for ll in self.lanelets:
            if ll.lanelet_id == lid:
                return ll
        raise KeyError(lid)

    def obstacle_by_id(self, oid) -> Optional["Obstacle"]:
        """commonroad's Scenario.obstacle_by_id: None when there is no such obstacle"""
        for ob in self.obstacles:
            if ob.obstacle_id == oid:
                return ob
        return None

    def obstacle_arrays(self, timestep):
        """corner points [O,4,2], centres [O,2], flags [O] (bit0 present, bit1 occludes: not a bicycle, Q10)"""
        O = len(self.obstacles)
        corn = np.zeros((O, 4, 2))
        cen = np.zeros((O, 2))
        flags = np.zeros(O, dtype=np.uint8)
        yaw = np.zeros(O)
        for i, ob in enumerate(self.obstacles):
            pose = ob.pose_at(timestep)
            if pose is None:
                continue
            corn[i] = ob.corners(pose)
            cen[i] = pose[:2]
            yaw[i] = pose[2]
            flags[i] = 1 | (0 if ob.obstacle_type == "bicycle" else 2)
        return corn, cen, flags, yaw


# ------------------------------------------------------------------------------------------------ XML
def _pts(node):
    return np.array([[float(p.find("x").text), float(p.find("y").text)] for p in node.findall("point")])


def _exact(node, name, default=0.0):
    n = node.find(name)
    if n is None:
        return default
    e = n.find("exact")
    if e is not None:
        return float(e.text)
    lo, hi = n.find("intervalStart"), n.find("intervalEnd")
    if lo is not None and hi is not None:
        return 0.5 * (float(lo.text) + float(hi.text))
    return default


def _state(node):
    p = node.find("position").find("point")
    return np.array([float(p.find("x").text), float(p.find("y").text), _exact(node, "orientation"),
                     _exact(node, "velocity")]), int(round(_exact(node, "time")))


def load_commonroad_xml(path) -> Scenario:
    root = ET.parse(path).getroot()
    dt = float(root.attrib.get("timeStepSize", 0.1))
    lanelets = []
    for ln in root.findall("lanelet"):
        ll = Lanelet(int(ln.attrib["id"]), _pts(ln.find("leftBound")), _pts(ln.find("rightBound")))
        ll.successors = [int(s.attrib["ref"]) for s in ln.findall("successor")]
        ll.predecessors = [int(s.attrib["ref"]) for s in ln.findall("predecessor")]
        al, ar = ln.find("adjacentLeft"), ln.find("adjacentRight")
        if al is not None:
            ll.adj_left, ll.adj_left_same_direction = int(al.attrib["ref"]), al.attrib.get("drivingDir") == "same"
        if ar is not None:
            ll.adj_right, ll.adj_right_same_direction = int(ar.attrib["ref"]), ar.attrib.get("drivingDir") == "same"
        lt = ln.find("laneletType")
        if lt is not None and lt.text:
            ll.lanelet_type = lt.text
        lanelets.append(ll)
    obstacles = []
    for tag, role in (("staticObstacle", "static"), ("dynamicObstacle", "dynamic")):
        for ob in root.findall(tag):
            rect = ob.find("shape").find("rectangle")
            if rect is None:
                continue  # circles / polygons do not occur in the example scenarios
            init, t0 = _state(ob.find("initialState"))
            states = []
            traj = ob.find("trajectory")
            if traj is not None:
                states = [_state(s)[0] for s in traj.findall("state")]
            obstacles.append(Obstacle(int(ob.attrib["id"]), role, ob.find("type").text, float(rect.find("length").text),
                                      float(rect.find("width").text), t0, init,
                                      np.array(states).reshape(-1, 4)))
    inters = []
    for it in root.findall("intersection"):
        incs = []
        for inc in it.findall("incoming"):
            incs.append({"incoming": [int(x.attrib["ref"]) for x in inc.findall("incomingLanelet")],
                         "right": [int(x.attrib["ref"]) for x in inc.findall("successorsRight")],
                         "straight": [int(x.attrib["ref"]) for x in inc.findall("successorsStraight")],
                         "left": [int(x.attrib["ref"]) for x in inc.findall("successorsLeft")]})
        inters.append({"id": int(it.attrib["id"]), "incomings": incs})
    ego = None
    pp = root.find("planningProblem")
    if pp is not None:
        ego, _ = _state(pp.find("initialState"))
    return Scenario(dt, lanelets, obstacles, inters, ego, root.attrib.get("benchmarkID", ""))


# ------------------------------------------------------------------------------------------------ map geometry
def _signed_area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def points_in_polygon(q, poly):
    """crossing-number test, same rule as the raster kernel (half-open in y); q [n,2] -> bool [n]"""
    x, y = q[:, 0][:, None], q[:, 1][:, None]
    xi, yi = poly[:, 0][None, :], poly[:, 1][None, :]
    xj, yj = np.roll(poly[:, 0], 1)[None, :], np.roll(poly[:, 1], 1)[None, :]
    cond = (yi > y) != (yj > y)
    with np.errstate(divide="ignore", invalid="ignore"):
        xc = xi + (y - yi) * (xj - xi) / (yj - yi)
    return (np.sum(cond & (x < xc), axis=1) % 2) == 1


def union_boundary_edges(polys, eps=1e-4, tol=1e-9):
    """Occluding boundary of the union of the lanelet polygons (sensor_model.py:195-199 takes the union with GEOS and
    then walks its exterior, :131-139).  Every polygon edge is split where other polygons cross or touch it; a piece
    is boundary iff the point `eps` outside of its midpoint lies in no polygon.  Returns [E,4] (ax, ay, bx, by).
    Interior rings (city blocks enclosed by roads) are part of the list; `boundary_rings` labels them so that the
    sensor model can treat the ones enclosed by the sensor footprint the way the reference does."""
    polys = [np.asarray(p, dtype=np.float64) for p in polys]
    boxes = np.array([[p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()] for p in polys])
    out = []
    for pi, p in enumerate(polys):
        n = len(p)
        a, b = p, np.roll(p, -1, axis=0)
        ccw = _signed_area(p) > 0
        apart = ((boxes[:, 0] > boxes[pi, 2] + tol) | (boxes[:, 2] < boxes[pi, 0] - tol) |
                 (boxes[:, 1] > boxes[pi, 3] + tol) | (boxes[:, 3] < boxes[pi, 1] - tol))
        apart[pi] = True
        near = [int(qi) for qi in np.nonzero(~apart)[0]]
        splits = [[0.0, 1.0] for _ in range(n)]
        for qi in near:
            q = polys[qi]
            c, d = q, np.roll(q, -1, axis=0)
            e = (b - a)[:, None, :]                 # [n,1,2]
            f = (d - c)[None, :, :]                 # [1,m,2]
            w = c[None, :, :] - a[:, None, :]       # [n,m,2]
            den = e[..., 0] * f[..., 1] - e[..., 1] * f[..., 0]
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (w[..., 0] * f[..., 1] - w[..., 1] * f[..., 0]) / den
                u = (w[..., 0] * e[..., 1] - w[..., 1] * e[..., 0]) / den
            hit = (np.abs(den) > 1e-14) & (t > tol) & (t < 1 - tol) & (u >= -tol) & (u <= 1 + tol)
            for i, j in zip(*np.nonzero(hit)):
                splits[i].append(float(t[i, j]))
            # vertices of q lying on an edge of p (collinear overlaps start/stop there)
            l2 = np.maximum(np.sum(e[..., :] ** 2, axis=-1), 1e-300)  # [n,1] (zero-length edges stay inert)
            tt = np.sum(w * e, axis=-1) / l2                          # [n,m]
            proj = a[:, None, :] + tt[..., None] * e
            dist = np.linalg.norm(c[None, :, :] - proj, axis=-1)
            on = (dist < 1e-7) & (tt > tol) & (tt < 1 - tol)
            for i, j in zip(*np.nonzero(on)):
                splits[i].append(float(tt[i, j]))
        seg_a, seg_b = [], []
        for i in range(n):
            ts = np.unique(np.round(np.array(splits[i]), 12))
            for t0, t1 in zip(ts[:-1], ts[1:]):
                if t1 - t0 < 1e-12:
                    continue
                seg_a.append(a[i] + t0 * (b[i] - a[i]))
                seg_b.append(a[i] + t1 * (b[i] - a[i]))
        if not seg_a:
            continue
        sa, sb = np.array(seg_a), np.array(seg_b)
        ev = sb - sa
        ln = np.linalg.norm(ev, axis=1, keepdims=True)
        nrm = np.concatenate((ev[:, 1:2], -ev[:, 0:1]), axis=1) / np.maximum(ln, 1e-300)
        if not ccw:
            nrm = -nrm
        probe = 0.5 * (sa + sb) + eps * nrm
        covered = np.zeros(len(probe), dtype=bool)
        for qi in near:
            bx = boxes[qi]
            cand = (~covered) & (probe[:, 0] >= bx[0]) & (probe[:, 0] <= bx[2]) & (probe[:, 1] >= bx[1]) & (probe[:, 1] <= bx[3])
            if cand.any():
                idx = np.nonzero(cand)[0]
                covered[idx] |= points_in_polygon(probe[idx], polys[qi])
        keep = ~covered
        out.append(np.concatenate((sa[keep], sb[keep]), axis=1))
    return np.concatenate(out, axis=0) if out else np.zeros((0, 4))


def lane_yaw_raster(lanelets, x0, y0, cs, nx, ny):
    """per raster cell: heading of the nearest centre-line segment of a lanelet containing the cell centre
    (what lanelet_orientation_at_position [ext] gives spawn_locator.py:653-660); NaN off-lane."""
    out = np.full((ny, nx), np.nan)
    for ll in lanelets:
        poly = ll.polygon
        ix0 = max(int(math.floor((poly[:, 0].min() - x0) / cs)), 0)
        ix1 = min(int(math.ceil((poly[:, 0].max() - x0) / cs)), nx)
        iy0 = max(int(math.floor((poly[:, 1].min() - y0) / cs)), 0)
        iy1 = min(int(math.ceil((poly[:, 1].max() - y0) / cs)), ny)
        if ix1 <= ix0 or iy1 <= iy0:
            continue
        gx, gy = np.meshgrid(x0 + (np.arange(ix0, ix1) + 0.5) * cs, y0 + (np.arange(iy0, iy1) + 0.5) * cs)
        q = np.stack((gx.ravel(), gy.ravel()), -1)
        inside = points_in_polygon(q, poly)
        if not inside.any():
            continue
        qi = q[inside]
        c = ll.center
        a, b = c[:-1], c[1:]
        e = b - a
        l2 = np.maximum(np.sum(e * e, axis=1), 1e-300)
        t = np.clip(np.sum((qi[:, None, :] - a[None]) * e[None], axis=2) / l2[None], 0.0, 1.0)
        proj = a[None] + t[..., None] * e[None]
        k = np.argmin(np.sum((qi[:, None, :] - proj) ** 2, axis=2), axis=1)
        yaw = np.arctan2(e[k, 1], e[k, 0])
        sub = out[iy0:iy1, ix0:ix1].ravel()
        idx = np.nonzero(inside)[0]
        free = np.isnan(sub[idx])       # first lanelet in list order wins where lanelets overlap
        sub[idx[free]] = yaw[free]
        out[iy0:iy1, ix0:ix1] = sub.reshape(iy1 - iy0, ix1 - ix0)
    return out


def spatial_order(edges, bits=16):
    """Boundary pieces sorted along a Z-order (Morton) curve of their midpoints: 64 consecutive pieces then cover a
    small patch of the map, which is what lets the ray and settle kernels skip whole chunks by bounding box.  The
    order only affects speed and the tie-break between pieces hit at exactly the same range (lower index wins)."""
    edges = np.asarray(edges, dtype=np.float64).reshape(-1, 4)
    if len(edges) < 2:
        return edges
    mid = 0.5 * (edges[:, :2] + edges[:, 2:])
    lo, hi = mid.min(axis=0), mid.max(axis=0)
    span = np.maximum(hi - lo, 1e-9)
    q = np.minimum(((mid - lo) / span * (2 ** bits - 1)).astype(np.uint64), 2 ** bits - 1)

    def spread(v):
        v = v & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v

    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))
    return edges[np.argsort(key, kind="stable")]


def _crossings(q, edges):
    """number of boundary pieces a horizontal ray from q to +x crosses (half-open in y, like the raster rule)"""
    ay, by = edges[:, 1], edges[:, 3]
    cond = (ay > q[1]) != (by > q[1])
    if not cond.any():
        return 0
    e = edges[cond]
    xc = e[:, 0] + (q[1] - e[:, 1]) * (e[:, 2] - e[:, 0]) / (e[:, 3] - e[:, 1])
    return int(np.count_nonzero(q[0] < xc))


def boundary_rings(edges, tol=1e-6):
    """Groups the boundary pieces into the connected rings of the road union's boundary and tells interior rings
    (holes) from exterior ones.  Returns (ring [E] int32, is_hole [n_rings] bool).

    Pieces are linked through shared end points (matched within `tol` on four staggered grids, so two points closer
    than tol / 2 always meet in one of them); a ring is a hole iff an odd number of the other rings enclose it
    (crossing number of one of its midpoints against the other ring's pieces).  The reference walks
    `visible_area.exterior` only (sensor_model.py:126-131): a hole of road ∩ footprint casts no shadow there."""
    E = len(edges)
    if E == 0:
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=bool)
    parent = np.arange(E)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    pts = np.concatenate((edges[:, :2], edges[:, 2:]), axis=0)          # point k belongs to edge k % E
    for ox in (0.0, 0.5):
        for oy in (0.0, 0.5):
            kx = np.floor(pts[:, 0] / tol + ox).astype(np.int64)
            ky = np.floor(pts[:, 1] / tol + oy).astype(np.int64)
            order = np.lexsort((ky, kx))
            same = (kx[order][1:] == kx[order][:-1]) & (ky[order][1:] == ky[order][:-1])
            for i in np.nonzero(same)[0]:
                a, b = find(order[i] % E), find(order[i + 1] % E)
                if a != b:
                    parent[b] = a
    roots = np.array([find(i) for i in range(E)])
    uniq, ring = np.unique(roots, return_inverse=True)
    ring = ring.astype(np.int32)
    n = len(uniq)
    is_hole = np.zeros(n, dtype=bool)
    if n > 1:
        members = [np.nonzero(ring == k)[0] for k in range(n)]
        boxes = np.array([[min(edges[m, 0].min(), edges[m, 2].min()), min(edges[m, 1].min(), edges[m, 3].min()),
                           max(edges[m, 0].max(), edges[m, 2].max()), max(edges[m, 1].max(), edges[m, 3].max())]
                          for m in members])
        for k in range(n):
            e = edges[members[k][0]]
            q = 0.5 * (e[:2] + e[2:])
            inside = 0
            for j in range(n):
                if j == k or q[0] < boxes[j, 0] or q[0] > boxes[j, 2] or q[1] < boxes[j, 1] or q[1] > boxes[j, 3]:
                    continue
                inside += _crossings(q, edges[members[j]]) & 1
            is_hole[k] = (inside & 1) == 1
    return ring, is_hole


def edge_lines(edges, tol=1e-6, sin_tol=1e-9):
    """Straight-line chains of the boundary pieces: two pieces get the same label when they share an end point that
    no third piece touches and one continues the other in a straight line (|sin| of the angle between them <=
    sin_tol).  Lanelet bounds are sampled polylines, so a straight kerb is dozens of collinear pieces; two rays that
    stop on the same chain see one occluder, and the chord between their hit points lies on it."""
    E = len(edges)
    parent = np.arange(E)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    if E == 0:
        return np.zeros(0, dtype=np.int32)
    pts = np.concatenate((edges[:, :2], edges[:, 2:]), axis=0)          # point k: end (k // E) of edge k % E
    d = edges[:, 2:] - edges[:, :2]
    ln = np.maximum(np.hypot(d[:, 0], d[:, 1]), 1e-300)
    # end points that coincide (rounded to `tol`; a pair split by the rounding only costs a missed merge)
    key = np.round(pts / tol).astype(np.int64)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    order = np.argsort(inv, kind="stable")
    starts = np.concatenate(([0], np.cumsum(cnt)[:-1]))
    two = np.nonzero(cnt == 2)[0]                                       # a free end or a junction stops a chain
    pa, pb = order[starts[two]], order[starts[two] + 1]
    a, b = pa % E, pb % E
    cross = d[a, 0] * d[b, 1] - d[a, 1] * d[b, 0]
    da = np.where((pa >= E)[:, None], d[a], -d[a])                      # direction pointing INTO the shared point
    db = np.where((pb >= E)[:, None], -d[b], d[b])                      # direction pointing OUT OF the shared point
    ok = (a != b) & (np.abs(cross) <= sin_tol * ln[a] * ln[b]) & ((da * db).sum(axis=1) > 0.0)
    for i, j in zip(a[ok], b[ok]):
        ri, rj = find(int(i)), find(int(j))
        if ri != rj:
            parent[rj] = ri
    lab = np.array([find(i) for i in range(E)])
    _, out = np.unique(lab, return_inverse=True)
    return out.astype(np.int32)


@dataclass
class MapGeometry:
    poly_off: np.ndarray     # int32 [P+1]
    poly_xy: np.ndarray      # [V,2]
    edges: np.ndarray        # [E,4]
    edge_ring: Optional[np.ndarray] = None   # int32 [E]: connected ring of the union boundary each piece lies on
    ring_is_hole: Optional[np.ndarray] = None  # bool [n_rings]
    edge_line: Optional[np.ndarray] = None   # int32 [E]: straight-line chain each piece belongs to (edge_lines)

    def __post_init__(self):
        e = np.asarray(self.edges, dtype=np.float64).reshape(-1, 4)
        if self.edge_ring is None or self.ring_is_hole is None:
            self.edge_ring, self.ring_is_hole = boundary_rings(e)
        if self.edge_line is None:
            self.edge_line = edge_lines(e)

    @classmethod
    def from_lanelets(cls, lanelets):
        polys = [ll.polygon for ll in lanelets]
        off = np.zeros(len(polys) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(p) for p in polys])
        return cls(off, np.concatenate(polys, axis=0), spatial_order(union_boundary_edges(polys)))

    def holes(self):
        """per interior ring: (ring id, its piece end points [n,2])"""
        out = []
        for k in np.nonzero(self.ring_is_hole)[0]:
            e = self.edges[self.edge_ring == k]
            out.append((int(k), np.concatenate((e[:, :2], e[:, 2:]), axis=0)))
        return out


# ------------------------------------------------------------------------------------------------ synthetic city
def synthetic_urban_grid(n_blocks=8, block=60.0, lane_w=3.5, lanes_per_dir=2, ds=2.0, n_parked=64, seed=20240134):
    """BASELINE config 3: Manhattan grid, n_blocks x n_blocks blocks of `block` m, `lanes_per_dir` lanes per direction;
    bounds sampled every `ds` m => O(10^4) boundary edges.  Parked cars (4.5 x 1.8 m, static) on the outer lanes."""
    rng = np.random.default_rng(seed)
    W = 2 * lanes_per_dir * lane_w
    pitch = block + W
    lanelets, lid = [], 1
    n_lines = n_blocks + 1

    def straight(p0, p1, off_l, off_r):
        d = p1 - p0
        L = np.linalg.norm(d)
        u = d / L
        nvec = np.array([-u[1], u[0]])
        k = max(int(round(L / ds)), 1)
        s = np.linspace(0.0, L, k + 1)[:, None]
        base = p0[None] + s * u[None]
        return base + off_l * nvec[None], base + off_r * nvec[None]

    for axis in (0, 1):
        for line in range(n_lines):
            c = line * pitch
            for seg in range(-1, 2 * n_lines - 1):  # even: intersection square, odd: street between squares
                if seg % 2 == 0:
                    a0 = (seg // 2) * pitch - W / 2
                    a1 = a0 + W
                else:
                    a0 = ((seg - 1) // 2) * pitch + W / 2
                    a1 = a0 + block
                if seg == -1:
                    continue
                for lane in range(2 * lanes_per_dir):
                    off_hi = W / 2 - lane * lane_w
                    off_lo = off_hi - lane_w
                    fwd = lane >= lanes_per_dir
                    if axis == 0:
                        p0, p1 = np.array([a0, c]), np.array([a1, c])
                    else:
                        p0, p1 = np.array([c, a0]), np.array([c, a1])
                    if not fwd:
                        p0, p1 = p1, p0
                        left, right = straight(p0, p1, -off_lo, -off_hi)
                    else:
                        left, right = straight(p0, p1, off_hi, off_lo)
                    lanelets.append(Lanelet(lid, left, right))
                    lid += 1
    # parked cars near the central intersection, on the outermost lanes
    mid = (n_lines // 2) * pitch
    obstacles = []
    for i in range(n_parked):
        axis = i % 2
        side = 1 if (i // 2) % 2 else -1
        along = mid + rng.uniform(-2.2 * pitch, 2.2 * pitch)
        k = round((along - mid) / pitch)
        if abs(along - (mid + k * pitch)) < W / 2 + 4.0:   # keep the intersections free
            along += math.copysign(W / 2 + 6.0, along - (mid + k * pitch) if along != mid + k * pitch else 1.0)
        line_c = mid + rng.integers(-1, 2) * pitch
        lat = line_c + side * (W / 2 - 1.0)
        pos = np.array([along, lat]) if axis == 0 else np.array([lat, along])
        yaw = 0.0 if axis == 0 else math.pi / 2
        obstacles.append(Obstacle(1000 + i, "static", "parkedVehicle", 4.5, 1.8, 0,
                                  np.array([pos[0], pos[1], yaw, 0.0]), np.zeros((0, 4))))
    ego = np.array([mid - 20.0, mid - lane_w / 2, 0.0, 8.0])
    return Scenario(0.1, lanelets, obstacles, [], ego, "synthetic_urban_grid")


# ------------------------------------------------------------------------------------------------ duck-typed CommonRoad
def _cr_type(t):
    v = getattr(t, "value", t)
    return str(v)


def obstacle_from_commonroad(ob) -> Obstacle:
    """commonroad-io obstacle (duck-typed: .obstacle_id, .obstacle_type, .obstacle_role, .obstacle_shape.length/width,
    .initial_state.{position, orientation, velocity, time_step}, .prediction.trajectory.state_list) -> Obstacle.
    Only the attributes fo_obstacle.py:60-116 reads are touched."""
    st = ob.initial_state
    init = np.array([st.position[0], st.position[1], getattr(st, "orientation", 0.0), getattr(st, "velocity", 0.0)],
                    dtype=np.float64)
    states = []
    pred = getattr(ob, "prediction", None)
    if pred is not None and getattr(pred, "trajectory", None) is not None:
        for s in pred.trajectory.state_list:
            states.append([s.position[0], s.position[1], getattr(s, "orientation", 0.0), getattr(s, "velocity", 0.0)])
    role = _cr_type(getattr(ob, "obstacle_role", "dynamic")).lower()
    role = "static" if "static" in role else "dynamic"
    return Obstacle(int(ob.obstacle_id), role, _cr_type(ob.obstacle_type), float(ob.obstacle_shape.length),
                    float(ob.obstacle_shape.width), int(getattr(st, "time_step", 0)), init,
                    np.array(states, dtype=np.float64).reshape(-1, 4))


def normalize_intersections(inters) -> List[dict]:
    """intersections as the rule families read them (spawn_locator.py:171-195): a list of
    ``{'incomings': [{'incoming': ids, 'right': ids, 'straight': ids, 'left': ids}, ...]}`` -- from this package's own
    dicts or from duck-typed CommonRoad ``Intersection`` objects (``incomings`` with ``incoming_lanelets`` /
    ``successors_right`` / ``successors_straight`` / ``successors_left`` [ext])"""
    out = []
    for it in inters or []:
        if isinstance(it, dict):
            out.append(it)
            continue
        inc = []
        for e in getattr(it, "incomings", []):
            inc.append({"incoming": list(getattr(e, "incoming_lanelets", []) or []),
                        "right": list(getattr(e, "successors_right", []) or []),
                        "straight": list(getattr(e, "successors_straight", []) or []),
                        "left": list(getattr(e, "successors_left", []) or [])})
        out.append({"incomings": inc})
    return out


def lanelets_of(net) -> List[Lanelet]:
    """list[Lanelet] | Scenario | duck-typed CommonRoad LaneletNetwork (.lanelets with .left_vertices/.right_vertices)"""
    if isinstance(net, Scenario):
        return net.lanelets
    items = net.lanelets if hasattr(net, "lanelets") else list(net)
    out = []
    for ll in items:
        if isinstance(ll, Lanelet):
            out.append(ll)
            continue
        new = Lanelet(int(ll.lanelet_id), np.asarray(ll.left_vertices, dtype=np.float64),
                      np.asarray(ll.right_vertices, dtype=np.float64))
        new.successors = list(getattr(ll, "successor", []) or [])
        new.predecessors = list(getattr(ll, "predecessor", []) or [])
        new.adj_left = getattr(ll, "adj_left", None)
        new.adj_left_same_direction = getattr(ll, "adj_left_same_direction", None)
        new.adj_right = getattr(ll, "adj_right", None)
        new.adj_right_same_direction = getattr(ll, "adj_right_same_direction", None)
        out.append(new)
    return out


class LaneletNetworkView:
    """what ``scenario.lanelet_network`` is for a :class:`Scenario` (keeps interface.py:75 attribute access working)"""

    def __init__(self, lanelets):
        self.lanelets = lanelets

    @property
    def lanelet_polygons(self):
        return [ll.polygon for ll in self.lanelets]


Scenario.lanelet_network = property(lambda self: LaneletNetworkView(self.lanelets))


def _scenario_add_objects(self, objs):
    """Scenario.add_objects of commonroad (agent.py:250 appends scripted real agents)"""
    for o in (objs if isinstance(objs, (list, tuple)) else [objs]):
        self.obstacles.append(o if isinstance(o, Obstacle) else obstacle_from_commonroad(o))


Scenario.add_objects = _scenario_add_objects


def load_geometry_npz(path) -> Scenario:
    """Scenario from the compact array form (what tests/golden/gen_scenario_fixture.py writes)"""
    g = np.load(path, allow_pickle=False)
    off = g["lanelet_off"]
    lanelets = []
    for i, lid in enumerate(g["lanelet_id"]):
        ll = Lanelet(int(lid), g["lanelet_left"][off[i]:off[i + 1]], g["lanelet_right"][off[i]:off[i + 1]])
        ll.successors = [int(q) for q in g["lanelet_successors"][i] if q >= 0]
        ll.predecessors = [int(q) for q in g["lanelet_predecessors"][i] if q >= 0]
        a = g["lanelet_adjacent"][i]
        if a[0] >= 0:
            ll.adj_left, ll.adj_left_same_direction = int(a[0]), bool(a[1])
        if a[2] >= 0:
            ll.adj_right, ll.adj_right_same_direction = int(a[2]), bool(a[3])
        lanelets.append(ll)
    so = g["obstacle_state_off"]
    obstacles = []
    for i, oid in enumerate(g["obstacle_id"]):
        obstacles.append(Obstacle(int(oid), str(g["obstacle_role"][i]), str(g["obstacle_type"][i]),
                                  float(g["obstacle_dims"][i, 0]), float(g["obstacle_dims"][i, 1]),
                                  int(g["obstacle_t0"][i]), g["obstacle_initial"][i],
                                  g["obstacle_states"][so[i]:so[i + 1]]))
    inters = {}
    names = {0: "incoming", 1: "right", 2: "straight", 3: "left"}
    for iid, k, code, lid in g["intersection_rows"]:
        it = inters.setdefault(int(iid), {})
        inc = it.setdefault(int(k), {"incoming": [], "right": [], "straight": [], "left": []})
        inc[names[int(code)]].append(int(lid))
    ilist = [{"id": iid, "incomings": [v[k] for k in sorted(v)]} for iid, v in inters.items()]
    return Scenario(float(g["dt"]), lanelets, obstacles, ilist, g["ego_initial"], str(g["benchmark_id"]))


# ------------------------------------------------------------------------------------------------ phantom vehicle routes
def enumerate_routes(lanelets, max_depth=2) -> Dict[int, List[List[int]]]:
    """candidate routes per start lanelet: depth-first over successors and over same-direction neighbours that have
    successors, down to `max_depth` hops (route_planner.py:54-90)"""
    by = {ll.lanelet_id: ll for ll in lanelets}
    out = {}

    def explore(cur, route, routes, depth):
        ll = by[cur]
        route.append(cur)
        nxt = [s for s in ll.successors if s in by]
        for adj, same in ((ll.adj_right, ll.adj_right_same_direction), (ll.adj_left, ll.adj_left_same_direction)):
            if adj is not None and same and adj in by and by[adj].successors:
                nxt.append(adj)
        if depth >= max_depth:
            nxt = []
        if not nxt:
            routes.append(list(route))
            return
        for s in nxt:
            explore(s, route, routes, depth + 1)
            route.pop()

    for ll in lanelets:
        routes = []
        explore(ll.lanelet_id, [], routes, 0)
        out[ll.lanelet_id] = routes
    return out


def route_polyline(lanelets_by_id, route):
    """reference path of a route: centre lines joined end to start; a lanelet that is left sideways (its follower is a
    neighbour, not a successor) contributes nothing -- the vehicle keeps its lateral offset to the neighbour's centre
    line, which is what the reference's min-var(v) Frenet sample does (agent.py:349-379, d1 = d0)"""
    parts = []
    for i, lid in enumerate(route):
        ll = lanelets_by_id[lid]
        if i + 1 < len(route) and route[i + 1] not in ll.successors:
            continue
        c = ll.center
        if parts and np.linalg.norm(c[0] - parts[-1][-1]) < 1e-2:
            c = c[1:]
        if len(c):
            parts.append(c)
    if not parts:
        parts = [lanelets_by_id[route[-1]].center]
    p = np.concatenate(parts, axis=0)
    keep = np.concatenate(([True], np.hypot(np.diff(p[:, 0]), np.diff(p[:, 1])) > 0.0))
    return p[keep]


@dataclass
class RouteTable:
    """flat device form: routes r < R of lanelet index p occupy vertices first[p*R+r] .. +count[p*R+r] of xy / s"""
    R: int
    first: np.ndarray    # int32 [P*R]
    count: np.ndarray    # int32 [P*R]  (0 = no such route)
    xy: np.ndarray       # [NV,2]
    s: np.ndarray        # [NV] arc length from the route's first vertex

    @classmethod
    def from_lanelets(cls, lanelets, R=3, max_depth=2):
        by = {ll.lanelet_id: ll for ll in lanelets}
        routes = enumerate_routes(lanelets, max_depth)
        first = np.zeros(len(lanelets) * R, dtype=np.int32)
        count = np.zeros(len(lanelets) * R, dtype=np.int32)
        xy, ss, nv = [], [], 0
        for p, ll in enumerate(lanelets):
            polys, seen = [], set()
            for rt in routes[ll.lanelet_id]:
                poly = route_polyline(by, rt)
                key = poly.tobytes()
                if len(poly) >= 2 and key not in seen:     # lane-change variants can collapse onto the same polyline
                    seen.add(key)
                    polys.append(poly)
            for r, poly in enumerate(polys[:R]):
                first[p * R + r], count[p * R + r] = nv, len(poly)
                xy.append(poly)
                ss.append(np.concatenate(([0.0], np.cumsum(np.hypot(np.diff(poly[:, 0]), np.diff(poly[:, 1]))))))
                nv += len(poly)
        return cls(R, first, count, np.concatenate(xy) if xy else np.zeros((0, 2)),
                   np.concatenate(ss) if ss else np.zeros(0))


def lanelet_index_raster(lanelets, x0, y0, cs, nx, ny):
    """per raster cell: list index of the first lanelet containing the cell centre (same rule as lane_yaw_raster);
    -1 off-lane"""
    out = np.full((ny, nx), -1, dtype=np.int32)
    for p, ll in enumerate(lanelets):
        poly = ll.polygon
        ix0 = max(int(math.floor((poly[:, 0].min() - x0) / cs)), 0)
        ix1 = min(int(math.ceil((poly[:, 0].max() - x0) / cs)), nx)
        iy0 = max(int(math.floor((poly[:, 1].min() - y0) / cs)), 0)
        iy1 = min(int(math.ceil((poly[:, 1].max() - y0) / cs)), ny)
        if ix1 <= ix0 or iy1 <= iy0:
            continue
        gx, gy = np.meshgrid(x0 + (np.arange(ix0, ix1) + 0.5) * cs, y0 + (np.arange(iy0, iy1) + 0.5) * cs)
        inside = points_in_polygon(np.stack((gx.ravel(), gy.ravel()), -1), poly).reshape(iy1 - iy0, ix1 - ix0)
        sub = out[iy0:iy1, ix0:ix1]
        sub[inside & (sub < 0)] = p
    return out
