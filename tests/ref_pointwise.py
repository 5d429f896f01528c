"""Pointwise restatement of the reference sensor model's polygon set algebra (test infrastructure only).

The reference builds its visible / occluded areas with GEOS polygon booleans (shapely 2.0.2, absent here).  Set
membership of a single point needs no polygon boolean, so the same sets can be evaluated at the cell centres with
numpy alone:

  visible(p)  = p in road  and  p in footprint
                and p in no shadow quad [v1, v2, v2 + 100 (v2 - ego), v1 + 100 (v1 - ego)] of a consecutive exterior
                vertex pair of road ∩ footprint                      (ref sensor_model.py:103-157, helper_functions.py:79-96)
                and p not in any present non-bicycle obstacle inflated by 5 mm, nor in its occlusion quad
                [c1, c2, c2 + 100 u(c2 - ego), c1 + 100 u(c1 - ego)]   (ref sensor_model.py:159-193, helper_functions.py:133-176)
  occluded(p) = p in the 100-point half fan of radius 1.5 r about the heading  and  p in road  and not visible(p)
                                                                      (ref sensor_model.py:82-93, 201-209)

footprint = shapely's `Point.buffer(r)` (regular 64-gon inscribed in the circle, first vertex at angle 0) for a full
circle sensor, else the 100-point fan of `_calc_relevant_sector`.  Exterior vertex pairs of road ∩ footprint that lie
on the footprint's own boundary cast shadows that fall outside the footprint, so only pieces of the road union's
boundary inside the footprint matter.  The reference walks exterior rings only: a hole of the road union
that the footprint encloses is an interior ring of road ∩ footprint and casts no shadow; a hole the footprint cuts
open lies on the exterior ring of road ∩ footprint and does.  (Not restated: the bogus vertex pair the reference
forms between the last vertex of one polygon and the first of the next when road ∩ footprint is a MultiPolygon,
sensor_model.py:126-127.)

This is what tests/test_scene_pointwise.py compares the ray-fan/cell discretisation with: the two can differ only in
cells whose centre lies within the discretisation error (one ray spacing / the chord sagitta / hole shadows) of a
shadow or range boundary."""
import math

import numpy as np


def points_in_polygon(q, poly):
    """Crossing-number test, vectorised over q [N,2]; poly [V,2] open ring."""
    x, y = q[:, 0], q[:, 1]
    inside = np.zeros(len(q), bool)
    px, py = poly[:, 0], poly[:, 1]
    n = len(poly)
    for i in range(n):
        j = (i + 1) % n
        yi, yj = py[i], py[j]
        if yi == yj:
            continue
        cond = (yi > y) != (yj > y)
        xc = px[i] + (y - yi) * (px[j] - px[i]) / (yj - yi)
        inside ^= cond & (x < xc)
    return inside


def ring_labels(edges, tol=1e-6):
    """(ring label per piece, is_hole per ring).  Rings = connected components of the end-point graph (scipy), hole =
    enclosed by an odd number of other rings (crossing number of a piece midpoint against the other ring's pieces)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    E = len(edges)
    pts = np.concatenate((edges[:, :2], edges[:, 2:]), 0)
    pairs = cKDTree(pts).query_pairs(tol, output_type="ndarray")
    i = np.concatenate((np.arange(E), pairs[:, 0] % E))
    j = np.concatenate((np.arange(E), pairs[:, 1] % E))
    n, lab = connected_components(coo_matrix((np.ones(len(i)), (i, j)), shape=(E, E)), directed=False)
    hole = np.zeros(n, bool)
    for k in range(n):
        mem = np.nonzero(lab == k)[0]
        e = edges[mem[0]]
        q = 0.5 * (e[:2] + e[2:])
        enclosed = 0
        for m in range(n):
            if m == k:
                continue
            o = edges[lab == m]
            cond = (o[:, 1] > q[1]) != (o[:, 3] > q[1])
            oc = o[cond]
            xc = oc[:, 0] + (q[1] - oc[:, 1]) * (oc[:, 2] - oc[:, 0]) / (oc[:, 3] - oc[:, 1])
            enclosed += int(np.count_nonzero(q[0] < xc)) & 1
        hole[k] = bool(enclosed & 1)
    return lab, hole


def exterior_edge_mask(edges):
    lab, hole = ring_labels(edges)
    return ~hole[lab]


def footprint_polygon(ego, r, yaw, fov_deg):
    if fov_deg >= 359.9:
        # shapely Point.buffer(r): quad_segs = 16 -> 64 segments, clockwise from angle 0 (orientation irrelevant here)
        ang = -np.arange(64) * (2.0 * math.pi / 64.0)
        return np.stack((ego[0] + r * np.cos(ang), ego[1] + r * np.sin(ang)), -1)
    return sector_polygon(ego, r, yaw - math.radians(fov_deg / 2), yaw + math.radians(fov_deg / 2))


def sector_polygon(ego, radius, a0, a1):
    ang = np.linspace(a0, a1, 100)
    pts = np.stack((ego[0] + radius * np.cos(ang), ego[1] + radius * np.sin(ang)), -1)
    return np.concatenate((np.asarray(ego, float)[None], pts), 0)


def _clip_segments_to_polygon(edges, poly, convex):
    """Pieces of the segments inside `poly`.  Convex polygons: Cyrus-Beck.  Otherwise: split at every crossing with
    the polygon boundary and keep the sub-pieces whose midpoint is inside."""
    out = []
    n = len(poly)
    for e in edges:
        p0, p1 = e[:2], e[2:]
        d = p1 - p0
        ts = [0.0, 1.0]
        for i in range(n):
            a, b = poly[i], poly[(i + 1) % n]
            f = b - a
            den = d[0] * f[1] - d[1] * f[0]
            if den == 0.0:
                continue
            w = a - p0
            t = (w[0] * f[1] - w[1] * f[0]) / den
            u = (w[0] * d[1] - w[1] * d[0]) / den
            if 0.0 < t < 1.0 and 0.0 <= u <= 1.0:
                ts.append(t)
        ts = sorted(set(ts))
        mids = np.array([p0 + 0.5 * (ta + tb) * d for ta, tb in zip(ts[:-1], ts[1:])])
        keep = points_in_polygon(mids, poly)
        for (ta, tb), kp in zip(zip(ts[:-1], ts[1:]), keep):
            if kp and tb - ta > 0:
                out.append(np.concatenate((p0 + ta * d, p0 + tb * d)))
    return np.array(out).reshape(-1, 4)


def _in_quads(q, quads):
    """q [N,2], quads [K,4,2] (convex, any orientation) -> [N] True if inside any quad."""
    hit = np.zeros(len(q), bool)
    for quad in quads:
        area2 = 0.0
        for i in range(4):
            a, b = quad[i], quad[(i + 1) % 4]
            area2 += a[0] * b[1] - a[1] * b[0]
        if abs(area2) < 1e-12:
            continue                                        # degenerate: shapely reports it invalid, the ref skips it
        sgn = 1.0 if area2 > 0 else -1.0
        inside = np.ones(len(q), bool)
        for i in range(4):
            a, b = quad[i], quad[(i + 1) % 4]
            cr = (b[0] - a[0]) * (q[:, 1] - a[1]) - (b[1] - a[1]) * (q[:, 0] - a[0])
            inside &= sgn * cr > 0.0
        hit |= inside
    return hit


def _unit(v):
    return v / np.linalg.norm(v)


def shadow_quad(v1, v2, ego):
    """ref helper_functions.py:79-96 (create_polygon_from_vertices): the area a boundary vertex pair hides from the ego --
    [v1, v2, v2 + 100 (v2 - ego), v1 + 100 (v1 - ego)]; pinned to the reference's own function by tests/golden/shadow_geometry.npz"""
    v1, v2, ego = (np.asarray(q, float) for q in (v1, v2, ego))
    return np.array([v1, v2, v2 + 100 * (v2 - ego), v1 + 100 * (v1 - ego)])


def obstacle_wedge(ego, corners):
    """ref helper_functions.py:139-176 (get_polygon_from_obstacle_occlusion): (wedge [c1, c2, c2 + 100 u(c2 - ego),
    c1 + 100 u(c1 - ego)], c1, c2) with (c1, c2) the corner pair that subtends the largest angle at the ego; pinned like
    shadow_quad"""
    ego = np.asarray(ego, float)
    c1, c2 = _projection_points(ego, np.asarray(corners, float).reshape(4, 2))
    return np.array([c1, c2, c2 + _unit(c2 - ego) * 100, c1 + _unit(c1 - ego) * 100]), c1, c2


def _projection_points(ego, corners):
    """ref helper_functions.py:143-166: the corner pair subtending the largest angle at the ego."""
    best, ret = 0.0, (corners[0], corners[0])
    for p1 in corners:
        for p2 in corners:
            u1, u2 = _unit(p1 - ego), _unit(p2 - ego)
            ang = math.acos(min(1.0, max(-1.0, float(np.dot(u1, u2)))))
            if ang > best:
                best, ret = ang, (p1, p2)
    return ret


def classify(q, lanelet_polys, boundary_edges, ego, yaw, r, fov_deg=360.0, obstacle_corners=(), obstacle_is_bicycle=()):
    """Returns (road, visible, occluded) boolean arrays for the points q [N,2]."""
    ego = np.asarray(ego, float)
    road = np.zeros(len(q), bool)
    for p in lanelet_polys:
        box = np.nonzero((q[:, 0] >= p[:, 0].min()) & (q[:, 0] <= p[:, 0].max()) &
                         (q[:, 1] >= p[:, 1].min()) & (q[:, 1] <= p[:, 1].max()))[0]
        if len(box):
            road[box] |= points_in_polygon(q[box], p)
    foot = footprint_polygon(ego, r, yaw, fov_deg)
    vis = road & points_in_polygon(q, foot)
    # rings of road ∩ footprint: a hole of the road union stays a hole (no shadow) only if the footprint encloses it
    lab, hole = ring_labels(boundary_edges)
    casts = np.ones(len(boundary_edges), bool)
    for k in np.nonzero(hole)[0]:
        e = boundary_edges[lab == k]
        if points_in_polygon(np.concatenate((e[:, :2], e[:, 2:]), 0), foot).all():
            casts[lab == k] = False
    pieces = _clip_segments_to_polygon(boundary_edges[casts], foot, convex=fov_deg >= 359.9)
    quads = np.array([shadow_quad(e[:2], e[2:], ego) for e in pieces]).reshape(-1, 4, 2)
    cand = np.nonzero(vis)[0]                               # only road cells inside the footprint can lose visibility
    vis[cand] &= ~_in_quads(q[cand], quads)
    for corn, bike in zip(obstacle_corners, obstacle_is_bicycle):
        if bike:
            continue
        corn = np.asarray(corn, float).reshape(4, 2)
        # obstacle inflated by 5 mm (mitre join): scale the rectangle about its centre along its own axes
        c = corn.mean(0)
        ax, ay = corn[1] - corn[0], corn[3] - corn[0]
        la, lb = np.linalg.norm(ax), np.linalg.norm(ay)
        infl = np.array([c + sx * (0.5 * la + 0.005) * ax / la + sy * (0.5 * lb + 0.005) * ay / lb
                         for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))])
        occl, _, _ = obstacle_wedge(ego, corn)
        cand = np.nonzero(vis)[0]
        vis[cand] &= ~_in_quads(q[cand], np.array([infl, occl]))
    half = sector_polygon(ego, 1.5 * r, yaw - math.radians(90), yaw + math.radians(90))
    occ = points_in_polygon(q, half) & road & ~vis
    return road, vis, occ


def obstacles_visible(lanelet_polys, boundary_edges, ego, yaw, r, fov_deg, corners, is_bicycle, ds=0.02, off=0.0075):
    """ref sensor_model.py:59-76: `obst.current_polygon.intersects(visible_area.buffer(0.01))`.  The visible area has the
    obstacle grown by 5 mm removed (:183), so the polygon touches the 1 cm buffer iff a point just outside that skin is
    visible: the obstacle's outline, pushed out by 7.5 mm and sampled every `ds`, is classified by `classify`."""
    out = []
    for c in corners:
        c = np.asarray(c, float).reshape(4, 2)
        pts = []
        for sd in range(4):
            a, b = c[sd], c[(sd + 1) % 4]
            e = b - a
            ln = np.linalg.norm(e)
            nrm = np.array([e[1], -e[0]]) / ln
            if np.dot(nrm, a - c.mean(0)) < 0:
                nrm = -nrm
            t = np.clip(np.arange(0.0, ln + ds, ds) / ln, 0.0, 1.0)
            pts.append(a[None] + t[:, None] * e[None] + off * nrm[None])
        _, vis, _ = classify(np.concatenate(pts), lanelet_polys, boundary_edges, ego, yaw, r, fov_deg, corners, is_bicycle)
        out.append(bool(vis.any()))
    return np.array(out)
