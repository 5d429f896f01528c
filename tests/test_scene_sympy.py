"""Primitives of the scene stage pinned to exact rational arithmetic from sympy.geometry (an independent third-party
package; GEOS, which the reference uses through shapely, is absent here):
  * first hit of a ray among segments (oracle/fo_oracle_scene.c `fo_oracle_raycast`, the restatement the HIP ray fan is
    compared with bit for bit) against `Ray.intersection(Segment)`;
  * the road raster (cell centre inside any lanelet polygon; ref sensor_model.py:195-199 builds the union) against
    `Polygon.encloses_point` on the scenario-1 lanelet polygons."""
import math
import os

import numpy as np
import pytest

sympy = pytest.importorskip("sympy")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_first_hit_of_a_ray_matches_exact_intersections(oracle):
    from sympy import Point, Rational, Ray, Segment
    rng = np.random.default_rng(7)
    R = lambda v: Rational(float(v))
    for case in range(1):
        ego = rng.uniform(-1.0, 1.0, 2)
        edges = []
        while len(edges) < 10:
            a = rng.uniform(-12.0, 12.0, 2)
            b = a + rng.uniform(-6.0, 6.0, 2)
            if min(np.hypot(*(a - ego)), np.hypot(*(b - ego))) > 1.5:
                edges.append([a[0], a[1], b[0], b[1]])
        edges = np.array(edges)
        n_rays, r = 12, 15.0
        dirs = oracle.ray_dirs(n_rays, ego_yaw=0.3 * case, fov_deg=360.0)
        rng_o, hid_o, _ = oracle.raycast(edges, np.zeros((0, 8)), np.zeros(0, np.uint8), ego, dirs, r)
        segs = [Segment(Point(R(e[0]), R(e[1])), Point(R(e[2]), R(e[3]))) for e in edges]
        o = Point(R(ego[0]), R(ego[1]))
        for i in range(n_rays):
            ray = Ray(o, Point(R(ego[0]) + R(dirs[i, 0]), R(ego[1]) + R(dirs[i, 1])))
            # the direction vectors are unit length only to rounding: range = |hit - ego| / |d|, like t in the kernels
            dn = sympy.sqrt(R(dirs[i, 0]) ** 2 + R(dirs[i, 1]) ** 2)
            best, who = None, -1
            for j, sg in enumerate(segs):
                hit = ray.intersection(sg)
                if not hit:
                    continue
                h = hit[0]
                if isinstance(h, Segment):      # collinear overlap: cannot happen with random data
                    continue
                t = o.distance(h) / dn
                if best is None or t < best:
                    best, who = t, j
            if best is None or float(best) > r:
                assert hid_o[i] == -1 and rng_o[i] == pytest.approx(r, abs=1e-12), (case, i)
            else:
                assert hid_o[i] == who, (case, i)
                assert rng_o[i] == pytest.approx(float(best), abs=1e-10), (case, i)


def test_road_raster_matches_exact_point_in_polygon(oracle):
    from sympy import Point, Polygon, Rational
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    geo = S.MapGeometry.from_lanelets(sc.lanelets)
    cs = 0.5
    x0 = math.floor(geo.poly_xy[:, 0].min() / cs) * cs - 1.0
    y0 = math.floor(geo.poly_xy[:, 1].min() / cs) * cs - 1.0
    nx = int(math.ceil((geo.poly_xy[:, 0].max() + 1.0 - x0) / cs))
    ny = int(math.ceil((geo.poly_xy[:, 1].max() + 1.0 - y0) / cs))
    raster = oracle.road_raster(geo.poly_off, geo.poly_xy, x0, y0, cs, nx, ny)
    R = lambda v: Rational(float(v))
    polys = [Polygon(*[Point(R(x), R(y)) for x, y in geo.poly_xy[geo.poly_off[p]:geo.poly_off[p + 1]]])
             for p in range(len(geo.poly_off) - 1)]
    # cells next to the road's border are the informative ones: a road cell with a non-road 4-neighbour and vice versa
    rd = raster.astype(bool)
    edge = np.zeros_like(rd)
    edge[1:-1, 1:-1] = (rd[1:-1, 1:-1] != rd[:-2, 1:-1]) | (rd[1:-1, 1:-1] != rd[2:, 1:-1]) | \
                       (rd[1:-1, 1:-1] != rd[1:-1, :-2]) | (rd[1:-1, 1:-1] != rd[1:-1, 2:])
    iy, ix = np.nonzero(edge)
    pick = np.random.default_rng(3).choice(len(ix), 36, replace=False)
    n_in = 0
    for k in pick:
        c = Point(R(x0 + (ix[k] + 0.5) * cs), R(y0 + (iy[k] + 0.5) * cs))
        on_border = any(c in seg for pg in polys for seg in pg.sides)
        if on_border:       # a centre exactly on a polygon side: the crossing-number convention decides, not geometry
            continue
        inside = any(pg.encloses_point(c) for pg in polys)
        assert bool(raster[iy[k], ix[k]]) == inside, (ix[k], iy[k])
        n_in += inside
    assert 6 < n_in < 30
