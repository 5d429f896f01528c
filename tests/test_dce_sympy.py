"""The rectangle distance behind DCE (ref: metrics/dce.py:79 -- shapely `Polygon.distance`, GEOS, absent here) pinned to
exact arithmetic from an independent third-party package that IS installed: sympy.geometry on rational coordinates.
Random commonroad-style rectangles (centre, yaw, length, width) are given to the oracle's quadrilateral distance and,
with every double converted exactly to a rational, to sympy: overlap by `Polygon.intersection` / `encloses_point`,
otherwise the minimum of the exact point-to-segment distances (`Segment.distance`) over all vertex / edge pairs -- the
distance of two disjoint convex polygons is attained at a vertex of one of them.  (sympy's own `Polygon.distance`, a
rotating-calipers routine, is not used: it returns 1.489 for a pair whose vertex-edge minimum is 1.250 and warns
about it.)"""
import math

import numpy as np
import pytest

sympy = pytest.importorskip("sympy")


def _sympy_distance(qa, qb):
    from sympy import Point, Polygon, Rational, Segment
    A = [Point(Rational(float(x)), Rational(float(y))) for x, y in qa]
    B = [Point(Rational(float(x)), Rational(float(y))) for x, y in qb]
    pa, pb = Polygon(*A), Polygon(*B)
    if pa.intersection(pb) or any(pa.encloses_point(v) for v in B) or any(pb.encloses_point(v) for v in A):
        return 0.0
    best = None
    for X, Y in ((A, B), (B, A)):
        for i in range(4):
            seg = Segment(Y[i], Y[(i + 1) % 4])
            for p in X:
                d = seg.distance(p)
                best = d if best is None or d < best else best
    return float(best)


@pytest.mark.parametrize("seed", range(3))
def test_rectangle_distance_matches_sympy(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    worst = 0.0
    n_zero = 0
    for it in range(14):
        la, wa = 4.508, 1.610                                     # ego (vehicle 2)
        lb, wb = rng.choice([(0.5, 0.5), (2.0, 0.9), (4.8, 2.0), (9.0, 2.5)])
        ca = rng.uniform(-3, 3, 2)
        gap = rng.choice([0.0, 0.3, 2.0, 7.0])                    # from overlapping to far apart
        ang = rng.uniform(0, 2 * math.pi)
        cb = ca + (gap + rng.uniform(0.0, 3.0)) * np.array([math.cos(ang), math.sin(ang)])
        ya, yb = rng.uniform(-math.pi, math.pi, 2)
        if it % 5 == 0:
            yb = ya + rng.choice([0.0, math.pi / 2])              # parallel / perpendicular: edge-edge configurations
        qa = oracle.rect_vertices(ca[0], ca[1], ya, la, wa).reshape(4, 2)
        qb = oracle.rect_vertices(cb[0], cb[1], yb, lb, wb).reshape(4, 2)
        got = oracle.quad_distance(qa, qb)
        ref = _sympy_distance(qa, qb)
        n_zero += ref == 0.0
        worst = max(worst, abs(got - ref))
        assert got == pytest.approx(ref, abs=1e-11), (it, got, ref)
    assert worst < 1e-11 and n_zero >= 1


def test_rectangle_vertices_are_the_commonroad_rectangle(oracle):
    """commonroad Rectangle semantics [ext]: vertices (+-l/2, +-w/2) rotated by yaw about the centre, then translated"""
    from sympy import Point, Polygon, Rational, cos, sin
    q = oracle.rect_vertices(1.5, -2.0, 0.7, 4.0, 2.0).reshape(4, 2)
    c, s = math.cos(0.7), math.sin(0.7)
    want = {(round(1.5 + dx * c - dy * s, 12), round(-2.0 + dx * s + dy * c, 12)) for dx in (-2.0, 2.0) for dy in (-1.0, 1.0)}
    assert {(round(float(x), 12), round(float(y), 12)) for x, y in q} == want
    poly = Polygon(*[Point(Rational(float(x)), Rational(float(y))) for x, y in q])
    assert abs(float(poly.area)) == pytest.approx(8.0, abs=1e-12)
