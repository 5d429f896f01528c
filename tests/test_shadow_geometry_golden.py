"""The shadow geometry of the scene stage against the reference's OWN arithmetic.

tests/golden/shadow_geometry.npz was produced by the reference's unmodified utils/helper_functions.py
(create_polygon_from_vertices :79-96, get_polygon_from_obstacle_occlusion / _identify_projection_points :139-176) and
sensor_model.SensorModel._calc_relevant_sector (:201-209), imported under a shapely stub whose ``Polygon`` records its vertex
list (tests/golden/gen_golden.py, ``gen_shadow_geometry``).  Checked here, without a GPU:
  * the hand restatement the cell classes are compared with (tests/ref_pointwise.py: shadow_quad, obstacle_wedge,
    sector_polygon, footprint fan) reproduces those polygons vertex for vertex;
  * the oracle's occluder predicate -- "the open segment ego -> p crosses the occluding piece" (oracle/fo_oracle_scene.c
    ray_segment / blocked_before, the statement the HIP kernels are bit-identical to) -- agrees with point-in-recorded-polygon
    for sample points away from the polygons' 100 m far edge and their boundaries: boundary quads directly, obstacle wedges
    as "in the wedge or inside the obstacle" (the reference removes both from the visible area, sensor_model.py:183-186)."""
import math
import os

import numpy as np
import pytest

import ref_pointwise as RP

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shadow_geometry.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def test_boundary_shadow_quads_equal_the_references(g):
    for ego, v1, v2, ref in zip(g["quad_ego"], g["quad_v1"], g["quad_v2"], g["quad_ref"]):
        assert np.array_equal(ref[4], ref[0])                      # the reference closes its ring explicitly (:94)
        assert np.array_equal(RP.shadow_quad(v1, v2, ego), ref[:4])


def test_obstacle_wedges_and_silhouette_pairs_equal_the_references(g):
    for ego, corn, ref, c1, c2 in zip(g["wedge_ego"], g["wedge_corners"], g["wedge_ref"], g["wedge_c1"], g["wedge_c2"]):
        w, a, b = RP.obstacle_wedge(ego, corn)
        assert np.array_equal(a, c1) and np.array_equal(b, c2)      # the same pair among the 16, ties broken the same way
        assert np.array_equal(w, ref)


def test_sectors_equal_the_references(g):
    for ego, r, a0, a1, fac, ref in zip(g["sector_ego"], g["sector_r"], g["sector_a0"], g["sector_a1"], g["sector_factor"],
                                         g["sector_ref"]):
        assert np.array_equal(ref[0], ego) and np.array_equal(ref[-1], ego) and len(ref) == 102
        assert np.array_equal(RP.sector_polygon(ego, r * fac, a0, a1), ref[:-1])
    # the product's host statement of the same fan (sensor_model.footprint_polygon: ego + 100 arc points) and of the
    # occluded area's half fan (half_fan_dirs x 1.5 r)
    from frenetix_occlusion.sensor_model import footprint_polygon, half_fan_dirs
    for ego, r, a0, a1, fac, ref in zip(g["sector_ego"], g["sector_r"], g["sector_a0"], g["sector_a1"], g["sector_factor"],
                                         g["sector_ref"]):
        yaw = 0.5 * (a0 + a1)
        if fac == 1.0:
            fov = math.degrees(a1 - a0)
            np.testing.assert_allclose(footprint_polygon(ego, yaw, fov, r), ref[:-1], rtol=0, atol=1e-9)
        else:
            np.testing.assert_allclose(ego[None] + 1.5 * r * half_fan_dirs(yaw), ref[1:-1], rtol=0, atol=1e-9)


def _edge_margin(p, poly):
    """smallest distance from the points p [N,2] to the edges of the closed polygon"""
    d = np.full(len(p), np.inf)
    for i in range(len(poly)):
        a, b = poly[i], poly[(i + 1) % len(poly)]
        e = b - a
        l2 = float(e @ e)
        t = np.clip(((p - a) @ e) / l2, 0.0, 1.0) if l2 > 0 else np.zeros(len(p))
        d = np.minimum(d, np.hypot(*(p - (a + t[:, None] * e)).T))
    return d


def test_oracle_occluder_predicate_agrees_with_point_in_the_references_quads(g, oracle):
    rng = np.random.default_rng(7)
    n_in = n_out = 0
    none_c, none_f = np.zeros((0, 4, 2)), np.zeros(0, np.uint8)
    for ego, v1, v2, ref in zip(g["quad_ego"][10:], g["quad_v1"][10:], g["quad_v2"][10:], g["quad_ref"][10:]):
        quad = ref[:4]
        mid = 0.5 * (v1 + v2)
        # sample around the far side of the edge, well inside the 100 m quad: up to 40 m behind the edge, +- its width
        u = mid - ego
        u = u / np.linalg.norm(u)
        w = np.array([-u[1], u[0]])
        span = np.linalg.norm(v2 - v1) + 2.0
        p = mid[None] + rng.uniform(-3.0, 40.0, (40, 1)) * u[None] + rng.uniform(-span, span, (40, 1)) * w[None]
        ok = _edge_margin(p, quad) > 1e-6
        p = p[ok]
        if not len(p):
            continue
        inside = RP._in_quads(p, quad[None])
        d = p - ego[None]
        dist = np.hypot(d[:, 0], d[:, 1])
        rng_, hid, _ = oracle.raycast(np.concatenate((v1, v2))[None], none_c, none_f, ego, d / dist[:, None], 1.0e4)
        blocked = rng_ < dist
        assert np.array_equal(blocked, inside)
        n_in += int(inside.sum())
        n_out += int((~inside).sum())
    assert n_in > 500 and n_out > 500


def test_oracle_obstacle_shadow_agrees_with_the_references_wedge_plus_obstacle(g, oracle):
    rng = np.random.default_rng(8)
    n_in = n_out = 0
    none_e = np.zeros((0, 4))
    for ego, corn, ref in zip(g["wedge_ego"], g["wedge_corners"], g["wedge_ref"]):
        cen = corn.mean(0)
        if RP.points_in_polygon(ego[None], corn)[0]:
            continue                                               # (an ego inside the obstacle: not a sensor pose)
        reach = np.linalg.norm(corn[0] - cen) + 25.0
        p = cen[None] + rng.uniform(-reach, reach, (60, 2))
        p = p[np.hypot(*(p - ego[None]).T) < 70.0]                 # away from the wedge's 100 m far edge
        ok = (_edge_margin(p, ref) > 1e-6) & (_edge_margin(p, corn) > 1e-6)
        # the wedge ends 100 m along its two silhouette rays (helper_functions.py:145-146); seen from close by, an obstacle
        # subtends a wide angle and the chord between those two end points passes near it.  Beyond that chord the reference's
        # wedge no longer covers the shadow (deviation (i)): compared on the ego's side of the chord only
        c3, c4 = ref[2], ref[3]
        nrm = np.array([-(c4 - c3)[1], (c4 - c3)[0]])
        side = (p - c3[None]) @ nrm
        ok &= np.sign(side) == np.sign((ego - c3) @ nrm)
        p = p[ok]
        if not len(p):
            continue
        want = RP._in_quads(p, ref[None]) | RP.points_in_polygon(p, corn)
        d = p - ego[None]
        dist = np.hypot(d[:, 0], d[:, 1])
        rng_, hid, _ = oracle.raycast(none_e, corn[None], np.array([3], np.uint8), ego, d / dist[:, None], 1.0e4)
        assert np.array_equal(rng_ < dist, want), (ego, corn)
        n_in += int(want.sum())
        n_out += int((~want).sum())
    assert n_in > 500 and n_out > 1000


def test_oracle_silhouette_pair_and_far_chord_equal_the_references(g, oracle):
    """the oracle's C statement of _identify_projection_points and of the polygon's far edge (fo_oracle_wedge_far, what
    fo_oracle_grid ends an obstacle's shadow with, and what the device's wedge_far_halfplane restates)"""
    rng = np.random.default_rng(9)
    for ego, corn, ref, c1, c2 in zip(g["wedge_ego"], g["wedge_corners"], g["wedge_ref"], g["wedge_c1"], g["wedge_c2"]):
        a, b, abc = oracle.wedge_far(ego, corn, 100.0)
        assert np.array_equal(a, c1) and np.array_equal(b, c2)
        assert abc is not None
        c3, c4 = ref[2], ref[3]                                    # [c1, c2, c2 + 100 u2, c1 + 100 u1]
        # the line through the reference's two end points, ego on the negative side
        for q in (c3, c4):
            assert abs(abc[0] * q[0] + abc[1] * q[1] + abc[2]) < 1e-9 * max(1.0, float(np.hypot(abc[0], abc[1])))
        assert abc[0] * ego[0] + abc[1] * ego[1] + abc[2] < 0.0
        assert oracle.wedge_far(ego, corn, math.inf)[2] is None and oracle.wedge_far(ego, corn, 0.0)[2] is None
    # with the far chord, the oracle's predicate equals point-in-(wedge or obstacle) everywhere -- no restriction to the
    # ego's side of the chord as in the test above
    n_in = n_out = n_beyond = 0
    none_e = np.zeros((0, 4))
    for ego, corn, ref in zip(g["wedge_ego"], g["wedge_corners"], g["wedge_ref"]):
        if RP.points_in_polygon(ego[None], corn)[0]:
            continue
        cen = corn.mean(0)
        u = (cen - ego) / np.linalg.norm(cen - ego)
        w = np.array([-u[1], u[0]])
        p = ego[None] + rng.uniform(0.5, 130.0, (80, 1)) * u[None] + rng.uniform(-60.0, 60.0, (80, 1)) * w[None]
        p = p[(_edge_margin(p, ref) > 1e-6) & (_edge_margin(p, corn) > 1e-6)]
        _, _, abc = oracle.wedge_far(ego, corn, 100.0)
        beyond = abc[0] * p[:, 0] + abc[1] * p[:, 1] + abc[2] > 0.0
        want = RP._in_quads(p, ref[None]) | RP.points_in_polygon(p, corn)
        d = p - ego[None]
        dist = np.hypot(d[:, 0], d[:, 1])
        rng_, hid, _ = oracle.raycast(none_e, corn[None], np.array([3], np.uint8), ego, d / dist[:, None], 1.0e4)
        got = ((rng_ < dist) & ~beyond) | RP.points_in_polygon(p, corn)
        assert np.array_equal(got, want), (ego, corn)
        n_in += int(want.sum())
        n_out += int((~want).sum())
        n_beyond += int(((rng_ < dist) & beyond).sum())
    assert n_in > 300 and n_out > 1000 and n_beyond > 100
