"""The ray-fan / cell discretisation of the scene stage against a pointwise restatement of the reference's polygon set
algebra (tests/ref_pointwise.py).  The reference's own arithmetic lives in GEOS (absent here), so this is what ties the
discretisation to the reference's definition of "visible" and "occluded": at every cell centre the two must agree,
except in cells within the discretisation error (ray pitch, chord sagitta) of a shadow or range boundary.  CPU only
(oracle + host logic)."""
import math
import os

import numpy as np
import pytest

from frenetix_occlusion import scenario as S
from frenetix_occlusion.sensor_model import HoleIndex, footprint_polygon, footprint_ranges, half_fan_dirs, ray_dirs

import ref_pointwise as RP

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _classes(oracle, sc, ego, yaw, r, fov, timestep, n_rays=720, cs=0.5, exact=True, shadow_length=100.0):
    g = S.MapGeometry.from_lanelets(sc.lanelets)
    corn, cen, flags, _ = sc.obstacle_arrays(timestep)
    xy = g.poly_xy
    x0 = math.floor((xy[:, 0].min() - 1.0) / cs) * cs
    y0 = math.floor((xy[:, 1].min() - 1.0) / cs) * cs
    nx = int(math.ceil((xy[:, 0].max() + 1.0 - x0) / cs))
    ny = int(math.ceil((xy[:, 1].max() + 1.0 - y0) / cs))
    raster = oracle.road_raster(g.poly_off, g.poly_xy, x0, y0, cs, nx, ny)
    dirs = ray_dirs(n_rays, yaw, fov)
    rmax = footprint_ranges(n_rays, yaw, fov, r)
    hi = HoleIndex(g)
    rings = hi.enclosed(ego, yaw, fov, r)
    skip = hi.edge_skip(rings) if rings else None
    rng, hid, _ = oracle.raycast(g.edges, corn.reshape(-1, 8), flags, ego, dirs, r, rmax=rmax, edge_skip=skip)
    hd = np.array([math.cos(yaw), math.sin(yaw)])
    ex = dict(hit_id=hid, edges=g.edges, ocorn=corn.reshape(-1, 8), oflags=flags, rmax=rmax, edge_skip=skip,
              half_dirs=half_fan_dirs(yaw), edge_line=g.edge_line, shadow_length=shadow_length) if exact else None
    cls, _, n_exact = oracle.grid(raster, x0, y0, cs, 0, 0, nx, ny, ego, hd, r, fov >= 359.9, dirs, rng, exact=ex,
                                  return_n_exact=True)
    iy, ix = np.mgrid[0:ny, 0:nx]
    q = np.stack((x0 + (ix.ravel() + 0.5) * cs, y0 + (iy.ravel() + 0.5) * cs), -1)
    present = (flags & 1) != 0
    # nothing is visible or occluded beyond 1.5 r: the restatement only has to look at the cells within reach
    sub = np.nonzero((np.abs(q[:, 0] - ego[0]) <= 1.5 * r + cs) & (np.abs(q[:, 1] - ego[1]) <= 1.5 * r + cs))[0]
    polys = [ll.polygon for ll in sc.lanelets]
    polys = [p for p in polys if p[:, 0].max() >= ego[0] - 1.5 * r - cs and p[:, 0].min() <= ego[0] + 1.5 * r + cs and
             p[:, 1].max() >= ego[1] - 1.5 * r - cs and p[:, 1].min() <= ego[1] + 1.5 * r + cs]
    road_s, vis_s, occ_s = RP.classify(q[sub], polys, g.edges, ego, yaw, r, fov, corn[present],
                                       [(f & 2) == 0 for f in flags[present]])
    road, vis, occ = (np.zeros(len(q), bool) for _ in range(3))
    road[sub], vis[sub], occ[sub] = road_s, vis_s, occ_s
    c = cls.ravel()
    road[np.setdiff1d(np.arange(len(q)), sub)] = ((c & 1) != 0)[np.setdiff1d(np.arange(len(q)), sub)]
    return dict(q=q, road=road, vis=vis, occ=occ, o_road=(c & 1) != 0, o_vis=(c & 2) != 0, o_occ=(c & 4) != 0,
                rings=rings, g=g, cs=cs, n_exact=n_exact)


CASES = [(1, 0, 360.0, 50.0), (1, 20, 360.0, 50.0), (2, 0, 360.0, 50.0), (3, 0, 360.0, 50.0), (1, 0, 120.0, 40.0),
         (3, 5, 200.0, 30.0)]


@pytest.mark.parametrize("k,timestep,fov,r", CASES)
def test_cell_classes_equal_the_reference_set_algebra_at_every_cell_centre(oracle, k, timestep, fov, r):
    """default configuration (polygon footprint, enclosed holes transparent, exact settlement of the cells the fan
    cannot decide): visible and occluded cells are exactly the cells whose centre the reference's polygons contain"""
    sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{k}_geometry.npz"))
    ego = sc.ego_initial
    s = _classes(oracle, sc, ego[:2], float(ego[2]), r, fov, timestep)
    assert np.array_equal(s["road"], s["o_road"])                   # same crossing-number rule
    assert int(s["vis"].sum()) > 300 and int(s["occ"].sum()) > 100
    assert np.array_equal(s["vis"], s["o_vis"])
    assert np.array_equal(s["occ"], s["o_occ"])
    assert 0 < s["n_exact"] < 0.5 * (s["vis"].sum() + s["occ"].sum())   # the fan decides most cells on its own


def test_an_obstacle_seen_from_one_metre_ends_its_shadow_where_the_references_polygon_does(oracle):
    """helper_functions.py:145-146: the occlusion polygon of an obstacle ends 100 m along its two silhouette sight lines.
    A 12 m truck across the road one metre ahead of the ego subtends ~160 deg: the chord between the two end points passes ~18 m from the
    ego, and the road beyond it is VISIBLE to the reference.  The default (shadow_length 100) reproduces that cell for
    cell; with the physical shadow (inf) those cells are hidden."""
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    yaw = float(ego[2])
    fwd = np.array([math.cos(yaw), math.sin(yaw)])
    left = np.array([-math.sin(yaw), math.cos(yaw)])
    for gap, shift in ((1.0, 0.0), (0.6, 2.0)):
        # across the road, `gap` metres ahead of the ego
        cen = ego[:2] + (gap + 1.25) * fwd + shift * left
        truck = S.Obstacle(900, "static", "truck", 12.0, 2.5, 0, np.array([cen[0], cen[1], yaw + 0.5 * math.pi, 0.0]),
                           np.zeros((0, 4)))
        sc2 = S.Scenario(sc.dt, sc.lanelets, [truck], sc.intersections, sc.ego_initial, sc.benchmark_id)
        s = _classes(oracle, sc2, ego[:2], yaw, 50.0, 360.0, 0)
        assert np.array_equal(s["vis"], s["o_vis"]) and np.array_equal(s["occ"], s["o_occ"])
        # the cells beyond the far chord: hidden by the physical shadow, visible in the reference
        t = _classes(oracle, sc2, ego[:2], yaw, 50.0, 360.0, 0, shadow_length=math.inf)
        lost = s["o_vis"] & ~t["o_vis"]
        assert lost.sum() > 50 and not (t["o_vis"] & ~s["o_vis"]).any()
        _, _, abc = oracle.wedge_far(ego[:2], truck.corners(truck.initial), 100.0)
        q = s["q"][lost]
        assert (abc[0] * q[:, 0] + abc[1] * q[:, 1] + abc[2] > 0.0).all()


def test_moving_ego_keeps_the_cell_classes_equal_to_the_reference_set_algebra(oracle):
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial.copy()
    for step in (7, 33, 61):
        pos = ego[:2] + 0.7 * step * np.array([math.cos(ego[2]), math.sin(ego[2])])
        s = _classes(oracle, sc, pos, float(ego[2]) + 0.01 * step, 50.0, 360.0, step)
        assert np.array_equal(s["vis"], s["o_vis"]) and np.array_equal(s["occ"], s["o_occ"]), step


@pytest.mark.parametrize("k,timestep,fov,r", CASES[:3])
def test_fan_rule_alone_differs_only_along_shadow_and_range_boundaries(oracle, k, timestep, fov, r):
    """cell_visibility = "fan": the chord rule loses the cells a shadow edge between two rays cuts through"""
    sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{k}_geometry.npz"))
    ego = sc.ego_initial
    s = _classes(oracle, sc, ego[:2], float(ego[2]), r, fov, timestep, exact=False)
    bad_vis, bad_occ = s["vis"] != s["o_vis"], s["occ"] != s["o_occ"]
    assert 0 < bad_vis.sum() <= 0.03 * s["vis"].sum() and bad_occ.sum() <= 0.03 * s["occ"].sum()
    q = s["q"]
    for i in np.nonzero(bad_vis | bad_occ)[0]:
        reach = 1.5 * s["cs"] + np.hypot(*(q[i] - ego[:2])) * math.radians(fov / 720.0) * 1.5
        near = np.nonzero((np.abs(q[:, 0] - q[i, 0]) <= reach) & (np.abs(q[:, 1] - q[i, 1]) <= reach))[0]
        assert (s["vis"][near] != s["vis"][i]).any(), (q[i], "isolated disagreement")


def test_scenario1_has_a_hole_the_footprint_encloses_and_it_casts_no_shadow(oracle):
    """the lanelets behind the scenario-1 ego leave a sliver between them -- an interior ring of the road
    union, 11-33 m behind the ego.  The reference walks exterior rings only, so the road beyond the sliver stays
    visible; with every piece occluding, ~190 cells behind the ego would be lost."""
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    s = _classes(oracle, sc, ego[:2], float(ego[2]), 50.0, 360.0, 0)
    g = s["g"]
    assert int(g.ring_is_hole.sum()) == 1 and len(s["rings"]) == 1
    assert np.array_equal(~g.ring_is_hole[g.edge_ring], RP.exterior_edge_mask(g.edges))
    hole = g.edges[g.ring_is_hole[g.edge_ring]]
    assert len(hole) == 7 and hole[:, [0, 2]].max() < -11.0 and hole[:, [0, 2]].min() > -33.0
    # far from the ego the footprint no longer encloses it: it is then part of the exterior of road ∩ footprint
    hi = HoleIndex(g)
    assert hi.enclosed(ego[:2] + np.array([35.0, 0.0]), float(ego[2]), 360.0, 50.0) == ()
    assert hi.enclosed(ego[:2], float(ego[2]), 90.0, 50.0) == ()        # forward fan: the sliver is behind


def test_footprint_ranges_trace_the_reference_polygons():
    for fov, n in ((360.0, 720), (360.0, 97), (120.0, 241), (200.0, 400)):
        yaw, r = -0.7, 42.0
        d, rm = ray_dirs(n, yaw, fov), footprint_ranges(n, yaw, fov, r)
        foot = footprint_polygon(np.zeros(2), yaw, fov, r)
        inner = slice(1, -1) if fov < 359.9 else slice(None)          # the fan's first / last ray run along its sides
        assert S.points_in_polygon((d * (rm - 1e-9)[:, None])[inner], foot).all()
        assert not S.points_in_polygon((d * (rm + 1e-9)[:, None])[inner], foot).any()
        assert rm.max() <= r + 1e-12 and rm.min() >= r * math.cos(math.pi / 64) - 1e-12
    # shapely's Point.buffer(r) [ext]: 64 segments, one vertex on the +x axis
    foot = footprint_polygon(np.array([1.0, 2.0]), 0.3, 360.0, 10.0)
    assert foot.shape == (64, 2) and np.allclose(foot[0], [11.0, 2.0])
    assert np.allclose(RP.footprint_polygon(np.array([1.0, 2.0]), 10.0, 0.3, 360.0)[0], foot[0])


def test_boundary_rings_of_nested_squares():
    """ring labelling: a square annulus road has one exterior ring and one hole; an island inside the hole is an
    exterior ring again (enclosed by two rings)"""
    def sq(a):
        return np.array([[-a, -a], [a, -a], [a, a], [-a, a]], float)

    def ring_edges(p):
        return np.concatenate((p, np.roll(p, -1, axis=0)), axis=1)
    edges = np.concatenate((ring_edges(sq(10)), ring_edges(sq(6)), ring_edges(sq(2))))
    ring, hole = S.boundary_rings(edges)
    assert len(hole) == 3 and [bool(hole[ring[i]]) for i in (0, 4, 8)] == [False, True, False]
    lab, hole2 = RP.ring_labels(edges)
    assert [bool(hole2[lab[i]]) for i in (0, 4, 8)] == [False, True, False]


def test_random_poses_fans_and_radii(oracle):
    """ten seeded poses on the three scenario maps (full circle and open fans, three radii, obstacles at random time
    steps): no cell differs.  Poses closer than 3 m to an obstacle are skipped -- there the reference's obstacle
    shadow, a quad reaching 100 m along the two silhouette rays (helper_functions.py:133-141), ends before the sensor
    range does once the obstacle subtends more than ~120 degrees; the ray fan has no such cut-off."""
    rng = np.random.default_rng(11)
    done = 0
    while done < 10:
        k = int(rng.integers(1, 4))
        sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{k}_geometry.npz"))
        c = sc.lanelets[int(rng.integers(len(sc.lanelets)))].center
        i = int(rng.integers(len(c) - 1))
        pos = c[i] + rng.uniform(0, 1) * (c[i + 1] - c[i]) + rng.normal(0, 0.4, 2)
        yaw = math.atan2(*(c[i + 1] - c[i])[::-1]) + rng.normal(0, 0.2)
        fov = float(rng.choice([360.0, 360.0, 120.0, 220.0, 90.0]))
        r = float(rng.choice([50.0, 30.0, 42.5]))
        ts = int(rng.integers(0, 80))
        _, cen, flags, _ = sc.obstacle_arrays(ts)
        if any(f & 1 and np.hypot(*(cc - pos)) < 3.0 + 2.7 for cc, f in zip(cen, flags)):
            continue
        s = _classes(oracle, sc, pos, yaw, r, fov, ts)
        assert np.array_equal(s["vis"], s["o_vis"]) and np.array_equal(s["occ"], s["o_occ"]), (k, pos, yaw, fov, r, ts)
        done += 1


def test_synthetic_urban_grid_matches_the_reference_set_algebra(oracle):
    """BASELINE configs[2] scene (9 360 boundary pieces, 64 parked cars, 65 rings of which 64 are city blocks the
    footprint cuts open): the cell classes of the 150 m window equal the reference's set algebra at every centre"""
    sc = S.synthetic_urban_grid()
    ego = sc.ego_initial
    s = _classes(oracle, sc, ego[:2], float(ego[2]), 50.0, 360.0, 0)
    assert int(s["g"].ring_is_hole.sum()) == 64 and s["rings"] == ()
    assert int(s["vis"].sum()) > 5000 and int(s["occ"].sum()) > 9000
    assert np.array_equal(s["vis"], s["o_vis"]) and np.array_equal(s["occ"], s["o_occ"])
    assert len(np.unique(s["g"].edge_line)) < 0.2 * len(s["g"].edges)      # straight kerbs collapse into chains


def test_visible_objects_agree_with_the_reference_predicate(oracle):
    """visible_objects_timestep (ref sensor_model.py:59-76) on the three scenario maps while the ego and the obstacles
    move: fan hits + probe points against "the obstacle's outline, pushed out past the 5 mm skin, has a visible point"."""
    n = 0
    for k in (1, 2, 3):
        sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{k}_geometry.npz"))
        g = S.MapGeometry.from_lanelets(sc.lanelets)
        polys = [ll.polygon for ll in sc.lanelets]
        e0 = sc.ego_initial
        for ts in (0, 10, 25, 40, 60):
            ego = e0[:2] + 0.7 * ts * np.array([math.cos(e0[2]), math.sin(e0[2])])
            yaw, r, fov = float(e0[2]), 50.0, 360.0
            corn, cen, flags, _ = sc.obstacle_arrays(ts)
            dirs, rmax = ray_dirs(720, yaw, fov), footprint_ranges(720, yaw, fov, r)
            hi = HoleIndex(g)
            rings = hi.enclosed(ego, yaw, fov, r)
            skip = hi.edge_skip(rings) if rings else None
            _, hid, _ = oracle.raycast(g.edges, corn.reshape(-1, 8), flags, ego, dirs, r, rmax=rmax, edge_skip=skip)
            ours = oracle.obstacle_visibility(g.edges, corn, cen, flags, ego, r, True, dirs, edge_skip=skip, hit_id=hid)
            present = (flags & 1) != 0
            ref = RP.obstacles_visible(polys, g.edges, ego, yaw, r, fov, corn[present], [(f & 2) == 0 for f in flags[present]])
            assert np.array_equal(ours[present].astype(bool), ref), (k, ts)
            n += int(present.sum())
    assert n >= 30
