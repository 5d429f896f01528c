"""Analytic known-answer tests for the scene half of the oracle (ray fan, cell classes, obstacle visibility, spawn
sampling, predictions) and for the host-side map preparation.  The reference computes these with GEOS polygon algebra
(absent here): these tests are what pins the restatement ("parity unpinned" vs the reference).  CPU only."""
import math
import os

import numpy as np
import pytest

from frenetix_occlusion import scenario as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rect_lanelet(lid, x0, x1, y0, y1, n=2):
    xs = np.linspace(x0, x1, n)
    return S.Lanelet(lid, np.stack((xs, np.full(n, y1)), -1), np.stack((xs, np.full(n, y0)), -1))


def room(x0=-20.0, x1=20.0, y0=-5.0, y1=5.0):
    ll = rect_lanelet(1, x0, x1, y0, y1)
    return S.MapGeometry.from_lanelets([ll])


def test_union_boundary_of_one_rectangle_is_its_four_sides():
    g = room()
    assert g.edges.shape == (4, 4)
    lens = np.hypot(g.edges[:, 2] - g.edges[:, 0], g.edges[:, 3] - g.edges[:, 1])
    assert sorted(lens) == [10.0, 10.0, 40.0, 40.0]


def test_union_boundary_drops_shared_edges_and_keeps_t_junction_pieces():
    a = rect_lanelet(1, 0, 10, 0, 4)
    b = rect_lanelet(2, 10, 20, 0, 4)           # shares the edge x = 10
    c = rect_lanelet(3, 4, 8, 4, 12)            # sits on top of a: T junction
    g = S.MapGeometry.from_lanelets([a, b, c])
    total = np.hypot(g.edges[:, 2] - g.edges[:, 0], g.edges[:, 3] - g.edges[:, 1]).sum()
    # perimeter of the union: outer rectangle 20 x 4 (48) - the 4 m where c attaches + c's three free sides (8+4+8)
    assert total == pytest.approx(48.0 - 4.0 + 20.0, abs=1e-9)
    # no edge lies on x = 10 strictly inside the road
    mid = 0.5 * (g.edges[:, :2] + g.edges[:, 2:])
    assert not np.any((np.abs(mid[:, 0] - 10.0) < 1e-9) & (mid[:, 1] > 0.1) & (mid[:, 1] < 3.9))


def test_ray_ranges_in_a_rectangular_room(oracle):
    g = room()
    ego = np.array([3.0, 1.0])
    dirs = oracle.ray_dirs(720)
    rng, hid, ring = oracle.raycast(g.edges, np.zeros((0, 8)), np.zeros(0, np.uint8), ego, dirs, 50.0)
    # analytic: distance to the nearest wall along each ray
    with np.errstate(divide="ignore"):
        tx = np.where(dirs[:, 0] > 0, (20 - ego[0]) / dirs[:, 0], np.where(dirs[:, 0] < 0, (-20 - ego[0]) / dirs[:, 0], np.inf))
        ty = np.where(dirs[:, 1] > 0, (5 - ego[1]) / dirs[:, 1], np.where(dirs[:, 1] < 0, (-5 - ego[1]) / dirs[:, 1], np.inf))
    np.testing.assert_allclose(rng, np.minimum(tx, ty), rtol=0, atol=1e-12)
    assert (hid >= 0).all() and (hid < 4).all()
    np.testing.assert_allclose(ring, ego[None] + rng[:, None] * dirs, atol=1e-12)
    # sensor radius clamps and reports "no occluder"
    rng2, hid2, _ = oracle.raycast(g.edges, np.zeros((0, 8)), np.zeros(0, np.uint8), ego, dirs, 4.5)
    assert (rng2 <= 4.5).all() and (hid2[rng2 == 4.5] == -1).all() and (hid2[rng2 < 4.5] >= 0).all()


def test_box_obstacle_casts_the_expected_shadow_and_bicycles_do_not(oracle):
    g = room(-40, 40, -20, 20)
    ego = np.array([0.0, 0.0])
    ob = S.Obstacle(7, "static", "car", 4.0, 2.0, 0, np.array([10.0, 0.0, 0.0, 0.0]), np.zeros((0, 4)))
    corn = ob.corners(ob.initial)[None]
    dirs = oracle.ray_dirs(3600)
    rng, hid, _ = oracle.raycast(g.edges, corn, np.array([3], np.uint8), ego, dirs, 100.0)
    E = len(g.edges)
    shadow = hid == E + 0
    # silhouette from the origin: the near face x = 8, |y| <= 1  ->  |angle| <= atan(1/8)
    ang = np.arctan2(dirs[:, 1], dirs[:, 0])
    expect = np.abs(ang) <= math.atan2(1.0, 8.0) + 1e-12
    assert np.array_equal(shadow, expect)
    np.testing.assert_allclose(rng[shadow], 8.0 / dirs[shadow, 0], atol=1e-12)
    # a bicycle (flag bit1 clear) never occludes (sensor_model.py:177), an absent obstacle neither
    for fl in (1, 0, 2):
        _, hid_b, _ = oracle.raycast(g.edges, corn, np.array([fl], np.uint8), ego, dirs, 100.0)
        assert (hid_b < E).all()


def _grid_for(oracle, g, ego, yaw, r, n_rays=720, cs=0.5, ocorn=None, oflags=None, fov=360.0):
    xy = g.poly_xy
    x0 = math.floor((xy[:, 0].min() - 1.0) / cs) * cs
    y0 = math.floor((xy[:, 1].min() - 1.0) / cs) * cs
    nx = int(math.ceil((xy[:, 0].max() + 1.0 - x0) / cs))
    ny = int(math.ceil((xy[:, 1].max() + 1.0 - y0) / cs))
    raster = oracle.road_raster(g.poly_off, g.poly_xy, x0, y0, cs, nx, ny)
    dirs = oracle.ray_dirs(n_rays, yaw, fov)
    ocorn = np.zeros((0, 8)) if ocorn is None else ocorn
    oflags = np.zeros(0, np.uint8) if oflags is None else oflags
    rng, hid, ring = oracle.raycast(g.edges, ocorn, oflags, ego, dirs, r)
    hd = np.array([math.cos(yaw), math.sin(yaw)])
    cls, occ = oracle.grid(raster, x0, y0, cs, 0, 0, nx, ny, ego, hd, r, fov >= 359.9, dirs, rng)
    return dict(raster=raster, x0=x0, y0=y0, nx=nx, ny=ny, cs=cs, dirs=dirs, rng=rng, hid=hid, ring=ring, cls=cls,
                occ=occ, hd=hd)


def test_road_raster_area(oracle):
    g = room()
    s = _grid_for(oracle, g, np.array([0.0, 0.0]), 0.0, 50.0)
    assert s["raster"].sum() * 0.25 == pytest.approx(400.0, abs=1e-9)   # 40 x 10 m, cell-aligned


def test_empty_room_is_fully_visible_and_nothing_is_occluded(oracle):
    g = room()
    s = _grid_for(oracle, g, np.array([0.1, 0.1]), 0.3, 50.0)
    road = (s["cls"] & 1) != 0
    vis = (s["cls"] & 2) != 0
    # every road cell is visible except a thin band along the walls (chord of the fan cuts the corners)
    assert vis.sum() >= 0.985 * road.sum()
    assert ((s["cls"] & 4) != 0).sum() <= 0.015 * road.sum()
    assert not (vis & ~road).any()


def test_sensor_radius_limits_visibility_and_half_disc_limits_occlusion(oracle):
    g = room(-100, 100, -4, 4)
    ego = np.array([0.0, 0.0])
    s = _grid_for(oracle, g, ego, 0.0, 20.0)
    ix = np.arange(s["nx"])
    xc = s["x0"] + (ix + 0.5) * s["cs"]
    row = s["cls"][s["ny"] // 2]            # a row through the corridor
    vis, occ = (row & 2) != 0, (row & 4) != 0
    assert vis[np.abs(xc) < 19.0].all() and not vis[np.abs(xc) > 20.5].any()
    # occluded = road, not visible, within 1.5 r, ahead of the ego (heading +x): 20 < x <= 30
    assert occ[(xc > 20.5) & (xc < 29.5)].all()
    assert not occ[xc < 0].any() and not occ[xc > 30.5].any()
    assert np.array_equal(np.nonzero((s["cls"] & 4).ravel())[0], s["occ"])   # ascending index list


def test_l_shaped_road_hides_the_side_arm(oracle):
    main = rect_lanelet(1, -30, 30, -3, 3, n=31)
    arm = S.Lanelet(2, np.stack((np.full(21, 10.0), np.linspace(3, 43, 21)), -1),
                    np.stack((np.full(21, 16.0), np.linspace(3, 43, 21)), -1))
    g = S.MapGeometry.from_lanelets([main, arm])
    ego = np.array([-20.0, 0.0])
    s = _grid_for(oracle, g, ego, 0.0, 50.0)
    cs = s["cs"]

    def cell(x, y):
        return s["cls"][int((y - s["y0"]) / cs), int((x - s["x0"]) / cs)]
    assert cell(13.0, 30.0) & 4 and not cell(13.0, 30.0) & 2       # deep in the arm: hidden behind the corner
    assert cell(25.0, 0.0) & 2                                      # straight ahead: visible
    # the sight line from the ego through the corner (10, 3) bounds the visible part of the arm
    x, y = 15.0, 4.0
    slope = 3.0 / 30.0                                              # line from (-20,0) to (10,3)
    assert (y < slope * (x + 20.0)) == bool(cell(x, y) & 2)
    assert cell(15.0, 8.0) & 4


def test_open_fan_sees_only_its_sector(oracle):
    g = room(-30, 30, -30, 30)
    ego = np.array([0.0, 0.0])
    s = _grid_for(oracle, g, ego, 0.0, 20.0, n_rays=181, fov=90.0)
    cs = s["cs"]

    def cell(x, y):
        return s["cls"][int((y - s["y0"]) / cs), int((x - s["x0"]) / cs)]
    assert cell(10.0, 2.0) & 2 and cell(10.0, -8.0) & 2
    assert not cell(2.0, 10.0) & 2 and not cell(-10.0, 0.0) & 2
    assert cell(2.0, 10.0) & 4                                      # ahead (x > 0), on the road, not visible
    assert not cell(-10.0, 0.3) & 4                                 # behind the ego: not in the half disc


def test_obstacle_visibility_flags(oracle):
    g = room(-60, 60, -20, 20)
    ego = np.array([0.0, 0.0])
    mk = lambda i, x, y, l=4.0, w=2.0, typ="car": S.Obstacle(i, "static", typ, l, w, 0, np.array([x, y, 0.0, 0.0]), np.zeros((0, 4)))
    obs = [mk(1, 10, 0), mk(2, 20, 0, 2.0, 1.0), mk(3, 0, 10), mk(4, 55, 0), mk(5, 30, 0, 2.0, 1.0, "bicycle"), mk(6, 10, 12)]
    corn = np.stack([o.corners(o.initial) for o in obs])
    cen = np.stack([o.initial[:2] for o in obs])
    flags = np.array([3, 3, 3, 3, 1, 0], dtype=np.uint8)
    dirs = oracle.ray_dirs(720)
    vis = oracle.obstacle_visibility(g.edges, corn, cen, flags, ego, 50.0, True, dirs)
    # 1 visible; 2 fully inside 1's shadow; 3 visible; 4 beyond the radius; 5 hidden behind 1 as well; 6 absent
    assert list(vis) == [1, 0, 1, 0, 0, 0]


def test_obstacle_seen_only_between_its_probe_points(oracle):
    """a long obstacle whose four corners and centre are all hidden behind three small blockers, but whose side is lit
    through the gaps between them: no probe point sees it, the fan rays that stop at it do (ref sensor_model.py:59-76:
    the obstacle polygon touches the visible area)"""
    g = room(-5, 40, -15, 15)
    ego = np.array([0.0, 0.0])
    mk = lambda i, x, y, l, w: S.Obstacle(i, "static", "car", l, w, 0, np.array([x, y, 0.0, 0.0]), np.zeros((0, 4)))
    obs = [mk(1, 30, 0, 2.0, 20.0), mk(2, 10, -3.3, 1.0, 1.2), mk(3, 10, 0.0, 1.0, 1.0), mk(4, 10, 3.3, 1.0, 1.2)]
    corn = np.stack([o.corners(o.initial) for o in obs])
    cen = np.stack([o.initial[:2] for o in obs])
    flags = np.full(4, 3, dtype=np.uint8)
    dirs = oracle.ray_dirs(720)
    E = len(g.edges)
    _, hid, _ = oracle.raycast(g.edges, corn.reshape(-1, 8), flags, ego, dirs, 50.0)
    assert (hid == E + 0).sum() > 10                                 # rays at ~10 degrees reach the long side
    assert list(oracle.obstacle_visibility(g.edges, corn, cen, flags, ego, 50.0, True, dirs)) == [0, 1, 1, 1]
    assert list(oracle.obstacle_visibility(g.edges, corn, cen, flags, ego, 50.0, True, dirs, hit_id=hid)) == [1, 1, 1, 1]


def test_spawn_sampling_takes_evenly_spaced_frontier_cells(oracle):
    main = rect_lanelet(1, -30, 30, -3, 3, n=31)
    arm = S.Lanelet(2, np.stack((np.full(21, 10.0), np.linspace(3, 43, 21)), -1),
                    np.stack((np.full(21, 16.0), np.linspace(3, 43, 21)), -1))
    g = S.MapGeometry.from_lanelets([main, arm])
    ego = np.array([-20.0, 0.0])
    s = _grid_for(oracle, g, ego, 0.0, 50.0)
    cell, pos, n, n_cand = oracle.spawn_cells(s["cls"], s["x0"], s["y0"], s["cs"], 0, 0, ego, s["hd"], 3.0, 60.0, 8)
    assert n == 8 and n_cand > 8
    cls = s["cls"].ravel()
    nx = s["nx"]
    for c in cell:
        assert cls[c] & 4
        assert any(cls[c + d] & 2 for d in (-1, 1, -nx, nx))
    assert (np.diff(cell) > 0).all()
    # fewer candidates than slots: all are taken, the rest stay -1
    cell2, _, n2, nc2 = oracle.spawn_cells(s["cls"], s["x0"], s["y0"], s["cs"], 0, 0, ego, s["hd"], 3.0, 60.0, 4096)
    assert n2 == nc2 == n_cand and (cell2[n2:] == -1).all()
    # gate: nothing closer than min_ahead along the heading, nothing beyond max_dist
    _, pos3, n3, _ = oracle.spawn_cells(s["cls"], s["x0"], s["y0"], s["cs"], 0, 0, ego, s["hd"], 32.0, 34.0, 4096)
    rel = pos3[:n3] - ego
    assert (rel[:, 0] >= 32.0).all() and (np.hypot(rel[:, 0], rel[:, 1]) <= 34.0).all()
    # all_occluded: every occluded cell in range qualifies, frontier or not
    cell4, _, n4, nc4 = oracle.spawn_cells(s["cls"], s["x0"], s["y0"], s["cs"], 0, 0, ego, s["hd"], 3.0, 60.0, 100000, True)
    rel_all = np.stack(((np.arange(s["nx"]) + 0.5) * s["cs"] + s["x0"] - ego[0],), 0)
    occ = np.nonzero(cls & 4)[0]
    cx = s["x0"] + (occ % nx + 0.5) * s["cs"] - ego[0]
    cy = s["y0"] + (occ // nx + 0.5) * s["cs"] - ego[1]
    expect = occ[(cx >= 3.0) & (cx * cx + cy * cy <= 3600.0)]
    assert nc4 == n4 == len(expect) > n_cand and np.array_equal(cell4[:n4], expect)


def test_pedestrian_heading_and_constant_velocity_prediction(oracle):
    path = np.stack((np.linspace(0, 50, 26), np.zeros(26)), -1)
    pos = np.array([[10.0, 4.0], [20.0, -3.0]])
    yaw = oracle.spawn_headings(pos, np.array([4, 4]), path)
    np.testing.assert_allclose(yaw, [1.5 * math.pi, 0.5 * math.pi], atol=1e-15)     # towards the path, in [0, 2 pi)
    lane = np.array([0.25, np.nan])
    yaw2 = oracle.spawn_headings(pos, np.array([0, 0]), path, lane)
    np.testing.assert_allclose(yaw2, [0.25, 0.5 * math.pi], atol=1e-15)              # lane heading, NaN -> fallback
    p, yl, vl, cov = oracle.cv_predictions(pos, yaw, np.array([1.4, 1.4]), 31, 0.1)
    # agent.py:492-505: velocity components rounded to 3 decimals; cov_k = 0.1 * 1.05^k I  (agent.py:260-280)
    np.testing.assert_allclose(p[0, :, 1], 4.0 - 1.4 * 0.1 * np.arange(31), atol=1e-12)
    np.testing.assert_allclose(p[0, :, 0], 10.0, atol=1e-12)
    np.testing.assert_allclose(cov[1, :, 0, 0], 0.1 * 1.05 ** np.arange(31), rtol=1e-14)
    assert (cov[:, :, 0, 1] == 0).all() and (yl[1] == yaw[1]).all() and (vl == 1.4).all()


@pytest.mark.parametrize("k,n_ll,n_ob", [(1, 12, 5), (2, 16, 1), (3, 16, 1)])
def test_scenario_fixtures_have_the_surveyed_shape(k, n_ll, n_ob):
    sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{k}_geometry.npz"))
    assert len(sc.lanelets) == n_ll and len(sc.obstacles) == n_ob
    if k == 1:
        assert sum(len(ll.polygon) for ll in sc.lanelets) == 352          # SURVEY §2 "≈352 polygon vertices"
        np.testing.assert_allclose(sc.ego_initial[:2], [0.0, 0.0])
        assert all(o.length == 5.0 and o.width == 2.0 for o in sc.obstacles)
        # fo_obstacle.py:79-93 timestep semantics
        ob = sc.obstacles[0]
        assert np.array_equal(ob.pose_at(ob.initial_time_step), ob.initial)
        assert np.array_equal(ob.pose_at(ob.initial_time_step + 1), ob.states[0])
        assert ob.pose_at(ob.initial_time_step + len(ob.states) + 1) is None


def test_scenario1_visibility_sanity(oracle):
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    g = S.MapGeometry.from_lanelets(sc.lanelets)
    ego = sc.ego_initial
    corn, cen, flags, _ = sc.obstacle_arrays(0)
    s = _grid_for(oracle, g, ego[:2], float(ego[2]), 50.0, ocorn=corn, oflags=flags)
    road, vis, occ = (s["cls"] & 1) != 0, (s["cls"] & 2) != 0, (s["cls"] & 4) != 0
    assert road.sum() > 1000 and vis.sum() > 100 and occ.sum() > 50
    assert not (vis & occ).any() and not ((vis | occ) & ~road).any()
    # the ego's own cell is visible
    iy, ix = int((ego[1] - s["y0"]) / s["cs"]), int((ego[0] - s["x0"]) / s["cs"])
    assert vis[iy, ix]


def test_route_enumeration_and_vehicle_predictions_along_routes(oracle):
    """route_planner.py:54-90 restated (depth-2 DFS over successors / same-direction neighbours) + one prediction per
    candidate route: the reference's min-var(v) Frenet sample -- speed held along the route, quintic lateral move over 3 s
    to the nearest of d1 in {-0.5, 0, 0.5} (agent.py:349-379, frenetix_handler.py:82-105)"""
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario3_geometry.npz"))
    routes = S.enumerate_routes(sc.lanelets)
    assert routes[1] == [[1, 3, 5], [1, 12, 9]]                    # incoming lanelet: left-ish and right turn
    tab = S.RouteTable.from_lanelets(sc.lanelets, R=3)
    idx = {l.lanelet_id: i for i, l in enumerate(sc.lanelets)}
    assert list(tab.count[idx[1] * 3: idx[1] * 3 + 3] > 0) == [True, True, False]
    pos0 = np.array([[10.0, 0.3], [20.0, -0.2], [5.0, 9.0], [12.0, 0.0]])
    types = np.array([0, 3, 4, 0])
    speed = np.array([10.0, 5.0, 1.4, 10.0])
    lan = np.array([idx[1], idx[1], -1, -1])
    pos, yaw, v, cov, ln = oracle.route_predictions(pos0, types, speed, lan, 3, tab.first, tab.count, tab.xy, tab.s,
                                                    np.array([0.0, 0.0, 1.0, 0.5]), 31, 0.1)
    assert ln.reshape(4, 3).tolist() == [[31, 31, 0], [31, 31, 0], [31, 0, 0], [31, 0, 0]]
    # car 0, route 0: on the straight part the offset to the centre line y = 0 moves from 0.3 towards the nearest lateral
    # target, 0.5, along the quintic; x advances 1 m per step (s' = v0)
    tau = np.arange(15) * 0.1 / 3.0
    np.testing.assert_allclose(pos[0, :15, 1], 0.3 + 0.2 * (10 * tau ** 3 - 15 * tau ** 4 + 6 * tau ** 5), atol=1e-12)
    np.testing.assert_allclose(pos[0, :15, 0], 10.0 + np.arange(15), atol=1e-9)
    assert yaw[0, 0] == 0.0 and 0.0 < yaw[0, 10] < 0.02 and v[0, 0] == 10.0 and 10.0 < v[0, 10] < 10.01
    # the two routes of car 0 part ways at the intersection: left (y grows) and right (y falls), heading follows
    assert pos[0, 30, 1] > 3.0 and pos[1, 30, 1] < -3.0 and yaw[0, 30] > 1.0 and yaw[1, 30] < -1.0
    step = np.hypot(np.diff(pos[1, :, 0]), np.diff(pos[1, :, 1]))
    assert np.all(np.abs(step - 1.0) < 0.2)                         # speed held along the centre line; the lateral offset (0.3 -> 0.5 m, outside of the bend) stretches the steps in the turn
    # pedestrian / off-lane vehicle: one straight prediction with the fallback heading, velocity components rounded
    np.testing.assert_allclose(pos[6, 10], pos0[2] + 1.0 * np.round(1.4 * np.array([math.cos(1.0), math.sin(1.0)]), 3))
    np.testing.assert_allclose(pos[9, 10], pos0[3] + 1.0 * np.round(10.0 * np.array([math.cos(0.5), math.sin(0.5)]), 3))
    # a route shorter than the horizon ends the prediction early
    short = S.Lanelet(1, np.array([[0, 1.75], [12.0, 1.75]]), np.array([[0, -1.75], [12.0, -1.75]]))
    t1 = S.RouteTable.from_lanelets([short], R=2)
    _, _, _, _, l1 = oracle.route_predictions(np.array([[2.0, 0.5]]), np.array([0]), np.array([10.0]), np.array([0]), 2,
                                              t1.first, t1.count, t1.xy, t1.s, np.array([0.0]), 31, 0.1)
    assert l1.tolist() == [11, 0]                                   # s = 2, 3, ..., 12


def test_lanelet_index_raster_matches_the_lane_heading_raster():
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ras = S.lanelet_index_raster(sc.lanelets, -60.0, -60.0, 0.5, 400, 300)
    yaw = S.lane_yaw_raster(sc.lanelets, -60.0, -60.0, 0.5, 400, 300)
    assert np.array_equal(ras >= 0, ~np.isnan(yaw)) and (ras >= 0).sum() > 1000 and ras.max() < len(sc.lanelets)


def test_future_visibility_extension_in_a_room_with_a_wall(oracle):
    """extension: a corridor with a box in the middle; poses before, beside and past the box.  The
    polygon of hit points fills the corridor minus the box shadow; the cells behind the box are revealed only from the
    poses that have passed it."""
    g = room(-30, 30, -5, 5)
    box = S.Obstacle(1, "static", "car", 2.0, 6.0, 0, np.array([0.0, -2.0, 0.0, 0.0]), np.zeros((0, 4)))
    corn = box.corners(box.initial)[None]
    flags = np.array([3], np.uint8)
    ego = np.array([-20.0, -2.0])
    s = _grid_for(oracle, g, ego, 0.0, 60.0, ocorn=corn.reshape(-1, 8), oflags=flags)
    occ = s["occ"]
    assert len(occ) > 100                                            # the shadow of the box down the corridor
    T = 5
    x = np.array([np.linspace(-20.0, 20.0, T)])                      # one trajectory driving past the box at y = 3.5
    y = np.full((1, T), 3.5)
    dirs = oracle.ray_dirs(256)
    rev, area = oracle.future_visibility(x, y, 1, dirs, 60.0, g.edges, corn.reshape(-1, 8), flags, occ, s["x0"], s["y0"],
                                         s["cs"], 0, 0, s["nx"])
    assert rev.shape == (1, T) and area.shape == (1, T)
    assert rev[0, 0] < rev[0, 2] < rev[0, 4]                          # more of the shadow is seen the farther it gets
    assert rev[0, 4] >= 0.9 * len(occ)                                # past the box (almost) everything behind it shows
    # the visible polygon can never exceed the corridor minus the box, and from x = 20 it is most of it
    assert (area <= 600.0 - 12.0 + 1e-9).all() and area[0, 2] > 0.95 * 588.0      # beside the box: almost no shadow
    np.testing.assert_allclose(area[0, ::-1], area[0], rtol=1e-9)      # the scene is mirror-symmetric about x = 0
    # stride: K = ceil(T / stride), pose k = sample k * stride
    rev2, area2 = oracle.future_visibility(x, y, 2, dirs, 60.0, g.edges, corn.reshape(-1, 8), flags, occ, s["x0"], s["y0"],
                                           s["cs"], 0, 0, s["nx"])
    assert np.array_equal(rev2[0], rev[0, ::2]) and np.array_equal(area2[0], area[0, ::2])


@pytest.mark.parametrize("n_rays", [97, 181, 720, 5, 4])
def test_every_direction_of_a_full_fan_has_a_sector(oracle, n_rays):
    """an empty hall much larger than the sensor range: every cell centre within the polygon of hit points is visible,
    whatever the parity of the ray count (a full fan searched in halves loses a sliver next to ray n/2 when n is odd)"""
    g = room(-40, 40, -40, 40)
    ego = np.array([0.13, -0.21])
    r = 20.0
    s = _grid_for(oracle, g, ego, 0.4, r, n_rays=n_rays)
    iy, ix = np.mgrid[0:s["ny"], 0:s["nx"]]
    cx, cy = s["x0"] + (ix + 0.5) * s["cs"], s["y0"] + (iy + 0.5) * s["cs"]
    d = np.hypot(cx - ego[0], cy - ego[1])
    vis = (s["cls"] & 2) != 0
    inner = d <= r * math.cos(math.pi / n_rays) - 1e-9           # inside the inscribed circle of the ray polygon
    assert vis[inner].all()
    assert not vis[d > r].any()


def test_edge_lines_chain_collinear_pieces_and_stop_at_corners_and_junctions():
    """an L-shaped kerb sampled every metre: two chains (one per leg); a third piece touching the corner makes the
    corner a junction; a detached piece is its own chain"""
    leg1 = np.array([[x, 0.0, x + 1.0, 0.0] for x in range(10)], float)            # (0,0) -> (10,0)
    leg2 = np.array([[10.0, y, 10.0, y + 1.0] for y in range(5)], float)           # (10,0) -> (10,5)
    far = np.array([[30.0, 30.0, 31.0, 30.0]])
    e = np.concatenate((leg1, leg2, far))
    lab = S.edge_lines(e)
    assert len(set(lab[:10])) == 1 and len(set(lab[10:15])) == 1 and lab[0] != lab[10] and lab[15] not in (lab[0], lab[10])
    # reversed piece orientation does not matter
    e2 = e.copy()
    e2[3] = e2[3][[2, 3, 0, 1]]
    assert len(set(S.edge_lines(e2)[:10])) == 1
    # a junction (three pieces meeting at (5,0)) stops the chain there
    e3 = np.concatenate((e, np.array([[5.0, 0.0, 5.0, -3.0]])))
    lab3 = S.edge_lines(e3)
    assert len(set(lab3[:5])) == 1 and len(set(lab3[5:10])) == 1 and lab3[0] != lab3[5]
    # a slight bend (1e-6 rad) is not a straight continuation at the default tolerance
    bent = np.array([[0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 2.0, 1e-6]])
    assert len(set(S.edge_lines(bent))) == 2 and len(set(S.edge_lines(bent, sin_tol=1e-5))) == 1


def test_spatial_order_is_a_permutation_that_groups_neighbours():
    rng = np.random.default_rng(3)
    p = rng.uniform(0, 500, (2000, 2))
    e = np.concatenate((p, p + rng.normal(0, 1, (2000, 2))), axis=1)
    o = S.spatial_order(e)
    assert sorted(map(tuple, o)) == sorted(map(tuple, e))                   # same pieces
    def mean_box(a):
        a = a[: len(a) // 64 * 64].reshape(-1, 64, 4)
        return float(np.mean((a[:, :, [0, 2]].max((1, 2)) - a[:, :, [0, 2]].min((1, 2))) *
                             (a[:, :, [1, 3]].max((1, 2)) - a[:, :, [1, 3]].min((1, 2)))))
    assert mean_box(o) < 0.1 * mean_box(e)                                  # 64 consecutive pieces cover a small patch
