"""The reference's spawn rules restated on the cell classes (oracle/fo_spawn_rules_ref.py, the checker of the device
implementation): known-answer scenes evaluated with the oracle's ray fan / cell grid on the CPU.  PARITY UNPINNED vs the reference (GEOS absent): these tests are the pin."""
import math

import numpy as np
import pytest

from frenetix_occlusion import scenario as S
from frenetix_occlusion.sensor_model import CellWindow
from oracle.fo_spawn_rules_ref import CellView, SpawnRules, segment_rect_distance
from frenetix_occlusion.utils.curvilinear import PolylineCS, curvature
from frenetix_occlusion.utils.fo_obstacle import FOObstacles

CFG = {"spawn_locator": {"spawn_points_behind_turn": True, "spawn_point_behind_static_obstacle": True,
                         "max_static_spawn_points": 1},
       "agent_manager": {"pedestrian": {"width": 0.5, "length": 0.3}}}


def _view(oracle, lanelets, obstacles, ego, yaw, r=50.0, cs=0.5):
    g = S.MapGeometry.from_lanelets(lanelets)
    xy = g.poly_xy
    x0, y0 = math.floor((xy[:, 0].min() - 1) / cs) * cs, math.floor((xy[:, 1].min() - 1) / cs) * cs
    nx, ny = int(math.ceil((xy[:, 0].max() + 1 - x0) / cs)), int(math.ceil((xy[:, 1].max() + 1 - y0) / cs))
    raster = oracle.road_raster(g.poly_off, g.poly_xy, x0, y0, cs, nx, ny)
    obs = FOObstacles(obstacles)
    obs.update(0)
    corn, cen, flags = obs.arrays() if len(obs) else (np.zeros((0, 4, 2)), np.zeros((0, 2)), np.zeros(0, np.uint8))
    dirs = oracle.ray_dirs(720, yaw)
    rng, hid, _ = oracle.raycast(g.edges, corn, flags, ego, dirs, r)
    cls, _ = oracle.grid(raster, x0, y0, cs, 0, 0, nx, ny, ego, np.array([math.cos(yaw), math.sin(yaw)]), r, True, dirs, rng)
    if len(obs):
        vis = oracle.obstacle_visibility(g.edges, corn, cen, flags, ego, r, True, dirs)
        for o, v in zip(obs, vis):
            o.current_visible = bool(v)
    lane_yaw = S.lane_yaw_raster(lanelets, x0, y0, cs, nx, ny)

    def lane_yaw_at(p):
        ix, iy = int((p[0] - x0) / cs), int((p[1] - y0) / cs)
        if not (0 <= ix < nx and 0 <= iy < ny) or np.isnan(lane_yaw[iy, ix]):
            return None
        return float(lane_yaw[iy, ix])

    def lanelet_of(p):
        for ll in lanelets:
            if S.points_in_polygon(np.asarray(p, float).reshape(1, 2), ll.polygon)[0]:
                return ll
        return None
    return CellView(cls, CellWindow(x0, y0, cs, 0, 0, nx, ny)), obs, lane_yaw_at, lanelet_of


def _straight(lid, x0, x1, y_lo, y_hi, n=41):
    xs = np.linspace(x0, x1, n)
    return S.Lanelet(lid, np.stack((xs, np.full(n, y_hi)), -1), np.stack((xs, np.full(n, y_lo)), -1))


def test_polyline_cs_round_trip_and_domain():
    path = np.array([[0.0, 0.0], [10.0, 0.0], [10.0, 10.0]])
    cs = PolylineCS(path)
    np.testing.assert_allclose(cs.convert_to_curvilinear_coords(4.0, 2.0), [4.0, 2.0])
    np.testing.assert_allclose(cs.convert_to_curvilinear_coords(12.0, 5.0), [15.0, -2.0])
    np.testing.assert_allclose(cs.convert_to_cartesian_coords(15.0, -2.0), [12.0, 5.0])
    with pytest.raises(ValueError):
        cs.convert_to_curvilinear_coords(-1.0, 0.0)
    with pytest.raises(ValueError):
        cs.convert_to_cartesian_coords(25.0, 0.0)
    pts = cs.convert_list_of_points_to_curvilinear_coords([np.array([[1.0], [1.0]]), np.array([[9.0], [-1.0]])], 4)
    np.testing.assert_allclose(pts, [[1.0, 1.0], [9.0, -1.0]])


def test_ego_intention_from_curvature():
    s = np.linspace(0, 40, 81)
    straight = np.stack((s, np.zeros_like(s)), -1)
    assert SpawnRules.ego_intention(straight) == "straight ahead"
    ang = np.linspace(0, math.pi / 2, 40)
    left = np.concatenate((np.stack((np.linspace(0, 10, 20), np.zeros(20)), -1),
                           np.stack((10 + 8 * np.sin(ang), 8 - 8 * np.cos(ang)), -1)))       # radius 8 m: kappa 0.125
    assert SpawnRules.ego_intention(left) == "left turn"
    right = left * np.array([1.0, -1.0])
    assert SpawnRules.ego_intention(right) == "right turn"
    assert abs(curvature(left)[30]) == pytest.approx(0.125, rel=0.05)


def test_segment_rectangle_distance():
    rect = np.array([[0, 0], [0, 2], [4, 2], [4, 0.0]])
    assert segment_rect_distance([5, 1], [7, 1], rect) == pytest.approx(1.0)
    assert segment_rect_distance([2, -3], [2, 5], rect) == 0.0
    assert segment_rect_distance([1, 1], [1, 1], rect) == 0.0
    assert segment_rect_distance([5, 3], [6, 4], rect) == pytest.approx(math.sqrt(2.0))


def test_pedestrian_behind_a_parked_car(oracle):
    """two-lane straight road, a parked car on the right edge 15 m ahead: one pedestrian spawn point in the car's
    shadow next to it, heading across the road (lane heading + 90 deg), source names the obstacle"""
    lanes = [_straight(1, -10, 70, -3.5, 0.0), _straight(2, -10, 70, 0.0, 3.5)]
    car = S.Obstacle(77, "static", "parkedVehicle", 4.5, 1.8, 0, np.array([17.0, -2.4, 0.0, 0.0]), np.zeros((0, 4)))
    ego = np.array([0.0, -1.0])
    view, obs, lane_yaw_at, lanelet_of = _view(oracle, lanes, [car], ego, 0.0)
    assert next(iter(obs)).current_visible
    path = np.stack((np.linspace(-5, 65, 141), np.full(141, -1.0)), -1)
    cs = PolylineCS(path)
    rules = SpawnRules(CFG, path, cs, lane_yaw_at, lanelet_of, obs)
    ego_cl = cs.convert_to_curvilinear_coords(ego[0], ego[1])
    pts = rules.find(view, ego, ego_cl, 8.0)
    assert rules.last_intention == "straight ahead"
    assert len(pts) == 1
    sp = pts[0]
    assert sp.agent_type == "Pedestrian" and sp.source == "behind static obstacle 77"
    assert sp.orientation == pytest.approx(math.pi / 2)
    assert view.class_at(sp.position) & 4 and not view.class_at(sp.position) & 2            # occluded, not visible
    # on the cross line just behind the far end of the car (s_max = 19.25 + 0.8), on the kerb side of the car
    assert sp.position[0] == pytest.approx(17.0 + 2.25 + 0.8, abs=0.3) and sp.position[1] < -1.0
    # nothing when the car is beyond 30 m, behind the ego, or invisible
    for x in (45.0, -6.0):
        far = S.Obstacle(78, "static", "parkedVehicle", 4.5, 1.8, 0, np.array([x, -2.4, 0.0, 0.0]), np.zeros((0, 4)))
        v2, o2, ly, lo = _view(oracle, lanes, [far], ego, 0.0)
        assert SpawnRules(CFG, path, cs, ly, lo, o2).find(v2, ego, ego_cl, 8.0) == []
    next(iter(obs)).current_visible = False
    assert rules.find(view, ego, ego_cl, 8.0) == []


def test_pedestrian_behind_a_right_turn(oracle):
    """T junction: the ego turns right into a side street hidden by the corner; the rule puts a pedestrian on the
    reference path's right side where the path first enters the occluded area"""
    main = [_straight(1, -40, 40, -3.5, 0.0), _straight(2, -40, 40, 0.0, 3.5)]
    ys = np.linspace(-3.5, -43.5, 41)
    side = [S.Lanelet(3, np.stack((np.full(41, 13.5), ys), -1), np.stack((np.full(41, 10.0), ys), -1)),
            S.Lanelet(4, np.stack((np.full(41, 17.0), ys), -1), np.stack((np.full(41, 13.5), ys), -1))]
    ego = np.array([-5.0, -1.75])
    view, obs, lane_yaw_at, lanelet_of = _view(oracle, main + side, [], ego, 0.0)
    ang = np.linspace(0, math.pi / 2, 30)
    path = np.concatenate((np.stack((np.linspace(-30, 7.75, 76), np.full(76, -1.75)), -1),
                           np.stack((7.75 + 4.0 * np.sin(ang), -5.75 + 4.0 * np.cos(ang)), -1)[1:],
                           np.stack((np.full(60, 11.75), np.linspace(-6.25, -36.0, 60)), -1)))
    cs = PolylineCS(path)
    rules = SpawnRules(CFG, path, cs, lane_yaw_at, lanelet_of, obs)
    ego_cl = cs.convert_to_curvilinear_coords(ego[0], ego[1])
    pts = rules.find(view, ego, ego_cl, 6.0)
    assert rules.last_intention == "right turn"
    assert len(pts) == 1 and pts[0].source == "right turn" and pts[0].agent_type == "Pedestrian"
    p = pts[0].position
    assert 10.0 < p[0] < 11.75 and p[1] < -4.0                     # in the side street, right of the path (d = -1)
    assert not view.disc_touches(p, 0.5, 2)                         # shifted until the 0.5 m disc leaves the visible area
    assert pts[0].orientation is None                               # derived later: towards the lane centre (agent.py:475)
    # going straight on the main road instead: no turn rule
    straight = np.stack((np.linspace(-30, 40, 141), np.full(141, -1.75)), -1)
    r2 = SpawnRules(CFG, straight, PolylineCS(straight), lane_yaw_at, lanelet_of, obs)
    assert r2.find(view, ego, PolylineCS(straight).convert_to_curvilinear_coords(ego[0], ego[1]), 6.0) == []


def test_car_and_bicycle_behind_an_oncoming_truck(oracle):
    """two-way road without an intersection: an oncoming truck hides the stretch of the oncoming lane behind it; the
    dynamic-obstacle rule fits a car (5.5 x 2.5 m) and then a bicycle (2 x 1 m) into the truck's shadow on that lane"""
    xs = np.linspace(-10, 70, 41)
    lane1 = S.Lanelet(1, np.stack((xs, np.zeros(41)), -1), np.stack((xs, np.full(41, -3.5)), -1))
    lane2 = S.Lanelet(2, np.stack((xs[::-1], np.zeros(41)), -1), np.stack((xs[::-1], np.full(41, 3.5)), -1))
    lane1.adj_left, lane1.adj_left_same_direction = 2, False
    lane2.adj_left, lane2.adj_left_same_direction = 1, False
    truck = S.Obstacle(31, "dynamic", "truck", 9.0, 3.2, 0, np.array([20.0, 1.75, math.pi, 8.0]), np.zeros((0, 4)))
    ego = np.array([0.0, -1.75])
    view, obs, lane_yaw_at, lanelet_of = _view(oracle, [lane1, lane2], [truck], ego, 0.0)
    assert next(iter(obs)).current_visible
    path = np.stack((np.linspace(-5, 65, 141), np.full(141, -1.75)), -1)
    cs = PolylineCS(path)
    rules = SpawnRules(CFG | {"spawn_locator": CFG["spawn_locator"] | {"spawn_point_behind_dynamic_obstacle": True,
                                                                       "max_dynamic_spawn_points": 1}},
                       path, cs, lane_yaw_at, lanelet_of, obs, lanelets=[lane1, lane2], intersections=[])
    ego_cl = cs.convert_to_curvilinear_coords(ego[0], ego[1])
    pts = [p for p in rules.find(view, ego, ego_cl, 8.0, 0.0) if p.source == "behind_dynamic_obstacle"]
    kinds = [p.agent_type for p in pts]
    assert kinds == ["Car", "Bicycle"]
    for p in pts:
        assert 25.5 < p.position[0] < 32.0 and 0.0 < p.position[1] < 3.5          # on the oncoming lane, behind the truck
        assert not view.class_at(p.position) & 2
    # a pedestrian-type or bicycle-type obstacle never triggers the rule; neither does a truck driving the same way
    for typ, yaw in (("bicycle", math.pi), ("truck", 0.0)):
        other = S.Obstacle(32, "dynamic", typ, 9.0, 2.5, 0, np.array([20.0, 1.75, yaw, 8.0]), np.zeros((0, 4)))
        v2, o2, ly, lo = _view(oracle, [lane1, lane2], [other], ego, 0.0)
        r2 = SpawnRules(CFG, path, cs, ly, lo, o2, lanelets=[lane1, lane2], intersections=[])
        got = [p for p in r2.find(v2, ego, ego_cl, 8.0, 0.0) if p.source == "behind_dynamic_obstacle"]
        if typ == "bicycle":
            assert got == []
        else:                                     # same direction: the global occluded area is used instead of the wedge
            assert all(p.agent_type in ("Car", "Bicycle") for p in got)
