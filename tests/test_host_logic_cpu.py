"""Host-side logic that needs no GPU: metric ordering, trajectory packing, obstacle state cache, ray fan definition,
cell-window arithmetic, config / coefficient loading, agent-manager conventions."""
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_metric_dependency_order_matches_the_reference_rules():
    """metric.py:125-147: wttc pulls ttc to the front, ttc/ttce/be pull dce to the front, hr pulls cp to the front"""
    from frenetix_occlusion.metrics.metric import check_required_metrics as order
    assert order(["hr", "ttc", "ttce", "dce", "wttc", "cp"]) == ["cp", "dce", "ttc", "hr", "ttce", "wttc"]
    assert order(["hr", "ttc"]) == ["cp", "dce", "hr", "ttc"]
    assert order(["wttc"]) == ["dce", "ttc", "wttc"]
    assert order(["be"]) == ["dce", "be"]
    assert order(["cp"]) == ["cp"] and order([]) == []
    lst = ["ttce"]
    assert order(lst) == ["dce", "ttce"] and lst == ["ttce"]          # the caller's list is not mutated


def test_trajectory_packing_accepts_objects_and_dicts():
    from frenetix_occlusion.metrics.metric import trajectories_to_arrays
    mk = lambda off: SimpleNamespace(cartesian=SimpleNamespace(x=np.arange(5.0) + off, y=np.zeros(5), theta=np.zeros(5),
                                                              v=np.ones(5), a=np.zeros(5)))
    arr = trajectories_to_arrays([mk(0.0), mk(10.0)])
    assert set(arr) == {"x", "y", "theta", "v", "a"} and arr["x"].shape == (2, 5) and arr["x"][1, 0] == 10.0
    d = {"x": np.zeros((3, 4))}
    assert trajectories_to_arrays(d) is d
    bad = mk(0.0)
    bad.cartesian.x = np.arange(4.0)
    with pytest.raises(ValueError):
        trajectories_to_arrays([mk(0.0), bad])


def test_obstacle_cache_follows_the_reference_timestep_semantics():
    """fo_obstacle.py:79-116: rel = step - initial; 0 -> initial state; >= 1 -> state_list[rel-1]; else absent"""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    obs = FOObstacles(sc.obstacles)
    assert len(obs) == 5
    o = sc.obstacles[0]
    obs.update(o.initial_time_step)
    first = next(iter(obs))
    np.testing.assert_array_equal(first.current_pos, o.initial[:2])
    obs.update(o.initial_time_step + 3)
    np.testing.assert_array_equal(first.current_pos, o.states[2, :2])
    c = first.current_corner_points
    assert c.shape == (4, 2)
    np.testing.assert_allclose(np.linalg.norm(c[0] - c[1]), o.width, atol=1e-12)       # (-l/2,-w/2) -> (-l/2,+w/2)
    np.testing.assert_allclose(np.linalg.norm(c[1] - c[2]), o.length, atol=1e-12)
    obs.update(o.initial_time_step + len(o.states) + 5)
    assert first.current_pos is None
    corn, cen, flags = obs.arrays()
    assert flags[0] == 0 and corn.shape == (5, 4, 2)
    # bicycles exist but do not occlude (sensor_model.py:177)
    bike = S.Obstacle(99, "dynamic", "bicycle", 2.0, 0.9, 0, np.array([1.0, 2.0, 0.3, 5.0]), np.zeros((0, 4)))
    obs.add(bike)
    obs.update(0)
    assert obs.arrays()[2][-1] == 1


def test_duck_typed_commonroad_objects_are_accepted():
    from frenetix_occlusion import scenario as S
    st = lambda x, y, yaw, v, ts=0: SimpleNamespace(position=np.array([x, y]), orientation=yaw, velocity=v, time_step=ts)
    cr = SimpleNamespace(obstacle_id=7, obstacle_type=SimpleNamespace(value="car"), obstacle_role=SimpleNamespace(value="dynamic"),
                         obstacle_shape=SimpleNamespace(length=4.5, width=1.8), initial_state=st(1, 2, 0.1, 3, 4),
                         prediction=SimpleNamespace(trajectory=SimpleNamespace(state_list=[st(2, 2, 0.1, 3), st(3, 2, 0.1, 3)])))
    o = S.obstacle_from_commonroad(cr)
    assert (o.obstacle_id, o.role, o.obstacle_type, o.initial_time_step) == (7, "dynamic", "car", 4)
    assert o.pose_at(5)[0] == 2.0 and o.pose_at(7) is None
    ll = SimpleNamespace(lanelet_id=3, left_vertices=np.array([[0, 1], [5, 1.0]]), right_vertices=np.array([[0, -1], [5, -1.0]]),
                         successor=[4], predecessor=[], adj_left=None, adj_right=None)
    out = S.lanelets_of(SimpleNamespace(lanelets=[ll]))
    assert out[0].lanelet_id == 3 and out[0].successors == [4] and out[0].polygon.shape == (4, 2)


def test_ray_fan_definition():
    from frenetix_occlusion.sensor_model import ray_dirs
    d = ray_dirs(720, 0.3, 360.0)
    assert d.shape == (720, 2)
    np.testing.assert_allclose(np.hypot(d[:, 0], d[:, 1]), 1.0, atol=1e-15)
    np.testing.assert_allclose(np.arctan2(d[1, 1], d[1, 0]) - 0.3, 2 * math.pi / 720, atol=1e-12)      # 0.5 degrees
    f = ray_dirs(181, 1.0, 90.0)
    np.testing.assert_allclose(np.arctan2(f[0, 1], f[0, 0]), 1.0 - math.pi / 4, atol=1e-12)
    np.testing.assert_allclose(np.arctan2(f[-1, 1], f[-1, 0]), 1.0 + math.pi / 4, atol=1e-12)
    cross = f[:-1, 0] * f[1:, 1] - f[:-1, 1] * f[1:, 0]
    assert (cross > 0).all()                                                                             # counter-clockwise


def test_cell_window_round_trip():
    from frenetix_occlusion.sensor_model import CellWindow
    w = CellWindow(x0=-10.0, y0=4.0, cs=0.5, ix0=3, iy0=-2, nx=7, ny=5)
    idx = np.arange(35)
    c = w.centers(idx)
    assert np.array_equal(w.cell_of(c), idx)
    assert c[0, 0] == -10.0 + 3.5 * 0.5 and c[0, 1] == 4.0 - 1.5 * 0.5
    assert w.cell_of([[100.0, 100.0]])[0] == -1


def test_default_config_and_coefficients_load():
    from frenetix_occlusion import interface
    from frenetix_occlusion.metrics.metric import load_harm_coeff
    from frenetix_occlusion.sweep import DEFAULT_HARM_COEFF
    cfg = interface.FOInterface._load_config(None)          # the reference computes this default but does not use it
    assert cfg["sensor_model"] == {"sensor_radius": 50, "sensor_angle": 360}
    assert cfg["metrics"]["activated_metrics"] == ["hr", "ttc", "ttce", "dce", "wttc", "cp"]
    assert cfg["agent_manager"]["prediction"]["variance_factor"] == 1.05
    assert set(cfg["agent_manager"]) >= {"bicycle", "car", "truck", "pedestrian", "prediction"}
    assert load_harm_coeff() == DEFAULT_HARM_COEFF            # harm_params.json entries the reference reads


def test_no_gpu_interface_refuses_to_start():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario2_geometry.npz"))
    with pytest.raises(RuntimeError):
        interface.FOInterface(sc, np.zeros((2, 2)), SimpleNamespace(length=4.5, width=1.6, wb_rear_axle=1.4, mass=1000.0,
                                                                     a_max=11.5), 0.1)


def _bare_agent_manager(ids=()):
    """FOAgentManager without a GPU: only the id bookkeeping is exercised"""
    from frenetix_occlusion.agent import FOAgentManager
    sc = SimpleNamespace(obstacles=[SimpleNamespace(obstacle_id=i) for i in ids])
    return FOAgentManager(sc, np.array([[0.0, 0.0], [10.0, 0.0]]), {}, 0, device="cpu")


def test_agent_ids_are_released_every_step():
    """agent.py:189-199 draws from 1001 values and never hands them back; 32 phantoms per step must not drain the
    pool (it used to spin forever at step 31)"""
    am = _bare_agent_manager(ids=(10005, 10500, 42))
    for step in range(300):
        ids = [am._create_id() for _ in range(32)]
        assert len(set(ids)) == 32 and all(10000 <= i <= 11000 for i in ids)
        assert 10005 not in ids and 10500 not in ids
        assert len(am.all_obstacle_id) == 3 + 32
        am.reset()
        assert sorted(am.all_obstacle_id) == [42, 10005, 10500]


def test_external_predictions_become_sweep_slots_keyed_by_obstacle_id():
    """EXTENSION (FOAgentManager.set_external_predictions): real agents' predictions join the agent axis after the
    phantoms, in ascending obstacle id; ragged list lengths are cut to the shortest; reset() drops them"""
    am = _bare_agent_manager(ids=(7, 3))
    L = 12
    cov = np.tile(np.array([[0.2, 0.05], [0.05, 0.3]]), (L, 1, 1))
    mk = lambda x0, n: {"pos_list": np.c_[x0 + np.arange(n), np.zeros(n)], "v_list": np.ones(n), "orientation_list": np.zeros(n),
                        "cov_list": cov[:n], "shape": {"length": 4.5, "width": 1.8}}
    p7, p3 = mk(10.0, L), mk(20.0, L - 2)
    p3["v_list"] = np.ones(L - 4)                          # a list that ends early
    am.set_external_predictions({7: p7, 3: p3, 9: mk(0.0, 0)}, types={7: "truck", 3: "Bicycle", 9: "car"})
    assert am.has_phantoms() and am.n_slots() == 2         # the empty prediction is dropped
    pos, yaw, v, c, shape, raw, typ, ln = am.sweep_arrays()
    from frenetix_occlusion._native import TYPE_CODES
    assert typ.tolist() == [TYPE_CODES["bicycle"], TYPE_CODES["truck"]] and ln.tolist() == [L - 4, L]
    assert pos.shape == (2, L, 2) and float(pos[0, 0, 0]) == 20.0 and float(pos[1, L - 1, 0]) == 10.0 + L - 1
    assert float(c[1, 3, 0, 1]) == 0.05 and shape.tolist() == raw.tolist() == [[4.5, 1.8], [4.5, 1.8]]
    preds = am.predictions
    assert list(preds) == [3, 7] and am.prediction_slots == [(3, 0), (7, 1)]
    am.reset()
    assert not am.has_phantoms() and am.sweep_arrays() is None


def test_agent_ids_of_scenario_agents_stay_taken_and_pool_exhaustion_raises():
    from frenetix_occlusion.agent import FOAgentManager, PhantomAgent
    am = _bare_agent_manager()
    keep = am._create_id()
    am.real_agents.append(PhantomAgent(keep, "Pedestrian", np.zeros(2), 0.0, 1.4, 0.5, 0.5))   # add_to_scenario=True
    am.reset()
    assert am.all_obstacle_id == [keep]
    # more agents than the reference's range holds: spills into the remaining five-digit ids, still unique
    ids = [am._create_id() for _ in range(1200)]
    assert len(set(ids)) == 1200 and keep not in ids and all(10000 <= i <= 99999 for i in ids)
    am.reset()
    am.ID_RANGE, am.ID_RANGE_WIDE = (10000, 10003), (10004, 10005)
    got = [am._create_id() for _ in range(6 - (10000 <= keep <= 10005))]
    assert len(set(got)) == len(got)
    with pytest.raises(RuntimeError):
        am._create_id()


def test_served_from_batch_checks_object_identity():
    """the per-trajectory cache is keyed by id(); a new object at a recycled address must not be served another
    trajectory's result (metric.py of this package, evaluate_metrics)"""
    from frenetix_occlusion.metrics.metric import Metric
    m = Metric.__new__(Metric)
    m.metrics = ["dce"]
    m.agent_manager = SimpleNamespace(has_phantoms=lambda: True)
    served = []
    m._batch = SimpleNamespace(mode="full", result_dict=lambda i: served.append(i) or ({"i": i}, True))
    objs = [SimpleNamespace(tag=i) for i in range(4)]
    m._batch_objs = list(objs)
    m._batch_ids = {id(t): i for i, t in enumerate(objs)}
    assert m.evaluate_metrics(objs[2]) == ({"i": 2}, True)
    stranger = SimpleNamespace(tag=99)
    m._batch_ids[id(stranger)] = 1                     # what address reuse after the planner dropped its list looks like
    fresh = []
    m.evaluate_batch = lambda trajs, mode="full": fresh.append(trajs) or SimpleNamespace(result_dict=lambda i: ({"new": 1}, False))
    assert m.evaluate_metrics(stranger) == ({"new": 1}, False)
    assert fresh and fresh[0][0] is stranger and served == [2]
    m.invalidate()
    assert m._batch is None and m._batch_objs == [] and m._batch_ids == {}


def test_list_views_follow_the_documented_buffer_layout():
    """include/fo_hip.h: cp [A][T-1][M] at offset 0, (ego harm, obstacle harm) pairs [A][T-1][M][2] at n, (ego risk,
    obstacle risk) pairs at 3 n; sweep.list_views returns strided views in the order of _native.LST"""
    from frenetix_occlusion import _native as N
    from frenetix_occlusion.sweep import list_views
    A, Tm1, M = 3, 4, 5
    n = A * Tm1 * M
    raw = np.arange(5 * n, dtype=np.float64)
    views = list_views(raw, A, Tm1, M)
    assert [v.shape for v in views] == [(A, Tm1, M)] * 5 and len(views) == N.NL
    k, t, m = 2, 1, 3
    i = (k * Tm1 + t) * M + m
    assert views[N.LST["cp"]][k, t, m] == raw[i]
    assert views[N.LST["ego_harm"]][k, t, m] == raw[n + 2 * i] and views[N.LST["obst_harm"]][k, t, m] == raw[n + 2 * i + 1]
    assert views[N.LST["ego_risk"]][k, t, m] == raw[3 * n + 2 * i] and views[N.LST["obst_risk"]][k, t, m] == raw[3 * n + 2 * i + 1]
    views[N.LST["obst_risk"]][k, t, m] = -1.0            # views, not copies
    assert raw[3 * n + 2 * i + 1] == -1.0
    torch = pytest.importorskip("torch")
    tv = list_views(torch.arange(5 * n, dtype=torch.float64), A, Tm1, M)
    assert float(tv[N.LST["ego_harm"]][k, t, m]) == n + 2 * i
    empty = list_views(np.zeros(0), 0, Tm1, M)
    assert all(v.shape == (0, Tm1, M) for v in empty)


def test_dependency_closure_equals_the_references():
    """which metrics get evaluated for a list of activated ones (metric.py:125-147), as the reference's own class
    reported them for the configurations of tests/golden/thresholds.npz"""
    from golden_util import load_threshold_case
    from frenetix_occlusion.metrics.metric import check_required_metrics
    for activated, _, evaluated, _ in load_threshold_case()[4]:
        assert sorted(check_required_metrics(list(activated))) == evaluated, activated


def test_pedestrian_heading_without_a_given_orientation_known_answers():
    """agent.py:475-481 + helper_functions.py:38-76: a pedestrian whose spawn point brings no orientation heads for the
    closest point of its curve -- the ego's reference path (mode 'ref_path') or the centre line of the lanelet it stands
    on (mode 'lane_center', agent.py:459-467; off-lanelet falls back to the reference path, Q12) -- as an angle in
    [0, 2 pi) (angle_between_positive).  Closed-form cases; the shapely projection itself is not available here."""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.agent import FOAgentManager
    xs = np.linspace(0.0, 40.0, 21)
    lane = S.Lanelet(1, np.stack((xs, np.full(21, 6.0)), -1), np.stack((xs, np.full(21, 2.0)), -1))     # centre line y = 4
    sc = S.Scenario(0.1, [lane], [])
    path = np.array([[0.0, 0.0], [10.0, 0.0], [10.0, 10.0]])                                           # an L: +x, then +y
    cfg = {"pedestrian": {"length": 0.3, "width": 0.5, "default_velocity": 1.4},
           "prediction": {"variance_factor": 1.05, "size_factor_length_s": 1.2, "size_factor_width_s": 1.3,
                          "size_factor_length_l": 1.4, "size_factor_width_l": 2.5}}
    am = FOAgentManager(sc, path, cfg, 0, device="cpu")
    h = am._heading_towards_path
    assert h(np.array([4.0, 3.0])) == pytest.approx(1.5 * math.pi)          # left of the first leg: straight down to it
    assert h(np.array([4.0, -2.0])) == pytest.approx(0.5 * math.pi)         # right of it: straight up
    assert h(np.array([13.0, 5.0])) == pytest.approx(math.pi)               # right of the second leg: towards -x
    assert h(np.array([12.0, -2.0])) == pytest.approx(0.75 * math.pi)       # outside the corner: towards the corner vertex
    assert h(np.array([-3.0, 4.0])) == pytest.approx(2.0 * math.pi - math.atan2(4.0, 3.0))   # before the start: towards the first vertex
    assert h(np.array([10.0, 14.0])) == pytest.approx(1.5 * math.pi)        # beyond the end: back towards the last vertex
    assert h(np.array([4.0, 0.0])) == 0.0                                   # on the curve: no direction -> 0
    # through add_agent: ref_path mode heads for the reference path, lane_center mode for the lanelet's centre line ...
    a = am.add_agent(np.array([20.0, 5.5]), agent_type="Pedestrian", mode="ref_path")
    assert a.initial_orientation == pytest.approx(math.pi)                 # closest point (10, 5.5) on the second leg
    b = am.add_agent(np.array([20.0, 5.5]), agent_type="Pedestrian", mode="lane_center")
    assert b.initial_orientation == pytest.approx(1.5 * math.pi)           # centre line y = 4 lies below
    # ... off-lanelet the lane_center mode falls back to the reference path instead of returning no trajectory (Q12)
    c = am.add_agent(np.array([4.0, -2.0]), agent_type="Pedestrian", mode="lane_center")
    assert c.initial_orientation == pytest.approx(0.5 * math.pi) and len(c.predictions) == 1
    # a given orientation is taken as is (interface.py:192-198 passes the spawn point's)
    d = am.add_agent(np.array([4.0, -2.0]), agent_type="Pedestrian", orientation=0.3)
    assert d.initial_orientation == 0.3
    # the velocity components are rounded to 3 decimals (agent.py:492-493, Q12)
    p = c.predictions[0]["pos_list"]
    np.testing.assert_allclose(p[10] - p[0], [round(1.4 * math.cos(c.initial_orientation), 3), round(1.4 * math.sin(c.initial_orientation), 3)], atol=1e-12)


HEADING_KAT = [((4.0, 3.0), 1.5 * math.pi), ((4.0, -2.0), 0.5 * math.pi), ((13.0, 5.0), math.pi), ((12.0, -2.0), 0.75 * math.pi),
               ((-3.0, 4.0), 2.0 * math.pi - math.atan2(4.0, 3.0)), ((10.0, 14.0), 1.5 * math.pi),
               ((8.0, 3.0), 0.0), ((9.5, 0.25), 1.5 * math.pi)]      # inside the corner: the nearer leg wins


def test_pedestrian_heading_known_answers_hold_for_the_oracle_as_well(oracle):
    """the same closed-form cases (an L-shaped path: left / right of a leg, inside and outside the kink, beyond both ends)
    for the checker of the device kernels, oracle/fo_oracle_scene.c fo_oracle_spawn_headings -- and for the host path, so
    that the three statements of agent.py:475-481 (host numpy, C oracle, HIP kernel: tests/test_rules_step_gpu.py) hang on
    one list of known answers"""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.agent import FOAgentManager
    path = np.array([[0.0, 0.0], [10.0, 0.0], [10.0, 10.0]])
    pos = np.array([p for p, _ in HEADING_KAT])
    want = np.array([a for _, a in HEADING_KAT])
    got = oracle.spawn_headings(pos, np.full(len(pos), 4, dtype=np.int32), path)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
    cfg = {"pedestrian": {"length": 0.3, "width": 0.5, "default_velocity": 1.4},
           "prediction": {"variance_factor": 1.05, "size_factor_length_s": 1.2, "size_factor_width_s": 1.3,
                          "size_factor_length_l": 1.4, "size_factor_width_l": 2.5}}
    am = FOAgentManager(S.Scenario(0.1, [], []), path, cfg, 0, device="cpu")
    np.testing.assert_allclose([am._heading_towards_path(q) for q in pos], want, rtol=0, atol=1e-15)


def test_ego_intention_thresholds_on_circular_arcs():
    """spawn_locator.py:729-741: left turn if max curvature > 0.10, right turn if min < -0.10, by
    compute_curvature_from_polyline [ext: np.gradient of the coordinates w.r.t. the path length, twice].  On a circle sampled
    at equal steps the central differences return the radius exactly (every interior vertex: kappa), the one-sided ends
    kappa/2 and 3 kappa/4 -- so arcs of kappa = +-0.09 read "straight ahead" and +-0.11 a turn at the reference path's
    spacings, and the decision does not hinge on the ends."""
    from frenetix_occlusion.utils.curvilinear import curvature
    for kappa in (0.09, -0.09, 0.11, -0.11):
        for ds in (0.2, 0.5, 1.0):
            n = int(40.0 / ds) + 1                      # the 40 m window of spawn_locator.py:678-693
            s = np.arange(n) * ds
            R, sg = 1.0 / abs(kappa), math.copysign(1.0, kappa)
            xy = np.stack((R * np.sin(s / R), sg * R * (1.0 - np.cos(s / R))), -1)
            k = curvature(xy)
            np.testing.assert_allclose(k[2:-2], kappa, rtol=1e-9)
            assert abs(k[0]) == pytest.approx(abs(kappa) / 2, rel=2e-3) and abs(k[1]) == pytest.approx(0.75 * abs(kappa), rel=2e-3)
            intention = "left turn" if k.max() > 0.10 else "right turn" if k.min() < -0.10 else "straight ahead"
            assert intention == {0.09: "straight ahead", -0.09: "straight ahead", 0.11: "left turn", -0.11: "right turn"}[kappa]
    # a straight lead-in in front of the arc (what a planner's route looks like) does not change the decision
    lead = np.stack((np.linspace(-10.0, -0.5, 20), np.zeros(20)), -1)
    s = np.arange(41) * 0.5
    for kappa, want in ((0.11, "left turn"), (-0.11, "right turn"), (0.09, "straight ahead")):
        R, sg = 1.0 / abs(kappa), math.copysign(1.0, kappa)
        xy = np.concatenate((lead, np.stack((R * np.sin(s / R), sg * R * (1.0 - np.cos(s / R))), -1)))
        k = curvature(xy)
        assert ("left turn" if k.max() > 0.10 else "right turn" if k.min() < -0.10 else "straight ahead") == want


def test_lazy_spawn_points_behave_like_the_references_list():
    """FOInterface.spawn_points is a list (interface.py:186) that fills itself from the device on first use"""
    import json
    from frenetix_occlusion.spawn_locator import LazySpawnPoints, PhantomBatch
    calls = []
    x = LazySpawnPoints(lambda: (calls.append(1), [3, 1, 2])[1])
    assert isinstance(x, list) and not x.materialised and calls == []
    assert x + [4] == [3, 1, 2, 4] and [0] + x == [0, 3, 1, 2] and calls == [1]
    assert len(x) == 3 and x[1:] == [1, 2] and 2 in x and sorted(x) == [1, 2, 3] and x == [3, 1, 2] and bool(x)
    assert json.dumps(x) == "[3, 1, 2]" and calls == [1] and x.materialised
    assert not LazySpawnPoints(lambda: []) and LazySpawnPoints(lambda: []) == []
    # a list whose step's buffers were reused before anybody held it raises instead of returning the next step's points;
    # one that is still alive when the next step is queued is read back at that moment
    import weakref
    b = PhantomBatch(*([None] * 12))
    y = LazySpawnPoints(lambda: ["step 0"], b)
    b._pending = weakref.ref(y)
    b.invalidate()
    assert y.materialised and y == ["step 0"]
    z = LazySpawnPoints(lambda: ["step 1"], b)
    b.step += 1                         # (as if the list had not been registered)
    with pytest.raises(RuntimeError):
        len(z)


def test_a_refused_rule_point_count_raises_when_the_list_is_read():
    """fo_scene_spawn_rules answers a rule family that ran out of table space with a count of -1 (include/fo_hip.h): the
    host view of the step's head refuses it instead of handing out a shorter list"""
    torch = pytest.importorskip("torch")
    from frenetix_occlusion.spawn_locator import PhantomBatch
    A, S_, Rp = 8, 24, 8
    o1, o2 = A * 16, A * 24
    o3 = o2 + Rp * 64
    head = np.zeros(o3 + 8 + 4 * S_, np.uint8)

    def batch(rule_n):
        head[o3:o3 + 8].view(np.int32)[:] = (0, rule_n)
        h = torch.from_numpy(head.copy())
        return PhantomBatch(None, None, torch.zeros(A, 2, dtype=torch.float64), None, None, None, None, None, None, None,
                            torch.zeros(S_, dtype=torch.int32), None, R=3, head=h, n_cell_agents=0, n_rule_points=Rp)
    assert batch(3).host_head()["rule_n"] == 3 and batch(3).live_agents() == [0, 1, 2]
    assert batch(99).host_head()["rule_n"] == Rp            # (a count beyond the buffer: what fits)
    with pytest.raises(RuntimeError, match="table space"):
        batch(-1).host_head()
    with pytest.raises(RuntimeError, match="table space"):
        batch(-1).live_agents()


def test_lazy_result_dicts_never_leak_placeholders_through_dict_fast_paths():
    """The per-trajectory result is a dict SUBCLASS with lazily built values (metrics/metric.py); CPython copies a dict
    subclass through C shortcuts that bypass ``__getitem__`` unless ``__iter__`` is overridden.  Every way a planner may
    copy or merge the result must deliver built values (the reference returns plain dicts, metric.py:35-100)."""
    from frenetix_occlusion.metrics import metric as MM

    class Lazy(MM._LazyDict):
        def __init__(self, keys, depth=1):
            super().__init__((k, MM._UNBUILT) for k in keys)
            self.built, self.depth = [], depth

        def _build(self, key):
            self.built.append(key)
            return Lazy(("x", "y"), 0) if (self.depth and key == "hr") else ("built", key)

    plain = {"cp": ("built", "cp"), "hr": {"x": ("built", "x"), "y": ("built", "y")}, "wttc": ("built", "wttc")}
    mk = lambda: Lazy(("cp", "hr", "wttc"))
    assert isinstance(mk(), dict)
    assert dict(mk()) == plain and {**mk()} == plain and mk().copy() == plain
    d = {}
    d.update(mk())
    assert d == plain
    assert ({"a": 1} | mk()) == {"a": 1, **plain} and (mk() | {"a": 1}) == {**plain, "a": 1}
    m = mk()
    assert {**m}["hr"]["x"] == ("built", "x")
    assert m.pop("cp") == ("built", "cp") and "cp" not in m and m.pop("cp", 7) == 7
    with pytest.raises(KeyError):
        m.pop("cp")
    assert m.popitem() == ("wttc", ("built", "wttc")) and list(m) == ["hr"]
    assert m.setdefault("hr", 0) == {"x": ("built", "x"), "y": ("built", "y")} and m.setdefault("new", 5) == 5
    m = mk()
    assert list(m.keys()) == ["cp", "hr", "wttc"] and len(m) == 3 and "hr" in m and m.built == []   # keys cost nothing
    assert m.get("wttc") == ("built", "wttc") and m.built == ["wttc"] and m.get("nope", 3) == 3
    assert [v for _, v in mk().items()][0] == ("built", "cp") and list(mk().values())[2] == ("built", "wttc")
    assert mk() == plain and not (mk() != plain) and mk() == mk()
    import copy
    import pickle
    assert pickle.loads(pickle.dumps(mk())) == plain and copy.deepcopy(mk()) == plain
    assert all(v is not MM._UNBUILT for v in dict(mk()).values())
    assert "_UNBUILT" not in repr(mk()) and "object object" not in repr(mk())


def test_native_trajectory_packer_equals_the_numpy_gather():
    """csrc/fo_pyhost.c (CPython extension, built by __graft_entry__.build_pyhost): [5][M][T] blocks from the planner's
    trajectory objects, bit for bit what metrics.metric.trajectories_to_arrays gathers; lists / float32 / strided vectors are
    converted, a vector of another length raises like the Python path"""
    import __graft_entry__ as g
    from frenetix_occlusion import _native as N
    from frenetix_occlusion.metrics.metric import trajectories_to_arrays
    if g.build_pyhost() is None:
        pytest.skip("no C compiler / Python headers here")
    H = N.pyhost()
    assert H is not None
    rng = np.random.default_rng(5)
    M, T = 300, 31
    objs = [SimpleNamespace(cartesian=SimpleNamespace(**{k: rng.normal(size=T) for k in ("x", "y", "theta", "v", "a")})) for _ in range(M)]
    objs[3].cartesian.x = list(objs[3].cartesian.x)                        # a list
    objs[4].cartesian.y = objs[4].cartesian.y.astype(np.float32)           # another dtype
    objs[5].cartesian.v = np.repeat(objs[5].cartesian.v, 2)[::2]           # a strided view
    want = trajectories_to_arrays(objs)
    out = np.full((5, M, T), np.nan)
    H.pack_trajectories(objs, out, ("x", "y", "theta", "v", "a"))
    for i, k in enumerate(("x", "y", "theta", "v", "a")):
        assert np.array_equal(out[i], want[k]), k
    H.pack_trajectories(tuple(objs), out, ("x", "y", "theta", "v", "a"))   # tuples as well
    objs[7].cartesian.a = np.zeros(T + 1)
    with pytest.raises(ValueError):
        H.pack_trajectories(objs, out, ("x", "y", "theta", "v", "a"))
    with pytest.raises(ValueError):
        H.pack_trajectories(objs[:10], out, ("x", "y", "theta", "v", "a"))
    with pytest.raises(AttributeError):
        H.pack_trajectories([SimpleNamespace(cartesian=SimpleNamespace(x=np.zeros(T)))] * M, out, ("x", "y", "theta", "v", "a"))


@pytest.mark.parametrize("helper", ["c", "numpy"])
def test_obstacle_rows_of_a_step_equal_the_per_obstacle_statement(helper, monkeypatch):
    """FOObstacles.update (all obstacles of a step at once: the C helper csrc/fo_pyhost.c obstacle_rows, or its numpy
    statement) against FOObstacle.update_at_timestep per obstacle (fo_obstacle.py:79-116, helper_functions.py:99-112):
    poses, corner points, presence and the rows a one-call step uploads, bit for bit -- present, not yet there, gone"""
    import __graft_entry__ as g
    g.build_pyhost()
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.utils import fo_obstacle as FO
    if helper == "numpy":
        monkeypatch.setattr(FO, "_pyhost", lambda: None)
    elif FO._pyhost() is None:
        pytest.skip("_fo_pyhost not built")
    for n in (1, 2, 3):
        sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{n}_geometry.npz"))
        obst = list(sc.obstacles) * (5 if helper == "numpy" else 1)      # (the numpy form serves from four obstacles on)
        a, b = FO.FOObstacles(obst), FO.FOObstacles(obst)
        for t in (0, 1, 5, 17, 40, 200):
            a.update(t)
            for o in b.fo_obstacles:
                o.update_at_timestep(t)
            assert a._packed is not None and b._packed is None
            assert a.packed().tobytes() == b.packed().tobytes(), (n, t)
            for x, y in zip(a, b):
                assert (x.current_pos is None) == (y.current_pos is None) and x.relative_time_step == y.relative_time_step
                if x.current_pos is not None:
                    assert np.array_equal(x.current_pos, y.current_pos) and x.current_orientation == y.current_orientation
                    assert np.array_equal(x.current_corner_points, y.current_corner_points)
            for u, v in zip(a.arrays_full(), b.arrays_full()):
                assert np.array_equal(u, v)
        # a step's views outlive the step: the next update works on fresh memory
        a.update(0)
        first = [o.current_corner_points for o in a if o.current_pos is not None]
        keep = [c.copy() for c in first]
        a.update(7)
        assert all(np.array_equal(c, k) for c, k in zip(first, keep))


def test_pending_visibility_is_applied_before_the_obstacles_move_on():
    """FOObstacles / FOObstacle: a pending visibility resolver (SensorModel.defer_visible_objects after a one-call step) runs
    when a visibility attribute is read, and at the latest when update() moves the obstacles to the next step"""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    fo = FOObstacles(sc.obstacles)
    fo.update(0)
    calls = []

    def resolver():
        fo._pending = None
        calls.append(1)
        for o in fo:
            o.current_visible = True
            o.last_visible_at_ts = 0
        fo.update_multipolygon()

    fo._pending = resolver
    assert fo.fo_obstacles[0].current_visible is True and len(calls) == 1            # a read resolves
    assert len(fo.visible_obstacle_multipolygon) == sum(o.current_pos is not None for o in fo)
    fo._pending = resolver
    fo.update(1)                                                                   # ... and so does the next step
    assert len(calls) == 2 and all(o.last_visible_at_ts == 0 for o in fo) and not any(o._current_visible for o in fo)


def test_host_mirror_pool_never_rewrites_memory_somebody_still_reads():
    """metrics.metric.HostMirrorPool (the host memory of a batch's full per-pair mirror, kept across planning steps): a
    buffer is reused only when no view of it is alive outside the pool; a planner that keeps a piece of a step's results keeps
    that step's buffer and the pool takes a new one"""
    import gc
    import torch
    from frenetix_occlusion.metrics.metric import HostMirrorPool
    pool = HostMirrorPool(pinned=False)
    a = torch.arange(24, dtype=torch.float64).reshape(2, 3, 4)
    v1 = pool.fetch("lists", a)
    assert np.array_equal(v1, a.numpy()) and pool.allocations == 1
    kept = v1[:, :, 1]                              # a column of step 1 stays with the planner
    v2 = pool.fetch("lists", a + 100.0)             # step 2
    assert pool.allocations == 2 and np.array_equal(kept, a.numpy()[:, :, 1]) and np.array_equal(v2, (a + 100.0).numpy())
    del v1, kept, v2
    gc.collect()
    v3 = pool.fetch("lists", a * 2.0)               # nobody holds step 2's views: the same memory again
    assert pool.allocations == 2 and np.array_equal(v3, (a * 2.0).numpy())
    v4 = pool.fetch("lists", torch.zeros(1000, dtype=torch.float64))      # a larger batch: a larger buffer
    assert pool.allocations == 3 and v4.shape == (1000,)
    w = pool.fetch("pair_i", torch.arange(6, dtype=torch.int32).reshape(2, 3))
    assert w.dtype == np.int32 and w.tolist() == [[0, 1, 2], [3, 4, 5]]
