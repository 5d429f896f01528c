"""The reference's spawn -> agent flow as a device-resident stage of the planning step (fo_step_run with spawn_mode
FO_SPAWN_RULES / FO_SPAWN_BOTH: fo_scene_spawn_rules -> fo_scene_spawn_rule_agents -> sweep; include/fo_hip.h).

What the reference does per step (interface.py:186-198): ``find_spawn_points`` -> for every point ``add_agent`` (pedestrian
heading from the path / lane-centre normal, agent.py:451-505; vehicles through the route table, :283-426).  Here the rule
points never leave HBM.  Checked against the host statement of the same flow -- the rule points read back, then
``FOAgentManager.add_agent`` per point (the product's host path for scripted agents: numpy) -- on the three known-answer
scenes of tests/test_spawn_rules_gpu.py and the scenario-1 fixture at steps 0 / 8 / 25 / 60: same agents, same slots,
predictions to 1e-12 (headings and speeds bit for bit where no transcendental differs), and the sweep of the one-call
step equal to the sweep over the host-built agents."""
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = {"spawn_locator": {"spawn_points_behind_turn": True, "spawn_point_behind_static_obstacle": True,
                         "spawn_point_behind_dynamic_obstacle": True, "max_static_spawn_points": 1,
                         "max_dynamic_spawn_points": 1},
       "agent_manager": {"pedestrian": {"width": 0.5, "length": 0.3, "default_velocity": 1.4},
                         "bicycle": {"width": 0.9, "length": 2.0, "default_velocity": 5.0},
                         "car": {"width": 2.0, "length": 4.8, "default_velocity": 10.0},
                         "truck": {"width": 2.5, "length": 9.0, "default_velocity": 8.0},
                         "prediction": {"variance_factor": 1.05, "size_factor_length_s": 1.2, "size_factor_width_s": 1.3,
                                        "size_factor_length_l": 1.4, "size_factor_width_l": 2.5}},
       "accelerator": {"spawn": {"mode": "rules", "routes": 3, "max_rule_points": 8}}}
VEH = (4.508, 1.610, 1.4227, 1093.3, 11.5)


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    return torch


def _straight(S, lid, x0, x1, y_lo, y_hi, n=41):
    xs = np.linspace(x0, x1, n)
    return S.Lanelet(lid, np.stack((xs, np.full(n, y_hi)), -1), np.stack((xs, np.full(n, y_lo)), -1))


def _scenes():
    """(name, lanelets, obstacles, path, ego, yaw, v, intersections, timestep) -- the KAT scenes + scenario 1"""
    from frenetix_occlusion import scenario as S
    out = []
    lanes = [_straight(S, 1, -10, 70, -3.5, 0.0), _straight(S, 2, -10, 70, 0.0, 3.5)]
    path = np.stack((np.linspace(-5, 65, 141), np.full(141, -1.0)), -1)
    car = S.Obstacle(77, "static", "parkedVehicle", 4.5, 1.8, 0, np.array([17.0, -2.4, 0.0, 0.0]), np.zeros((0, 4)))
    out.append(("parked car", lanes, [car], path, np.array([0.0, -1.0]), 0.0, 8.0, None, 0))
    main = [_straight(S, 1, -40, 40, -3.5, 0.0), _straight(S, 2, -40, 40, 0.0, 3.5)]
    ys = np.linspace(-3.5, -43.5, 41)
    side = [S.Lanelet(3, np.stack((np.full(41, 13.5), ys), -1), np.stack((np.full(41, 10.0), ys), -1)),
            S.Lanelet(4, np.stack((np.full(41, 17.0), ys), -1), np.stack((np.full(41, 13.5), ys), -1))]
    ang = np.linspace(0, math.pi / 2, 30)
    tpath = np.concatenate((np.stack((np.linspace(-30, 7.75, 76), np.full(76, -1.75)), -1),
                            np.stack((7.75 + 4.0 * np.sin(ang), -5.75 + 4.0 * np.cos(ang)), -1)[1:],
                            np.stack((np.full(60, 11.75), np.linspace(-6.25, -36.0, 60)), -1)))
    out.append(("right turn", main + side, [], tpath, np.array([-5.0, -1.75]), 0.0, 6.0, None, 0))
    mirror = lambda a: a * np.array([1.0, -1.0])
    lanes_l = [S.Lanelet(ll.lanelet_id, mirror(ll.right), mirror(ll.left)) for ll in main + side]
    out.append(("left turn", lanes_l, [], mirror(tpath), mirror(np.array([-5.0, -1.75])), 0.0, 6.0, None, 0))
    xs = np.linspace(-10, 70, 41)
    lane1 = S.Lanelet(1, np.stack((xs, np.zeros(41)), -1), np.stack((xs, np.full(41, -3.5)), -1))
    lane2 = S.Lanelet(2, np.stack((xs[::-1], np.zeros(41)), -1), np.stack((xs[::-1], np.full(41, 3.5)), -1))
    lane1.adj_left, lane1.adj_left_same_direction = 2, False
    lane2.adj_left, lane2.adj_left_same_direction = 1, False
    opath = np.stack((np.linspace(-5, 65, 141), np.full(141, -1.75)), -1)
    truck = S.Obstacle(31, "dynamic", "truck", 9.0, 3.2, 0, np.array([20.0, 1.75, math.pi, 8.0]), np.zeros((0, 4)))
    out.append(("oncoming truck", [lane1, lane2], [truck], opath, np.array([0.0, -1.75]), 0.0, 8.0, None, 0))
    # the same street with the oncoming lane in pieces and a side street leaving it: the phantom vehicles behind the truck
    # stand on a lanelet with two candidate routes (straight on, turn off)
    rev = lambda a, b, n: np.linspace(a, b, n)
    seg = lambda lid, a, b, n: S.Lanelet(lid, np.stack((rev(a, b, n), np.zeros(n)), -1), np.stack((rev(a, b, n), np.full(n, 3.5)), -1))
    l2a, l2b, l2c = seg(21, 70, 45, 14), seg(22, 45, 10, 19), seg(23, 10, -10, 11)
    ysd = np.linspace(3.5, 23.5, 11)
    l2d = S.Lanelet(24, np.stack((np.full(11, 6.5), ysd), -1), np.stack((np.full(11, 10.0), ysd), -1))
    l2a.successors, l2b.predecessors, l2b.successors = [22], [21], [23, 24]
    l2c.predecessors, l2d.predecessors = [22], [22]
    lane1b = S.Lanelet(1, lane1.left.copy(), lane1.right.copy())
    lane1b.adj_left, lane1b.adj_left_same_direction = 22, False
    for l in (l2a, l2b, l2c):
        l.adj_left, l.adj_left_same_direction = 1, False
    out.append(("oncoming truck, branching lane", [lane1b, l2a, l2b, l2c, l2d], [truck], opath, np.array([0.0, -1.75]), 0.0, 8.0, None, 0))
    sc3 = S.load_geometry_npz(os.path.join(GOLDEN, "scenario3_geometry.npz"))      # right turn at an intersection, parked car
    by = {l.lanelet_id: l for l in sc3.lanelets}
    parts = [by[1].center]
    for lid in (12, 9):                                   # incoming lanelet -> right-turn lanelet -> side street
        c = by[lid].center
        parts.append(c[1:] if np.linalg.norm(c[0] - parts[-1][-1]) < 1e-2 else c)
    for x in (12.0, 18.0):
        out.append((f"scenario3 x={x}", sc3.lanelets, sc3.obstacles, np.concatenate(parts), np.array([x, 0.0]), 0.0, 8.0,
                    sc3.intersections, 0))
    car2 = S.Obstacle(78, "static", "parkedVehicle", 4.5, 1.8, 0, np.array([23.0, -2.5, 0.05, 0.0]), np.zeros((0, 4)))
    out.append(("parked car, slightly turned", lanes, [car2], path, np.array([2.0, -1.0]), 0.0, 6.0, None, 0))
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    yaw = float(ego0[2])
    spath = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    for step in (0, 8, 25, 60):
        ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
        out.append((f"scenario1 step {step}", sc.lanelets, sc.obstacles, spath, ego, yaw, float(ego0[3]), sc.intersections, step))
    return out


def _stack(torch, lanelets, obstacles, path, intersections, timestep, M=192, T=31, mode="rules", seed=3, ego=None, yaw=0.0,
           out_mode="pair"):
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as SY
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.step import PlanningStep
    from frenetix_occlusion.sweep import MetricSweep
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    import copy
    cfg = copy.deepcopy(CFG)
    cfg["accelerator"]["spawn"]["mode"] = mode
    cfg["accelerator"]["spawn"]["max_agents"] = 6
    ctx = N.Context(0)
    obs = FOObstacles(obstacles)
    obs.update(timestep)
    sm = SensorModel(lanelets, path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, ctx=ctx, routes=3,
                     intersections=intersections)
    sl = SpawnLocator(None, path, cfg, sm, fo_obstacles=obs, dt=0.1, horizon=(T - 1) * 0.1)
    sw = MetricSweep(VEH, 0.1, thresholds={"harm": 0.1, "risk": 1}, ctx=ctx)
    traj = SY.make_trajectories(M, T, 0.1, seed=seed, ego_pos=ego, ego_yaw=yaw)
    tr = [torch.as_tensor(traj[k]).cuda() for k in ("x", "y", "theta", "v", "a")]
    return SimpleNamespace(ctx=ctx, obs=obs, sm=sm, sl=sl, sw=sw, tr=tr, traj=traj, cfg=cfg,
                           step=lambda: PlanningStep(sm, sl, sw, *tr, mode=out_mode))


def _host_agents(lanelets, path, points, timestep, T):
    """today's host path: FOAgentManager.add_agent per rule point, the way interface.py:192-198 calls it"""
    from frenetix_occlusion.agent import FOAgentManager
    am = FOAgentManager(SimpleNamespace(lanelet_network=lanelets, obstacles=[]), path, CFG["agent_manager"], timestep, dt=0.1,
                        device="cpu")
    for sp in points:
        mode = "lane_center" if sp.source in ("left turn", "right turn") else "ref_path"
        am.add_agent(pos=sp.position, velocity="default", agent_type=sp.agent_type, timestep=timestep, horizon=(T - 1) * 0.1,
                     mode=mode, orientation=sp.orientation)
    return am._manual


def _device_agents(b):
    """(pos0, yaw0, per-slot dict) of the live rule agents of a batch, read from the device"""
    h = b.host_head()
    g = lambda t: t.cpu().numpy()
    pos, yaw, v, cov, shape, raw, typ, ln = (g(t) for t in b.sweep_args())
    return h, dict(pos=pos, yaw=yaw, v=v, cov=cov, shape=shape, raw=raw, type=typ, len=ln)


def test_rule_agents_on_the_device_equal_the_host_add_agent_flow(torch_cuda, oracle):
    """... and, for the rule families' Cars and Bicycles, the C oracle's route predictions (fo_oracle_route_predictions: the
    checker of the cell sampler's vehicles, pinned by tests/golden/routes.npz and sampling_matrix.npz) from the same spawn
    pose, the lanelet under it and the map's route table -- an anchor outside the product for the vehicle half"""
    torch = torch_cuda
    from frenetix_occlusion import scenario as SCN
    from frenetix_occlusion.spawn_locator import TYPE_CODE
    T, n_ped, n_veh, n_multi, n_anchor = 31, 0, 0, 0, 0
    for name, lanelets, obstacles, path, ego, yaw, v, inter, step in _scenes():
        k = _stack(torch, lanelets, obstacles, path, inter, step, ego=ego, yaw=yaw)
        k.sm.upload_obstacles(k.obs)
        ps = k.step()
        ps.run(ego, yaw, v)
        torch.cuda.synchronize()
        b = k.sl.batch
        h, dev = _device_agents(b)
        points = k.sl._points_from_head(b)[1]
        assert len(points) == h["rule_n"], name
        host = _host_agents(lanelets, path, points, step, T)
        assert len(host) == len(points), name
        R, a0 = b.R, b.n_cell_agents
        assert a0 == 0 and R == 3
        for i, (sp, ag) in enumerate(zip(points, host)):
            j = a0 + i
            assert np.array_equal(h["pos0"][j], sp.position), name
            assert h["yaw0"][j] == pytest.approx(ag.initial_orientation, abs=1e-12), (name, i)
            assert len(ag.predictions) <= R
            n_ped += ag.agent_type == "Pedestrian"
            n_veh += ag.agent_type != "Pedestrian"
            n_multi += len(ag.predictions) > 1
            for r in range(R):
                s = j * R + r
                if r >= len(ag.predictions):
                    assert dev["len"][s] == 0, (name, i, r)
                    continue
                p = ag.predictions[r]
                L = len(p["pos_list"])
                assert dev["len"][s] == L and dev["type"][s] == TYPE_CODE[ag.agent_type.lower()], (name, i, r)
                np.testing.assert_allclose(dev["pos"][s, :L], p["pos_list"], rtol=0, atol=1e-12, err_msg=f"{name} {i} {r}")
                np.testing.assert_allclose(dev["yaw"][s, :L], p["orientation_list"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(dev["v"][s, :L], p["v_list"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(dev["cov"][s, :L], p["cov_list"], rtol=1e-14, atol=0)
                assert np.all(dev["pos"][s, L:] == 0.0)
                np.testing.assert_allclose(dev["shape"][s], (p["shape"]["length"], p["shape"]["width"]), rtol=1e-15)
                assert tuple(dev["raw"][s]) == (ag.length, ag.width)
        for s in range((a0 + h["rule_n"]) * R, len(dev["len"])):
            assert dev["len"][s] == 0                                     # slots of points that do not exist
        # oracle-side anchor of the vehicles: pose + lanelet + route table -> predictions by oracle/fo_oracle.c
        vi = [i for i, sp in enumerate(points) if sp.agent_type != "Pedestrian"]
        if vi and k.sm.route_table is not None:
            tab = k.sm.route_table
            lan = []
            for i in vi:
                inside = [q for q, ll in enumerate(lanelets) if SCN.points_in_polygon(points[i].position.reshape(1, 2), ll.polygon)[0]]
                lan.append(inside[0] if inside else -1)
            types = np.array([TYPE_CODE[points[i].agent_type.lower()] for i in vi], dtype=np.int32)
            speed = np.array([CFG["agent_manager"][points[i].agent_type.lower()]["default_velocity"] for i in vi], dtype=np.float64)
            pos0 = np.stack([points[i].position for i in vi])
            po, yo, vo, co, lo = oracle.route_predictions(pos0, types, speed, np.array(lan, dtype=np.int32), R, tab.first, tab.count,
                                                          tab.xy, tab.s, h["yaw0"][[a0 + i for i in vi]], T, 0.1, 0.1, 1.05)
            for q, i in enumerate(vi):
                for r in range(R):
                    s_, so = (a0 + i) * R + r, q * R + r
                    assert dev["len"][s_] == lo[so], (name, i, r)
                    L = int(lo[so])
                    np.testing.assert_allclose(dev["pos"][s_, :L], po[so, :L], rtol=0, atol=1e-9, err_msg=f"{name} {i} {r}")
                    np.testing.assert_allclose(dev["yaw"][s_, :L], yo[so, :L], rtol=0, atol=1e-12)
                    np.testing.assert_allclose(dev["v"][s_, :L], vo[so, :L], rtol=0, atol=1e-12)
                    np.testing.assert_allclose(dev["cov"][s_, :L], co[so, :L], rtol=1e-13, atol=0)
                    n_anchor += L > 0
    assert n_ped >= 3 and n_veh >= 3 and n_multi >= 1, (n_ped, n_veh, n_multi)   # pedestrians, vehicles on one and on several routes
    assert n_anchor >= 3, n_anchor


def test_one_call_rules_step_equals_the_sweep_over_the_host_built_agents(torch_cuda):
    """cost vectors, flags and pair scalars of fo_step_run (spawn_mode rules) against fo_sweep_run over the agent arrays the
    host flow builds (FOAgentManager.sweep_arrays): integers exact, floats to 1e-9"""
    torch = torch_cuda
    from frenetix_occlusion.agent import FOAgentManager
    from frenetix_occlusion.sweep import MetricSweep
    T, seen = 31, 0
    for name, lanelets, obstacles, path, ego, yaw, v, inter, step in _scenes():
        k = _stack(torch, lanelets, obstacles, path, inter, step, ego=ego, yaw=yaw)
        k.sm.upload_obstacles(k.obs)
        out = k.step().run(ego, yaw, v)
        torch.cuda.synchronize()
        b = k.sl.batch
        points = k.sl._points_from_head(b)[1]
        if not points:
            assert bool(out.safe.bool().all()), name                    # nothing spawned: every candidate is safe
            continue
        seen += 1
        am = FOAgentManager(SimpleNamespace(lanelet_network=lanelets, obstacles=[]), path, CFG["agent_manager"], step, dt=0.1,
                            device=torch.device("cuda", 0))
        for sp in points:
            mode = "lane_center" if sp.source in ("left turn", "right turn") else "ref_path"
            am.add_agent(pos=sp.position, velocity="default", agent_type=sp.agent_type, timestep=step, horizon=3.0, mode=mode,
                         orientation=sp.orientation)
        sw2 = MetricSweep(VEH, 0.1, thresholds={"harm": 0.1, "risk": 1})
        sw2.set_agents(*am.sweep_arrays())
        ref = sw2.run(*k.tr, mode="pair")
        torch.cuda.synchronize()
        # host slots are dense (one per prediction); device slots are (point, route) with gaps
        R = b.R
        ln = b.len.cpu().numpy()
        live = [s for s in range(len(ln)) if ln[s] > 0]
        assert len(live) == ref.pair_f.shape[1], name
        pf, pi = out.pair_f.cpu().numpy()[:, live], out.pair_i.cpu().numpy()[:, live]
        rf, ri = ref.pair_f.cpu().numpy(), ref.pair_i.cpu().numpy()
        assert np.array_equal(pi[:1], ri[:1]) and np.array_equal(pi[3], ri[3]), name   # time_dce, hr_valid
        np.testing.assert_allclose(pf[:9], rf[:9], rtol=0, atol=1e-9, err_msg=name)
        c, rc = out.cost.cpu().numpy(), ref.cost.cpu().numpy()
        np.testing.assert_allclose(c[:, :9], rc[:, :9], rtol=0, atol=1e-9, err_msg=name)
        assert np.array_equal(out.safe.cpu().numpy(), ref.safe.cpu().numpy()), name
    assert seen >= 4


@pytest.mark.parametrize("mode", ["rules", "both"])
def test_one_call_step_with_rules_equals_the_stage_calls(torch_cuda, mode):
    """fo_step_run with the rule stage against the same stages queued one by one (fo_scene_fan / fo_scene_visibility /
    fo_scene_spawn / fo_scene_spawn_rules / fo_scene_spawn_rule_agents / fo_sweep_set_agents / fo_sweep_run): every output
    bit for bit, over several steps of scenario 1"""
    torch = torch_cuda
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    yaw = float(ego0[2])
    path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    got = {}
    for how in ("stages", "one-call"):
        k = _stack(torch, sc.lanelets, sc.obstacles, path, sc.intersections, 0, mode=mode, ego=ego0[:2], yaw=yaw, M=500)
        ps = k.step() if how == "one-call" else None
        res = []
        for step in (0, 8, 25, 60):
            ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
            k.obs.update(step)
            k.sm.upload_obstacles(k.obs)
            if ps is not None:
                out = ps.run(ego, yaw, float(ego0[3]))
            else:
                k.sm.launch(ego, yaw)
                k.sl.find_spawn_points(ego, yaw, None, float(ego0[3]), lazy=True)
                k.sw.set_agents(*k.sl.batch.sweep_args(), check=False)
                out = k.sw.run(*k.tr, mode="pair")
            torch.cuda.synchronize()
            b = k.sl.batch
            res.append([t.cpu().numpy().copy() for t in (out.cost, out.safe, out.pair_f, out.pair_i, k.sm.cell_class, b.pos, b.yaw,
                                                          b.v, b.len, b.type, b.head)])
        got[how] = res
    n_rule = 0
    for a, b in zip(got["stages"], got["one-call"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
        n_rule += int((a[8] > 0).sum())
    assert n_rule > 0


def test_interface_in_rules_mode_keeps_the_spawn_points_on_the_device(torch_cuda, tmp_path):
    """FOInterface.evaluate_scenario with spawn.mode rules: no add_agent on the host, spawn_points is a lazy view, the agent
    registry and the per-trajectory result dicts are the reference's views of the device batch"""
    import yaml
    torch = torch_cuda
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion import synthetic as SY
    from frenetix_occlusion.spawn_locator import LazySpawnPoints
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"].update(mode="rules")
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    ego0 = sc.ego_initial
    yaw = float(ego0[2])
    path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    veh = SimpleNamespace(length=VEH[0], width=VEH[1], wb_rear_axle=VEH[2], mass=VEH[3], a_max=VEH[4])
    fo = interface.FOInterface(sc, path, veh, 0.1, config_path=str(cfg_path))
    calls = []
    orig = fo.agent_manager.add_agent
    fo.agent_manager.add_agent = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    total = 0
    for step in (0, 8, 25):
        ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
        fo.evaluate_scenario({}, ego, yaw, None, float(ego0[3]), step)
        assert isinstance(fo.spawn_points, LazySpawnPoints) and not fo.spawn_points.materialised   # nothing read back yet
        traj = SY.make_trajectories(64, 31, 0.1, seed=1, ego_pos=ego, ego_yaw=yaw)
        ba = fo.trajectory_safety_assessment_batch(traj, mode="reduced")
        assert ba is not None and ba.cost.shape == (64, 16)
        pts = list(fo.spawn_points)
        agents = fo.agent_manager.phantom_agents
        assert len(agents) == len(pts) and [a.agent_type for a in agents] == [p.agent_type for p in pts]
        for a, p in zip(agents, pts):
            assert np.array_equal(a.initial_position, p.position)
        preds = fo.agent_manager.predictions
        assert len(preds) >= len(agents) and all(int(str(pid)[:5]) in {a.agent_id for a in agents} for pid in preds)
        if not pts:
            assert bool(ba.safe.bool().all()) and fo.trajectory_safety_assessment(SimpleNamespace(cartesian=SimpleNamespace(
                **{q: traj[q][0] for q in ("x", "y", "theta", "v", "a")})))[0] == {}
        total += len(pts)
    assert total > 0 and not calls
