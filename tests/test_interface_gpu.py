"""End-to-end check of the drop-in boundary on a real MI355X: FOInterface.evaluate_scenario +
trajectory_safety_assessment(_batch) against the oracle fed with the same phantom predictions."""
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    return torch


def _traj_objects(traj):
    return [SimpleNamespace(cartesian=SimpleNamespace(**{k: v[i] for k, v in traj.items()})) for i in range(len(traj["x"]))]


def _setup(tmp_path, thresholds=None, max_agents=16, metrics=None, include_real_agents=False):
    import yaml
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion import synthetic as SY
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    cfg["accelerator"]["spawn"]["max_agents"] = max_agents
    cfg["accelerator"]["include_real_agents"] = include_real_agents
    if thresholds:
        cfg["metrics"]["metric_thresholds"].update(thresholds)
    if metrics:
        cfg["metrics"]["activated_metrics"] = list(metrics)
    p = tmp_path / "occlusion.yaml"
    p.write_text(yaml.safe_dump(cfg))
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    s = np.linspace(0.0, 80.0, 81)
    ref_path = ego[None, :2] + s[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
    veh = SimpleNamespace(length=SY.VEHICLE_BMW320I[0], width=SY.VEHICLE_BMW320I[1], wb_rear_axle=SY.VEHICLE_BMW320I[2],
                          mass=SY.VEHICLE_BMW320I[3], a_max=SY.VEHICLE_BMW320I[4])
    fo = interface.FOInterface(sc, ref_path, veh, 0.1, config_path=str(p))
    return fo, sc, ego, SY


def _oracle_agents(fo):
    b = fo.spawn_locator.batch
    return {"pos": b.pos.cpu().numpy(), "yaw": b.yaw.cpu().numpy(), "v": b.v.cpu().numpy(), "cov": b.cov.cpu().numpy(),
            "shape": b.shape.cpu().numpy(), "raw_dims": b.raw_dims.cpu().numpy(), "type": b.type.cpu().numpy(),
            "len": b.len.cpu().numpy()}


def test_evaluate_scenario_then_batch_and_per_trajectory_calls(torch_cuda, oracle, tmp_path):
    thr = {"harm": 0.1, "risk": 0.05}
    fo, sc, ego, SY = _setup(tmp_path, thr)
    vis = fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    assert vis is fo.sensor_model.visible_area and not vis.is_empty
    assert vis.exterior.shape == (720, 2) and vis.area > 50.0
    assert len(fo.spawn_points) > 0 and len(fo.agent_manager.phantom_agents) == len(fo.spawn_points)
    preds = fo.agent_manager.predictions
    assert len(preds) >= len(fo.spawn_points)            # a phantom vehicle has one prediction per candidate route
    for pid, p in preds.items():
        assert set(p) == {"orientation_list", "v_list", "pos_list", "shape", "cov_list"}
        L = len(p["pos_list"])
        assert 1 <= L <= 31 and p["pos_list"].shape == (L, 2) and p["cov_list"].shape == (L, 2, 2)
        assert fo.agent_manager.agent_by_prediction_id(pid) is not None
    peds = [a for a in fo.agent_manager.phantom_agents if a.agent_type == "Pedestrian"]
    assert all(len(a.predictions) == 1 and len(a.predictions[0]["pos_list"]) == 31 for a in peds)

    traj = SY.make_trajectories(64, seed=99, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    objs = _traj_objects(traj)
    ref = oracle.sweep(traj, _oracle_agents(fo), SY.VEHICLE_BMW320I, 0.1, thr=thr)

    ba = fo.trajectory_safety_assessment_batch(objs, mode="reduced")
    torch_cuda.cuda.synchronize()
    cost = ba.cost.cpu().numpy()
    fin = np.isfinite(ref["cost"])
    assert np.array_equal(np.isfinite(cost), fin)
    np.testing.assert_allclose(cost[fin], ref["cost"][fin], rtol=0, atol=1e-9)
    assert np.array_equal(ba.safe.cpu().numpy(), ref["safe"])

    # the reference's per-trajectory contract, served from a cached full batch
    fo.trajectory_safety_assessment_batch(objs, mode="full")
    slots = dict(fo.agent_manager.prediction_slots)
    for m in (0, 17, 63):
        res, safe = fo.trajectory_safety_assessment(objs[m])
        assert safe == bool(ref["safe"][m])
        assert list(res.keys()) == ["cp", "dce", "ttc", "hr", "ttce", "wttc"]      # metric.py:125-147 order
        for pid, k in slots.items():
            np.testing.assert_allclose(res["cp"][pid], ref["lists"][m, k, 0][:len(res["cp"][pid])], atol=1e-9)
            assert len(res["cp"][pid]) == 30                            # collision_probability.py: always T-1 entries
            assert res["dce"][pid]["time_dce"] == ref["pair_i"][m, k, oracle.PI["time_dce"]]
            assert res["dce"][pid]["dce"] == pytest.approx(ref["pair_f"][m, k, oracle.PF["dce"]], abs=1e-9)
            assert res["ttce"][pid] == pytest.approx(ref["pair_f"][m, k, oracle.PF["ttce"]], abs=1e-12)
            h = res["hr"][pid]
            assert h["max_obst_risk"] == pytest.approx(ref["pair_f"][m, k, oracle.PF["max_obst_risk"]], abs=1e-9)
            Lh = min(30, len(preds[pid]["pos_list"]))                   # harm_model.py:66
            assert len(h["ego_harm_traj"]) == Lh and len(h["obst_risk_traj"]) == Lh
        assert res["hr"]["max_obst_risk_all"] == pytest.approx(ref["cost"][m, oracle.COST["max_obst_risk_all"]], abs=1e-9)
        w = ref["cost"][m, oracle.COST["wttc"]]
        assert res["wttc"] == w or (math.isinf(w) and math.isinf(res["wttc"]))
    # a trajectory that was not part of the batch: one-trajectory launch
    other = _traj_objects(SY.make_trajectories(3, seed=5, ego_pos=ego[:2], ego_yaw=float(ego[2])))[1]
    res, safe = fo.trajectory_safety_assessment(other)
    assert "hr" in res and isinstance(safe, bool)


def test_no_phantoms_means_empty_result_and_safe(torch_cuda, tmp_path):
    fo, sc, ego, SY = _setup(tmp_path)
    fo.spawn_locator.min_ahead = 1e6          # no candidate survives the gate
    fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    assert fo.spawn_points == [] and not fo.agent_manager.phantom_agents
    obj = _traj_objects(SY.make_trajectories(2, seed=1))[0]
    assert fo.trajectory_safety_assessment(obj) == ({}, True)           # metric.py:44-45
    assert fo.trajectory_safety_assessment_batch([obj]) is None


def test_manual_agents_and_error_conventions(torch_cuda, oracle, tmp_path):
    fo, sc, ego, SY = _setup(tmp_path, max_agents=4)
    fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    am = fo.agent_manager
    n0 = len(am.phantom_agents)
    a = am.add_agent(pos=np.array([15.0, 3.0]), velocity="default", agent_type="Pedestrian", timestep=0)
    assert a is not None and len(am.phantom_agents) == n0 + 1
    assert am.add_agent(pos=[0, 0], agent_type="Car", timestep=5) is None            # agent.py:69-70
    with pytest.raises(NotImplementedError):
        am.add_agent(pos=[0, 0], agent_type="Tank", timestep=0)                       # agent.py:120
    with pytest.raises(ValueError):
        am.add_agent(pos=[0, 0], agent_type="Car", velocity="fast", timestep=0)       # agent.py:100
    fo.metrics.invalidate()
    traj = SY.make_trajectories(16, seed=4, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    ba = fo.trajectory_safety_assessment_batch(traj, mode="pair")
    torch_cuda.cuda.synchronize()
    arrs = am.sweep_arrays()
    agents = dict(zip(("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len"), [t.cpu().numpy() for t in arrs]))
    assert agents["pos"].shape[0] == 4 * fo.spawn_locator.R + 1          # 4 agents x R route slots + 1 manual
    ref = oracle.sweep(traj, agents, SY.VEHICLE_BMW320I, 0.1, thr={"harm": 1, "risk": 1})
    got = ba.result.pair_f.permute(2, 1, 0).cpu().numpy()
    f = np.isfinite(ref["pair_f"])
    assert np.array_equal(np.isnan(got), np.isnan(ref["pair_f"]))
    np.testing.assert_allclose(got[f], ref["pair_f"][f], rtol=0, atol=1e-9)
    with pytest.raises(ValueError):
        from frenetix_occlusion.metrics.metric import Metric
        Metric({"activated_metrics": ["nope"], "metric_thresholds": {}}, fo.vehicle_params, am, dt=0.1)


def test_rule_based_spawn_points_through_the_interface(torch_cuda, oracle, tmp_path):
    """accelerator.spawn.mode = rules / both on scenario 3 (right turn at an intersection, parked car in the side
    street): the reference's 'behind turn' rule evaluated on the GPU's cell classes puts a pedestrian behind the
    corner; rule points become agents on the device (fo_scene_spawn_rule_agents) and are assessed like every other
    phantom."""
    import yaml
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion import synthetic as SY
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario3_geometry.npz"))
    by = {l.lanelet_id: l for l in sc.lanelets}
    parts = [by[1].center]
    for lid in (12, 9):                                   # incoming lanelet -> right-turn lanelet -> side street
        c = by[lid].center
        parts.append(c[1:] if np.linalg.norm(c[0] - parts[-1][-1]) < 1e-2 else c)
    ref_path = np.concatenate(parts)
    ego = np.array([12.0, 0.0, 0.0, 8.0])
    veh = SimpleNamespace(length=SY.VEHICLE_BMW320I[0], width=SY.VEHICLE_BMW320I[1], wb_rear_axle=SY.VEHICLE_BMW320I[2],
                          mass=SY.VEHICLE_BMW320I[3], a_max=SY.VEHICLE_BMW320I[4])
    found = {}
    for mode in ("rules", "both"):
        cfg["accelerator"]["spawn"]["mode"] = mode
        cfg["accelerator"]["spawn"]["max_agents"] = 6
        p = tmp_path / f"occ_{mode}.yaml"
        p.write_text(yaml.safe_dump(cfg))
        fo = interface.FOInterface(sc, ref_path, veh, 0.1, config_path=str(p))
        fo.evaluate_scenario({}, ego[:2], float(ego[2]), None, float(ego[3]), 0, None)
        rule_pts = fo.spawn_locator.rule_points
        found[mode] = (len(fo.spawn_points), len(rule_pts))
        assert len(rule_pts) == 1 and rule_pts[0].source == "right turn" and rule_pts[0].agent_type == "Pedestrian"
        pos = rule_pts[0].position
        assert 31.0 < pos[0] < 33.5 and -4.5 < pos[1] < -1.0          # behind the corner, right of the reference path
        assert not fo.sensor_model.visible_area.contains(pos[None])[0]
        assert len(fo.agent_manager.phantom_agents) == len(fo.spawn_points)
        ped = fo.agent_manager.phantom_agents[-1]
        assert ped.agent_type == "Pedestrian" and 0.0 <= ped.initial_orientation < 2 * math.pi
        traj = SY.make_trajectories(40, seed=3, ego_pos=ego[:2], ego_yaw=float(ego[2]))
        ba = fo.trajectory_safety_assessment_batch(traj, mode="pair")
        torch_cuda.cuda.synchronize()
        arrs = fo.agent_manager.sweep_arrays()
        agents = dict(zip(("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len"), [t.cpu().numpy() for t in arrs]))
        ref = oracle.sweep(traj, agents, SY.VEHICLE_BMW320I, 0.1, thr={"harm": 1, "risk": 1}, want_lists=False)
        got = ba.cost.cpu().numpy()
        f = np.isfinite(ref["cost"])
        assert np.array_equal(np.isfinite(got), f)
        # the agent index of the largest risk only means something where that risk is numerically there (risks of 1e-19
        # are the erf table's noise: either side may name any agent)
        ia, ir = oracle.COST["argmax_risk"], oracle.COST["max_obst_risk_all"]
        sig = ref["cost"][:, ir] > 1e-9
        assert np.array_equal(got[sig, ia], ref["cost"][sig, ia])
        got[:, ia] = ref["cost"][:, ia]
        np.testing.assert_allclose(got[f], ref["cost"][f], rtol=0, atol=1e-9)
    assert found["rules"] == (1, 1) and found["both"][0] > 1 and found["both"][1] == 1


def test_result_dict_column_gather_equals_host_mirror(torch_cuda, tmp_path):
    """large batches do not mirror their per-pair outputs on the host: the trajectory's column is gathered on the
    device instead -- same nested dict either way"""
    fo, sc, ego, SY = _setup(tmp_path, max_agents=8)
    fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    traj = SY.make_trajectories(12, seed=9, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    ba = fo.trajectory_safety_assessment_batch(traj, mode="full")
    host = [ba.result_dict(m) for m in (0, 5, 11)]
    ba2 = fo.trajectory_safety_assessment_batch(traj, mode="full")
    ba2.HOST_CACHE_BYTES = 0
    col = [ba2.result_dict(m) for m in (0, 5, 11)]
    assert ba2._to_host()["lists"] is None

    def same(a, b):
        if isinstance(a, dict):
            return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, np.ndarray):
            return np.array_equal(a, b, equal_nan=True)
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b or (isinstance(a, float) and math.isnan(a) and math.isnan(b))
    assert all(same(h, c) for h, c in zip(host, col))


def test_spawn_point_types_match_the_phantom_agents_with_route_slots(torch_cuda, tmp_path):
    """with R > 1 prediction slots per agent the spawn-point list still names agent j's own type (slot j * R)"""
    fo, sc, ego, SY = _setup(tmp_path, max_agents=12)
    assert fo.spawn_locator.R > 1
    fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    assert len(fo.spawn_points) == len(fo.agent_manager.phantom_agents) > 4
    pattern = [p.lower() for p in fo.config["accelerator"]["spawn"]["pattern"]]
    for j, (sp, ag) in enumerate(zip(fo.spawn_points, fo.agent_manager.phantom_agents)):
        assert sp.agent_type.lower() == ag.agent_type.lower() == pattern[j % 4]
        np.testing.assert_allclose(sp.position, ag.initial_position if hasattr(ag, "initial_position") else sp.position)


def test_many_planning_steps_do_not_exhaust_the_agent_ids(torch_cuda, tmp_path):
    """the reference API mints one id per phantom per step (agent.py:189-199, 1001 values); 250 steps at 32 phantoms
    through phantom_agents / predictions / agent_by_prediction_id must neither hang nor collide"""
    fo, sc, ego, SY = _setup(tmp_path, max_agents=32)
    scenario_ids = {o.obstacle_id for o in sc.obstacles}
    traj = SY.make_trajectories(8, seed=5, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    objs = _traj_objects(traj)
    n_seen = 0
    for step in range(250):
        fo.evaluate_scenario({}, ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
        agents = fo.agent_manager.phantom_agents
        ids = [a.agent_id for a in agents]
        assert len(set(ids)) == len(ids) and not (set(ids) & scenario_ids)
        assert all(10000 <= i <= 99999 for i in ids)
        n_seen += len(ids)
        if step % 50 == 0:
            preds = fo.agent_manager.predictions
            assert all(fo.agent_manager.agent_by_prediction_id(pid) is not None for pid in preds)
            res, safe = fo.trajectory_safety_assessment(objs[0])
            assert isinstance(safe, bool)
        assert len(fo.agent_manager.all_obstacle_id) <= len(scenario_ids) + len(ids) + len(fo.agent_manager.real_agents)
    assert n_seen >= 250 * 8


def _real_predictions(sc, ego, L=31, dt=0.1):
    """what a prediction module hands the planner for two real obstacles: constant velocity towards the ego's lane,
    covariances that grow and carry correlation"""
    preds = {}
    for j, ob in enumerate(sc.obstacles[:2]):
        yaw = float(ego[2]) + (2.2 if j == 0 else -1.9)
        p0 = ego[:2] + np.array([14.0 + 6.0 * j, 2.5 if j == 0 else -2.0])
        t = np.arange(L) * dt
        v = 1.5 + 2.0 * j
        pos = p0[None, :] + (v * t)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
        sx, sy = np.sqrt(0.05 * 1.06 ** np.arange(L)), np.sqrt(0.09 * 1.04 ** np.arange(L))
        rho = 0.7 if j == 0 else -0.93
        cov = np.zeros((L, 2, 2))
        cov[:, 0, 0], cov[:, 1, 1] = sx * sx, sy * sy
        cov[:, 0, 1] = cov[:, 1, 0] = rho * sx * sy
        preds[ob.obstacle_id] = {"pos_list": pos, "v_list": np.full(L, v), "orientation_list": np.full(L, yaw),
                                 "cov_list": cov, "shape": {"length": ob.length, "width": ob.width}}
    return preds


def test_real_agent_predictions_join_the_sweep_when_enabled(torch_cuda, oracle, tmp_path):
    """EXTENSION (accelerator.include_real_agents): the predictions handed to evaluate_scenario are evaluated next to
    the phantoms, keyed by obstacle id; off by default, like the reference (interface.py:216-219: phantoms only)."""
    fo, sc, ego, SY = _setup(tmp_path, max_agents=4, include_real_agents=True)
    preds = _real_predictions(sc, ego)
    fo.evaluate_scenario(dict(preds), ego[:2], float(ego[2]), (0.0, 0.0), float(ego[3]), 0, None)
    am = fo.agent_manager
    n_ph = 4 * fo.spawn_locator.R
    assert am.n_slots() == n_ph + 2
    traj = SY.make_trajectories(24, seed=9, ego_pos=ego[:2], ego_yaw=float(ego[2]))
    objs = _traj_objects(traj)
    ba = fo.trajectory_safety_assessment_batch(objs, mode="full")
    torch_cuda.cuda.synchronize()
    agents = dict(zip(("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len"),
                      [t.cpu().numpy() for t in am.sweep_arrays()]))
    assert np.all(agents["cov"][n_ph:, :, 0, 1] != 0.0)                   # the real agents' slots carry correlation
    ref = oracle.sweep(traj, agents, SY.VEHICLE_BMW320I, 0.1, thr={"harm": 1, "risk": 1})
    got = ba.result.pair_f.permute(2, 1, 0).cpu().numpy()
    f = np.isfinite(ref["pair_f"])
    assert np.array_equal(np.isnan(got), np.isnan(ref["pair_f"]))
    np.testing.assert_allclose(got[f], ref["pair_f"][f], rtol=0, atol=1e-9)
    assert ref["pair_f"][:, n_ph:, oracle.PF["max_collision_probability"]].max() > 1e-3   # and they matter
    res, safe = fo.trajectory_safety_assessment(objs[0])                  # served from the batch, reference's schema
    for oid in preds:
        assert oid in res["hr"] and oid in res["dce"]
        k = [s for pid, s in am.prediction_slots if pid == oid][0]
        np.testing.assert_allclose(res["hr"][oid]["max_collision_probability"],
                                   ref["pair_f"][0, k, oracle.PF["max_collision_probability"]], rtol=0, atol=1e-9)

    fo2, sc2, ego2, _ = _setup(tmp_path, max_agents=4)                    # default: the predictions are not evaluated
    fo2.evaluate_scenario(dict(preds), ego2[:2], float(ego2[2]), (0.0, 0.0), float(ego2[3]), 0, None)
    assert fo2.agent_manager.n_slots() == 4 * fo2.spawn_locator.R


def test_one_call_evaluate_scenario_equals_the_stage_calls(torch_cuda, tmp_path):
    """``accelerator.one_call`` (default True): evaluate_scenario queues the GPU side of the step as ONE native call (fo_step_run
    without candidates) -- visible area, cell classes, visible objects, spawn points, phantom agents and the sweep over them are
    those of the stage calls, bit for bit, in both spawn modes"""
    import yaml
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion import synthetic as SY
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    yaw = float(ego0[2])
    path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    veh = SimpleNamespace(length=SY.VEHICLE_BMW320I[0], width=SY.VEHICLE_BMW320I[1], wb_rear_axle=SY.VEHICLE_BMW320I[2],
                          mass=SY.VEHICLE_BMW320I[3], a_max=SY.VEHICLE_BMW320I[4])
    for mode in ("rules", "cells", "both"):
        got = {}
        for one_call in (True, False):
            with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
                cfg = yaml.safe_load(f)
            cfg["accelerator"]["spawn"].update(mode=mode, max_agents=12)
            cfg["accelerator"]["one_call"] = one_call
            cfg["metrics"]["metric_thresholds"].update(harm=0.1, risk=0.05)
            p = tmp_path / f"c_{mode}_{one_call}.yaml"
            p.write_text(yaml.safe_dump(cfg))
            fo = interface.FOInterface(sc, path, veh, 0.1, config_path=str(p))
            assert fo._one_call is one_call
            res = []
            for step in (0, 8, 25, 60):
                ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
                vis = fo.evaluate_scenario({}, ego, yaw, None, float(ego0[3]), step)
                traj = SY.make_trajectories(96, 31, 0.1, seed=step, ego_pos=ego, ego_yaw=yaw)
                ba = fo.trajectory_safety_assessment_batch(traj, mode="pair")
                torch_cuda.cuda.synchronize()
                b = fo.spawn_locator.batch
                pts = [(q.agent_type, q.source, tuple(q.position)) for q in fo.spawn_points]
                res.append(dict(cls=fo.sensor_model.cell_class.cpu().numpy().copy(), ring=vis.exterior.copy(),
                                vis_objs=list(fo.sensor_model.visible_objects_timestep),
                                occl={k: v.copy() for k, v in fo.sensor_model.obstacle_occlusions.items()},
                                pts=pts, n_agents=len(fo.agent_manager.phantom_agents),
                                pos=b.pos.cpu().numpy().copy(), ln=b.len.cpu().numpy().copy(),
                                cost=None if ba is None else ba.cost.cpu().numpy().copy(),
                                pair=None if ba is None else ba.result.pair_f.cpu().numpy().copy()))
            got[one_call] = res
        n_pts = 0
        for a, b in zip(got[True], got[False]):
            assert np.array_equal(a["cls"], b["cls"]) and np.array_equal(a["ring"], b["ring"])
            assert a["vis_objs"] == b["vis_objs"] and a["pts"] == b["pts"] and a["n_agents"] == b["n_agents"]
            assert a["occl"].keys() == b["occl"].keys() and all(np.array_equal(a["occl"][k], b["occl"][k]) for k in a["occl"])
            assert np.array_equal(a["ln"], b["ln"]) and np.array_equal(a["pos"][a["ln"] > 0], b["pos"][b["ln"] > 0])
            assert (a["cost"] is None) == (b["cost"] is None)
            if a["cost"] is not None:
                assert np.array_equal(a["cost"], b["cost"], equal_nan=True) and np.array_equal(a["pair"], b["pair"], equal_nan=True)
            n_pts += len(a["pts"])
        assert n_pts > 0, mode


def test_visible_object_bookkeeping_of_a_one_call_step_is_a_late_view(torch_cuda, tmp_path):
    """After a one-call ``evaluate_scenario`` the reference's visible-object side effects (sensor_model.py:58-101, 183) are
    views of the step's own device-to-host mirror (``fo_step_t::h_mirror``): nothing is copied until somebody looks, and what
    nobody looked at is applied when the obstacles move on -- ``last_visible_at_ts`` after a run in which the host never
    read a thing equals that of the stage calls, which read every step"""
    import yaml
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion import synthetic as SY
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    yaw = float(ego0[2])
    path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    veh = SimpleNamespace(length=SY.VEHICLE_BMW320I[0], width=SY.VEHICLE_BMW320I[1], wb_rear_axle=SY.VEHICLE_BMW320I[2],
                          mass=SY.VEHICLE_BMW320I[3], a_max=SY.VEHICLE_BMW320I[4])
    fos = {}
    for one_call in (True, False):
        with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
            cfg = yaml.safe_load(f)
        cfg["accelerator"]["one_call"] = one_call
        p = tmp_path / f"late_{one_call}.yaml"
        p.write_text(yaml.safe_dump(cfg))
        fos[one_call] = interface.FOInterface(sc, path, veh, 0.1, config_path=str(p))
    lazy, eager = fos[True], fos[False]
    steps = (0, 3, 8, 25, 40, 60)
    for step in steps:
        ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
        for fo in (lazy, eager):
            fo.evaluate_scenario({}, ego, yaw, None, float(ego0[3]), step)
        assert lazy.sensor_model._vis_pending is not None and lazy.fo_obstacles._pending is not None     # nobody has looked
    # the first look, one step late for none of them: everything of the last step and the memory of the earlier ones
    seen = [(o.obstacle_id, o.current_visible, o.last_visible_at_ts) for o in lazy.fo_obstacles]
    assert lazy.sensor_model._vis_pending is None and lazy.fo_obstacles._pending is None
    assert seen == [(o.obstacle_id, o.current_visible, o.last_visible_at_ts) for o in eager.fo_obstacles]
    assert any(v for _, v, _ in seen) and any(ts not in (None, steps[-1]) for _, _, ts in seen)
    assert lazy.sensor_model.visible_objects_timestep == eager.sensor_model.visible_objects_timestep
    a, b = lazy.fo_obstacles.visible_obstacle_multipolygon, eager.fo_obstacles.visible_obstacle_multipolygon
    assert len(a) == len(b) and all(np.array_equal(u, v) for u, v in zip(a, b))
    occ_a, occ_b = lazy.sensor_model.obstacle_occlusions, eager.sensor_model.obstacle_occlusions
    assert occ_a.keys() == occ_b.keys() and all(np.array_equal(occ_a[k], occ_b[k]) for k in occ_a)


def test_step_host_transfers_at_the_c_boundary(torch_cuda, tmp_path):
    """``fo_step_t::h_obstacles`` / ``h_mirror`` (ABI 12): the caller's obstacle buffer is free again when ``fo_step_run`` returns
    (the rows were staged), the mirror holds the step's hit ids and visibility flags once ``fo_step_mirror_wait`` returns, rows
    beyond the staging slot are refused, and a sensor model staging more than the slot holds falls back to its own copy"""
    import ctypes as C
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion import synthetic as SY
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    yaw = float(ego0[2])
    path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    veh = SimpleNamespace(length=SY.VEHICLE_BMW320I[0], width=SY.VEHICLE_BMW320I[1], wb_rear_axle=SY.VEHICLE_BMW320I[2],
                          mass=SY.VEHICLE_BMW320I[3], a_max=SY.VEHICLE_BMW320I[4])
    fo = interface.FOInterface(sc, path, veh, 0.1)
    sm = fo.sensor_model
    fo.evaluate_scenario({}, ego0[:2], yaw, None, float(ego0[3]), 8)
    step, s = fo._scene_step, fo._scene_step._s
    assert s.h_mirror and s.mirror_bytes == sm._buf["hv"].numel() and sm._obst_host is None      # (consumed by the call)
    # the same step by hand: scribble over the host rows right after the call -- the device must hold the real ones
    fo.fo_obstacles.update(8)
    rows = fo.fo_obstacles.packed().copy()
    sm.stage_obstacles(fo.fo_obstacles)
    sm._obst_host = rows
    step.run(ego0[:2], yaw, float(ego0[3]), None)
    rows[:] = 0xAB
    fo.ctx.call("fo_step_mirror_wait")
    torch_cuda.cuda.synchronize()
    assert sm._obst_dev.cpu().numpy().tobytes() == fo.fo_obstacles.packed().tobytes()
    assert np.array_equal(sm._buf["hv_host"].numpy(), sm._buf["hv"].cpu().numpy())
    assert sm._buf["hv_host"].numpy()[4 * sm.n_rays:4 * sm.n_rays + len(fo.fo_obstacles)].any()       # somebody is visible
    # both mirror paths in one process (the knob is looked at on every call): the kernels' own posted stores (default) and the
    # copy command behind the step (FO_STEP_MIRROR_COPY=1) leave the same bytes
    direct = sm._buf["hv_host"].numpy().copy()
    sm._buf["hv_host"].zero_()
    os.environ["FO_STEP_MIRROR_COPY"] = "1"
    try:
        sm.stage_obstacles(fo.fo_obstacles)
        step.run(ego0[:2], yaw, float(ego0[3]), None)
        fo.ctx.call("fo_step_mirror_wait")
    finally:
        del os.environ["FO_STEP_MIRROR_COPY"]
    assert np.array_equal(sm._buf["hv_host"].numpy(), direct) and direct[:4 * sm.n_rays].view(np.int32).max() >= 0
    sm._buf["hv_host"].zero_()
    sm.stage_obstacles(fo.fo_obstacles)
    step.run(ego0[:2], yaw, float(ego0[3]), None)          # and back to the direct stores
    fo.ctx.call("fo_step_mirror_wait")
    assert np.array_equal(sm._buf["hv_host"].numpy(), direct)
    # rows that do not fit the staging slot: refused by the C entry ...
    big = np.zeros(70000, dtype=np.uint8)
    s.h_obstacles, s.obstacles_bytes = big.ctypes.data, big.nbytes
    rc = fo.ctx._lib.fo_step_run(fo.ctx._h, C.byref(s), N.current_stream(sm._dev_index))
    assert rc != 0 and b"staging slot" in fo.ctx._lib.fo_last_error(fo.ctx._h)
    s.h_obstacles, s.obstacles_bytes = None, 0
    # ... and never offered by the sensor model: a crowd beyond the slot takes the plain copy
    crowd = type(fo.fo_obstacles)(list(sc.obstacles) * 130)
    crowd.update(8)
    assert crowd.packed().nbytes > sm.STAGE_BYTES
    sm.stage_obstacles(crowd)
    assert sm._obst_host is None and sm._obst[3] == len(crowd)
