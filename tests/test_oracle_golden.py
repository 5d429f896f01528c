"""Pins the CPU oracle (oracle/fo_oracle.c) against outputs of the reference's own CP / HR / TTC / TTCE / WTTC
code (fixtures made by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from golden_util import CASES, load_case

TOL = 1e-12  # oracle is float64 like the reference; closed-form CP == mvnun to ~2e-16


@pytest.mark.parametrize("name", CASES)
def test_cp_harm_risk_lists_match_reference(oracle, name):
    g, traj, agents, veh, dt = load_case(name)
    out = oracle.sweep(traj, agents, veh, dt)
    lists = out["lists"]  # [M,A,5,T-1]
    for key, slot in (("cp", 0), ("ego_harm", 1), ("obst_harm", 2), ("ego_risk", 3), ("obst_risk", 4)):
        ref = g["ref_" + key]
        got = lists[:, :, slot, :]
        assert np.array_equal(np.isnan(ref), np.isnan(got)), key
        np.testing.assert_allclose(got, ref, rtol=0, atol=TOL, equal_nan=True, err_msg=key)


@pytest.mark.parametrize("name", CASES)
def test_hr_scalars_match_reference(oracle, name):
    g, traj, agents, veh, dt = load_case(name)
    out = oracle.sweep(traj, agents, veh, dt)
    pf, pi, cost = out["pair_f"], out["pair_i"], out["cost"]
    for key in ("max_ego_risk", "max_obst_risk", "max_obst_harm_with_cp", "max_ego_harm", "max_obst_harm",
                "max_collision_probability"):
        np.testing.assert_allclose(pf[:, :, oracle.PF[key]], g["ref_" + key], rtol=0, atol=TOL, err_msg=key)
    # argmax of the risk list: only meaningful where the maximum is above the reference's own numerical noise
    # (MVNDST sorts the two variables and sums four tail products; below ~1e-16 its result is rounding residue,
    # e.g. 3.7e-51 where the true box probability is 1e-102 -- see DESIGN.md "CP tails").
    sig = g["ref_max_obst_risk"] > 1e-12
    assert np.array_equal(pi[:, :, oracle.PI["max_obst_risk_index"]][sig], g["ref_max_obst_risk_index"][sig])
    assert sig.sum() > 0 or name in ("short_traj", "angle_bins")   # (cases whose agents never enter the gate)
    for key in ("max_ego_risk_all", "max_obst_risk_all", "max_ego_harm_all", "max_obst_harm_all",
                "max_collision_probability_all", "max_obst_harm_with_cp_all"):
        np.testing.assert_allclose(cost[:, oracle.COST[key]], g["ref_" + key], rtol=0, atol=TOL, err_msg=key)


@pytest.mark.parametrize("name", ["probe_ped_crossing", "random_equal_len"])
def test_ttc_ttce_wttc_postprocessing_matches_reference(oracle, name):
    """TTC/TTCE/WTTC are pure functions of (dce, time_dce): restate them on the fixture's chosen inputs."""
    g, _, _, _, dt = load_case(name)
    dce, tdce = g["in_dce"], g["in_time_dce"]
    r3 = np.vectorize(oracle.lib().fo_oracle_round3)
    ttc = np.where(np.abs(dce) <= 1e-8, r3(tdce * dt), np.inf)
    ttce = r3(tdce * dt)
    assert np.array_equal(ttc, g["ref_ttc"])
    assert np.array_equal(ttce, g["ref_ttce"])
    assert np.array_equal(ttc.min(axis=1), g["ref_wttc"])


def test_probe_headline_numbers(oracle):
    """The single pedestrian-crossing probe: a few scalars written out for humans."""
    g, traj, agents, veh, dt = load_case("probe_ped_crossing")
    out = oracle.sweep(traj, agents, veh, dt)
    cp = out["lists"][0, 0, 0]
    assert cp.shape == (30,)
    assert abs(cp.max() - float(g["ref_cp"].max())) < TOL
    assert int(np.argmax(cp)) == int(np.argmax(g["ref_cp"][0, 0]))


def test_safety_decision_matches_the_references_metric_class(oracle):
    """Metric.evaluate_metrics of the reference itself (metric.py:35-100: threshold logic; :125-147: which metrics get
    evaluated) decided `safe` for ten metric / threshold configurations -- thresholds without their metric, thresholds
    the reference never checks (wttc, ttce), the cp threshold that needs 'hr'.  Only its 'dce' inputs came from this
    oracle (GEOS is not installable here)."""
    from golden_util import load_threshold_case
    traj, agents, veh, dt, configs = load_threshold_case()
    mixed = 0
    for activated, thr, evaluated, safe_ref in configs:
        t = {k: v for k, v in thr.items() if k in ("harm", "risk", "be", "cp", "ttc", "dce") and v is not None}
        out = oracle.sweep(traj, agents, veh, dt, metrics=tuple(activated), thr=t)
        assert np.array_equal(out["safe"].astype(bool), safe_ref), (activated, thr)
        mixed += 0 < safe_ref.mean() < 1
    assert mixed >= 6


def test_pedestrian_predictions_match_the_references_agent_class(oracle):
    """OAPPedestrianAgent of the reference (agent.py:429-536, imported unmodified under stub modules) produced
    tests/golden/ped_predictions.npz for 24 spawn points that bring their orientation: velocity components rounded to
    1e-3 (Q12), int(horizon/dt)+1 samples (21 for a horizon of 2.05 s), covariance growth, shape inflation.  Pinned here:
    the oracle's fo_oracle_cv_predictions (what the GPU spawn kernel is compared with bit for bit) and the host's
    FOAgentManager.add_agent path."""
    import os
    from types import SimpleNamespace
    from golden_util import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "ped_predictions.npz")))
    dt, n = float(g["dt"]), len(g["speed"])
    for T in sorted(set(g["ref_len"].tolist())):
        sel = np.nonzero(g["ref_len"] == T)[0]
        pos, yaw, v, cov = oracle.cv_predictions(g["pos0"][sel], g["yaw"][sel], g["speed"][sel], int(T), dt, 0.1,
                                                 float(g["cfg_variance_factor"]))
        assert np.array_equal(pos, g["ref_pos"][sel, :T]) and np.array_equal(yaw, g["ref_yaw"][sel, :T])
        assert np.array_equal(v, g["ref_v"][sel, :T])
        np.testing.assert_allclose(cov, g["ref_cov"][sel, :T], rtol=1e-15, atol=0)
    # the host path of manually added agents
    from frenetix_occlusion.agent import FOAgentManager
    cfg = {"pedestrian": {"default_velocity": 1.4, "length": float(g["raw_length"]), "width": float(g["raw_width"])},
           "prediction": {"size_factor_length_s": float(g["cfg_size_factor_length_s"]),
                          "size_factor_width_s": float(g["cfg_size_factor_width_s"]),
                          "size_factor_length_l": 1.4, "size_factor_width_l": 2.5,
                          "variance_factor": float(g["cfg_variance_factor"])}}
    am = FOAgentManager(SimpleNamespace(obstacles=[]), np.array([[0.0, 0.0], [10.0, 0.0]]), cfg, 0, dt=dt, device="cpu")
    for i in range(n):
        a = am.add_agent(g["pos0"][i], velocity=float(g["speed"][i]), agent_type="Pedestrian", timestep=0,
                         horizon=float(g["horizon"][i]), orientation=float(g["yaw"][i]))
        p, L = a.predictions[0], int(g["ref_len"][i])
        assert len(p["pos_list"]) == L
        assert np.array_equal(p["pos_list"], g["ref_pos"][i, :L]) and np.array_equal(p["v_list"], g["ref_v"][i, :L])
        assert np.array_equal(p["orientation_list"], g["ref_yaw"][i, :L])
        assert np.array_equal(p["cov_list"], g["ref_cov"][i, :L])
        assert (p["shape"]["length"], p["shape"]["width"]) == tuple(g["ref_shape"][i])


def test_obstacle_state_cache_matches_the_references():
    """FOObstacles of the reference (utils/fo_obstacle.py + helper_functions.calc_corner_points, imported unmodified
    under inert shapely stubs) recorded presence, pose and corner points of six obstacles over sixteen time steps:
    tests/golden/obstacle_states.npz.  Pinned: this package's FOObstacles / Obstacle.pose_at / Obstacle.corners, i.e.
    what is uploaded to the ray-cast kernels."""
    import os
    from golden_util import GOLDEN
    from frenetix_occlusion.scenario import Obstacle
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    g = dict(np.load(os.path.join(GOLDEN, "obstacle_states.npz")))
    obs = [Obstacle(100 + i, str(g["role"][i]).lower(), "car", float(g["length"][i]), float(g["width"][i]), int(g["t0"][i]),
                    g["initial"][i], g["states"][i, :int(g["n_states"][i])]) for i in range(len(g["t0"]))]
    fo = FOObstacles(obs)
    for t in range(g["ref_present"].shape[1]):
        fo.update(t)
        corn, cen, flags = fo.arrays()
        for i, o in enumerate(fo):
            assert (o.current_pos is not None) == bool(g["ref_present"][i, t]), (i, t)
            assert bool(flags[i] & 1) == bool(g["ref_present"][i, t])
            if o.current_pos is not None:
                assert np.array_equal(o.current_pos, g["ref_pos"][i, t]) and o.current_orientation == g["ref_yaw"][i, t]
                np.testing.assert_allclose(o.current_corner_points, g["ref_corners"][i, t], rtol=0, atol=1e-13)
                np.testing.assert_allclose(corn[i], g["ref_corners"][i, t], rtol=0, atol=1e-13)


def test_route_enumeration_matches_the_references_route_planner():
    """FORoutePlanner._find_all_routes of the reference (route_planner.py:54-90, imported unmodified) listed the candidate
    routes from every start lanelet of the three example scenarios' networks and of two random ones:
    tests/golden/routes.npz.  Pinned: scenario.enumerate_routes (same routes, same order), which feeds the route table
    of the phantom-vehicle predictions."""
    import os
    from golden_util import GOLDEN
    from frenetix_occlusion.scenario import Lanelet, enumerate_routes
    g = dict(np.load(os.path.join(GOLDEN, "routes.npz")))
    z = np.zeros((2, 2))
    total = 0
    for q in range(int(g["n_nets"])):
        ids, so, sf = g[f"net{q}_id"], g[f"net{q}_succ_off"], g[f"net{q}_succ"]
        lls = []
        for i, lid in enumerate(ids):
            al, ar = int(g[f"net{q}_adj_left"][i]), int(g[f"net{q}_adj_right"][i])
            lls.append(Lanelet(int(lid), z, z, successors=[int(s) for s in sf[so[i]:so[i + 1]]],
                               adj_left=None if al < 0 else al, adj_left_same_direction=bool(g[f"net{q}_adj_left_same"][i]),
                               adj_right=None if ar < 0 else ar, adj_right_same_direction=bool(g[f"net{q}_adj_right_same"][i])))
        mine = enumerate_routes(lls, max_depth=2)
        ref = {}
        off, flat = g[f"net{q}_route_off"], g[f"net{q}_route_flat"]
        for r, start in enumerate(g[f"net{q}_route_start"]):
            ref.setdefault(int(start), []).append([int(x) for x in flat[off[r]:off[r + 1]]])
        assert set(mine) == set(ref)
        for lid in ref:
            assert mine[lid] == ref[lid], (q, lid)
            total += len(ref[lid])
    assert total > 300


def test_brake_evaluation_matches_the_references_own_bisection(oracle):
    """metrics/be.py of the reference, imported unmodified (gen_golden.py be), ran its bisection -- bracket from the candidate's
    own strongest deceleration rounded to two decimals, <= 10 halvings, stop below 0.1, a constant deceleration from the second
    sample on, the path re-sampled by scipy's interp1d over the travelled chord length -- on 40 candidates x 10 predictions;
    only `intersects` of two rectangles came from this repository (shapely is not installable here).  The oracle reproduces
    every required deceleration and brake threat number EXACTLY (they are dyadic midpoints of the bracket: a single different
    collision verdict anywhere in a bisection would move the result by >= 0.07), and which pairs have a value at all."""
    from golden_util import load_be_case
    g, traj, agents, veh, dt = load_be_case()
    ref = oracle.sweep(traj, agents, veh, dt, metrics=("dce", "ttc", "be"))
    d, b = ref["pair_f"][..., oracle.PF["be_decel"]], ref["pair_f"][..., oracle.PF["be_btn"]]
    assert d.shape == g["be_decel"].shape == (40, 10)
    assert np.array_equal(d, g["be_decel"]) and np.array_equal(b, g["be_btn"])
    has = g["be_decel"] > 0
    assert has.sum() > 100 and len(np.unique(g["be_decel"][has])) > 40 and (~has).sum() > 100      # both kinds of pairs
    # a value exactly where the reference's own TTC saw a collision after t = 0
    assert np.array_equal(has, np.isfinite(g["ttc"]) & (g["ttc"] > 0))
    np.testing.assert_array_equal(np.nan_to_num(ref["pair_f"][..., oracle.PF["ttc"]], posinf=np.inf), g["ttc"])
    assert np.array_equal(ref["cost"][:, oracle.COST["max_btn"]], g["be_btn"].max(axis=1))


def test_dce_loop_matches_the_references_own_walk_over_the_time_steps(oracle):
    """metrics/dce.py of the reference, imported unmodified (gen_golden.py dce), walked the time steps of 31 candidates x 16
    predictions -- np.round(distance, 3), first strict minimum, the stop at the first zero and where a prediction ends -- and its
    TTC / TTCE / WTTC read the results; only the distance between two rectangles under the loop came from this repository
    (shapely is not installable here; that primitive is pinned to exact arithmetic by tests/test_dce_sympy.py).  The oracle
    reproduces dce, time_dce, ttc, ttce and wttc exactly: ties on stationary pairs (every step), two equal minima of a
    symmetric pass-by, touching faces (distance 0), predictions of 1 ... 40 samples against 31 of the candidate."""
    from golden_util import load_dce_case
    g, traj, agents, veh, dt = load_dce_case()
    ref = oracle.sweep(traj, agents, veh, dt, metrics=("dce", "ttc", "ttce", "wttc"))
    assert np.array_equal(ref["pair_f"][..., oracle.PF["dce"]], g["ref_dce"])
    assert np.array_equal(ref["pair_i"][..., oracle.PI["time_dce"]], g["ref_time_dce"])
    assert np.array_equal(ref["pair_f"][..., oracle.PF["ttc"]], g["ref_ttc"])            # inf where the reference says np.inf
    assert np.array_equal(ref["pair_f"][..., oracle.PF["ttce"]], g["ref_ttce"])
    assert np.array_equal(ref["cost"][:, oracle.COST["wttc"]], g["ref_wttc"])
    td = g["ref_time_dce"]
    assert (g["ref_dce"] == 0).sum() > 50 and len(np.unique(td)) == 31 and (td > 0).sum() > 300


def test_relevant_lanelets_nearest_index_and_intention_match_the_references_spawn_locator():
    """tests/golden/relevant_lanelets.npz (gen_golden.py relevant): the reference's OWN ``SpawnLocator`` helpers under the
    dynamic-obstacle rule -- first intersection in list order that holds the ego's lanelet, incoming + successor lanelets
    without the ego's; without one: the left neighbours of the first lanelets under every fifth vertex of the window
    (spawn_locator.py:186-202, 584-635, 666-676) --, ``_find_nearest_index`` (:744-750, ties included) and the thresholds of
    ``_find_ego_intention`` (:729-741).  The rule checker (oracle/fo_spawn_rules_ref.py, what the device is compared with) and
    the host's window / intention code reproduce all of it."""
    import os
    from types import SimpleNamespace as NS
    from golden_util import GOLDEN
    from frenetix_occlusion.spawn_locator import SpawnLocator, intention_from_curvature
    from oracle.fo_spawn_rules_ref import SpawnRules
    g = np.load(os.path.join(GOLDEN, "relevant_lanelets.npz"), allow_pickle=False)
    n_inter_cases = 0
    for c in range(int(g["n_cases"])):
        ids, adj = g[f"c{c}_ids"].tolist(), g[f"c{c}_adj"].tolist()
        lanelets = {i: NS(lanelet_id=i, adj_left=(None if a < 0 else a)) for i, a in zip(ids, adj)}
        inters = [dict(incomings=[]) for _ in range(int(g[f"c{c}_n_inter"]))]
        flat, pos = g[f"c{c}_inter_flat"].tolist(), 0
        for a, b, kind, cnt in g[f"c{c}_inter_meta"].tolist():
            inc = inters[a]["incomings"]
            while len(inc) <= b:
                inc.append(dict(incoming=[], left=[], right=[], straight=[]))
            inc[b][("incoming", "left", "right", "straight")[kind]] = flat[pos:pos + cnt]
            pos += cnt
        off, at_flat = g[f"c{c}_at_off"].tolist(), g[f"c{c}_at_flat"].tolist()
        pts = [np.array([float(k), float(c)]) for k in range(len(off) - 1)]
        table = {tuple(q): at_flat[off[k]:off[k + 1]] for k, q in enumerate(pts)}
        ego = np.array([-1.0, float(c)])
        table[tuple(ego)] = g[f"c{c}_at_ego"].tolist()
        r = object.__new__(SpawnRules)
        r.intersections, r.reference, r.ego_pos = inters, pts, ego
        r.lanelet_of = lambda xy: (lanelets[table[tuple(np.asarray(xy, dtype=np.float64))][0]]
                                   if table[tuple(np.asarray(xy, dtype=np.float64))] else None)
        it, rel, inner = r._relevant_lanelet_ids()
        found, want_rel, want_inner = int(g[f"c{c}_found"]), g[f"c{c}_rel"].tolist(), g[f"c{c}_inner"].tolist()
        assert (-1 if it is None else [id(x) for x in inters].index(id(it))) == found, c
        if found >= 0:
            n_inter_cases += 1
            assert sorted(rel) == want_rel and sorted(inner) == want_inner, c
        else:                                  # (the reference's list may name a neighbour twice and holds None for "no neighbour")
            assert rel == set(want_rel) - {-1} and inner == set(), c
    assert n_inter_cases >= 20
    # the window's end points: np.argmin(|s - q|), the first of equally near vertices
    sl = object.__new__(SpawnLocator)
    sl._s_list = g["path_s"].tolist()
    assert [sl._nearest_vertex(float(q)) for q in g["query_s"]] == g["nearest"].tolist()
    # the intention thresholds
    off, flat = g["curv_off"].tolist(), g["curv_flat"]
    names = ("straight ahead", "left turn", "right turn")
    for k in range(len(off) - 1):
        kk = flat[off[k]:off[k + 1]]
        assert intention_from_curvature(kk) == int(g["intention"][k]) and SpawnRules.intention_of(kk) == names[int(g["intention"][k])]
    assert set(g["intention"].tolist()) == {0, 1, 2}


def test_find_spawn_points_orchestration_matches_the_references():
    """the same fixture's second half: the reference's own ``find_spawn_points`` (spawn_locator.py:80-143) with its three rule
    methods replaced by recorders -- ``s_threshold`` (:113, incl. 4 v = 25 exactly), the 40 m window's vertex range (:678-693),
    which family runs under which intention and switch (:124-138) and in which order their points are appended (None results
    and None entries dropped, :637-645).  The host's ``rule_params`` gives the same threshold, window and intention; the rule
    checker's ``find`` calls the same families and returns the points in the same order."""
    import os
    from types import SimpleNamespace as NS
    from golden_util import GOLDEN
    import frenetix_occlusion.utils.curvilinear as CV
    from frenetix_occlusion.spawn_locator import SpawnLocator
    import oracle.fo_spawn_rules_ref as R
    g = np.load(os.path.join(GOLDEN, "relevant_lanelets.npz"), allow_pickle=False)
    rows, toks, off = g["orch"], g["orch_tokens"].tolist(), g["orch_tok_off"].tolist()
    names = ("straight ahead", "left turn", "right turn")
    n_turn = 0
    keep_cv, keep_r = CV.curvature, R.curvature
    try:
        for c, row in enumerate(rows):
            n, ego_s, ego_v = int(row[0]), float(row[1]), float(row[2])
            sw, kind = [bool(x) for x in row[3:6]], [int(x) for x in row[6:9]]
            s_thr, i0, n_ref, inten = float(row[9]), int(row[10]), int(row[11]), int(row[12])
            called, want = [bool(x) for x in row[13:16]], toks[off[c]:off[c + 1]]
            s_arr, k = g[f"o{c}_s"], g[f"o{c}_k"]
            CV.curvature = R.curvature = lambda ref, k=k: k          # (commonroad_dc's curvature is an input of the fixture)
            # host: threshold, window, intention
            sl = object.__new__(SpawnLocator)
            sl._rules_ready, sl._s_list, sl.ref_path = True, s_arr.tolist(), np.stack((np.arange(n, dtype=np.float64), np.zeros(n)), -1)
            sl._rule_cfg = dict(ped_width=0.5, ped_length=0.3, behind_static=int(sw[1]), behind_turn=int(sw[2]), behind_dynamic=int(sw[0]),
                                max_static=1, max_dynamic=1)
            sl.sensor_model = NS()
            p = sl.rule_params(np.zeros(2), 0.0, np.array([ego_s, 0.3]), ego_v)
            assert p.s_threshold == s_thr, c
            if n_ref > 0:
                assert (p.win_i0, p.win_i1) == (i0, i0 + n_ref), c
            else:
                assert p.win_i1 <= p.win_i0, c
            if n_ref >= 3:       # (fewer vertices: commonroad_dc's curvature raises in the reference; the host says straight ahead)
                assert p.intention == inten and sl.last_intention == names[inten], c
            # checker: which families run, what comes back in which order
            r = object.__new__(R.SpawnRules)
            r.behind_dynamic, r.behind_static, r.behind_turn = sw
            r.ref_path, r.ref_s = sl.ref_path, s_arr
            calls = []
            ret = lambda base, k_: [] if k_ < 2 else [base] if k_ == 2 else [base, base + 1]
            r.behind_dynamic_obstacle = lambda view: (calls.append(0), ret(100, kind[0]))[1]
            r.behind_static_obstacle = lambda view: (calls.append(1), ret(200, kind[1]))[1]
            r.behind_turn_point = lambda view, intention: (calls.append(2), None if kind[2] == 0 else 300)[1]
            if n_ref >= 3:
                got = r.find(None, np.zeros(2), np.array([ego_s, 0.3]), ego_v)
                assert r.s_threshold == s_thr and r.last_intention == names[inten], c
                assert [q in calls for q in range(3)] == called and got == want, c
                n_turn += called[2]
    finally:
        CV.curvature, R.curvature = keep_cv, keep_r
    assert n_turn >= 5
