"""Analytic known-answer tests for the parts of the oracle the reference delegates to shapely/GEOS + commonroad
(DCE geometry, Metric thresholds).  These are what pins those parts ("parity unpinned" vs the reference)."""
import math

import numpy as np
import pytest

VEH = (4.508, 1.610, 1.4227, 1093.3, 11.5)


def rect(o, cx, cy, yaw, l, w):
    return o.rect_vertices(cx, cy, yaw, l, w)


def test_rect_vertex_order_matches_commonroad_rectangle(oracle):
    q = rect(oracle, 1.0, 2.0, 0.0, 4.0, 2.0)
    np.testing.assert_allclose(q, [[-1, 1], [-1, 3], [3, 3], [3, 1]])
    q = rect(oracle, 0.0, 0.0, math.pi / 2, 4.0, 2.0)
    np.testing.assert_allclose(q, [[1, -2], [-1, -2], [-1, 2], [1, 2]], atol=1e-15)


@pytest.mark.parametrize("dx,dy,expect", [
    (10.0, 0.0, 10.0 - 2.0 - 1.0),          # face to face along x: gap = 10 - l1/2 - l2/2
    (0.0, 5.0, 5.0 - 1.0 - 0.5),            # face to face along y
    (10.0, 5.0, math.hypot(7.0, 3.5)),      # corner to corner
    (3.0, 0.0, 0.0),                        # touching faces -> 0
    (2.0, 0.5, 0.0),                        # overlapping
    (0.0, 0.0, 0.0),                        # contained
])
def test_axis_aligned_distance(oracle, dx, dy, expect):
    a = rect(oracle, 0, 0, 0, 4.0, 2.0)
    b = rect(oracle, dx, dy, 0, 2.0, 1.0)
    assert oracle.quad_distance(a, b) == pytest.approx(expect, abs=1e-14)
    assert oracle.quad_distance(b, a) == pytest.approx(expect, abs=1e-14)


def test_rotated_square_vertex_to_face(oracle):
    a = rect(oracle, 0, 0, 0, 2.0, 2.0)
    b = rect(oracle, 5.0, 0.0, math.pi / 4, 2.0, 2.0)  # diamond, nearest vertex at x = 5 - sqrt(2)
    assert oracle.quad_distance(a, b) == pytest.approx(5.0 - math.sqrt(2.0) - 1.0, abs=1e-14)


def test_cross_shaped_overlap_without_contained_vertices(oracle):
    a = rect(oracle, 0, 0, 0, 10.0, 1.0)
    b = rect(oracle, 0, 0, math.pi / 2, 10.0, 1.0)
    assert oracle.quad_distance(a, b) == 0.0


def test_edge_to_edge_parallel_rotated(oracle):
    yaw = 0.3
    a = rect(oracle, 0, 0, yaw, 4.0, 2.0)
    n = np.array([-math.sin(yaw), math.cos(yaw)])
    c = 3.7 * n
    b = rect(oracle, c[0], c[1], yaw, 4.0, 1.0)
    assert oracle.quad_distance(a, b) == pytest.approx(3.7 - 1.0 - 0.5, abs=1e-14)


def _straight_traj(T, v, y=0.0, dt=0.1):
    t = np.arange(T) * dt
    return {"x": (v * t)[None], "y": np.full((1, T), y), "theta": np.zeros((1, T)), "v": np.full((1, T), v),
            "a": np.zeros((1, T))}


def _static_agent(x, y, yaw, L, T=31, dims=(2.0, 1.0), typ=0):
    return {"pos": np.tile([[x, y]], (T, 1))[None], "yaw": np.full((1, T), yaw), "v": np.zeros((1, T)),
            "cov": np.tile(np.eye(2) * 0.1, (1, T, 1, 1)), "shape": np.array([[dims[0] * 1.2, dims[1] * 1.3]]),
            "raw_dims": np.array([dims]), "type": np.array([typ], dtype=np.int32), "len": np.array([L], dtype=np.int32)}


def test_dce_closed_form_and_rounding(oracle):
    """ego centre = rear axle + wb; passes a static box 3 m to the side: min gap = 3 - w_e/2 - w_a/2."""
    traj = _straight_traj(31, 10.0)
    ag = _static_agent(12.0, 3.0, 0.0, 31)
    out = oracle.sweep(traj, ag, VEH, 0.1)
    gap = 3.0 - VEH[1] / 2 - 0.5
    assert out["pair_f"][0, 0, oracle.PF["dce"]] == round(gap, 3)
    # first timestep at which the x-extents overlap: ego centre x = 10 t*0.1 + wb; |12 - cx| <= l_e/2 + 1
    cx = 10.0 * np.arange(31) * 0.1 + VEH[2]
    first = int(np.argmax(np.abs(12.0 - cx) <= VEH[0] / 2 + 1.0))
    assert out["pair_i"][0, 0, oracle.PI["time_dce"]] == first  # ties keep the earliest index (dce.py:82)
    assert out["pair_f"][0, 0, oracle.PF["ttc"]] == np.inf
    assert out["pair_f"][0, 0, oracle.PF["ttce"]] == round(first * 0.1, 3)


def test_dce_stops_at_first_collision_and_ttc(oracle):
    traj = _straight_traj(31, 10.0)
    ag = _static_agent(15.0, 0.0, 0.0, 31)
    out = oracle.sweep(traj, ag, VEH, 0.1)
    cx = 10.0 * np.arange(31) * 0.1 + VEH[2]
    hit = int(np.argmax(15.0 - 1.0 - (cx + VEH[0] / 2) < 5e-4))  # Q7: rounded to 0 below 5e-4
    assert out["pair_f"][0, 0, oracle.PF["dce"]] == 0.0
    assert out["pair_i"][0, 0, oracle.PI["time_dce"]] == hit
    assert out["pair_f"][0, 0, oracle.PF["ttc"]] == round(hit * 0.1, 3)
    assert out["cost"][0, oracle.COST["wttc"]] == round(hit * 0.1, 3)


def test_dce_stops_when_agent_prediction_ends(oracle):
    traj = _straight_traj(31, 10.0)
    ag = _static_agent(15.0, 0.0, 0.0, 5)  # agent known for 5 steps only: never reached
    out = oracle.sweep(traj, ag, VEH, 0.1)
    cx4 = 10.0 * 4 * 0.1 + VEH[2]
    assert out["pair_f"][0, 0, oracle.PF["dce"]] == round(15.0 - 1.0 - (cx4 + VEH[0] / 2), 3)
    assert out["pair_i"][0, 0, oracle.PI["time_dce"]] == 4


def test_ttc_collision_boundary_5e_4(oracle):
    """Q7: collision <=> raw distance rounds to 0.000, i.e. raw < 5e-4."""
    T = 3
    for gap, collide in ((4.0e-4, True), (6.0e-4, False)):
        x_rear = 20.0 - 1.0 - gap - VEH[0] / 2 - VEH[2]
        traj = {"x": np.full((1, T), x_rear), "y": np.zeros((1, T)), "theta": np.zeros((1, T)),
                "v": np.zeros((1, T)), "a": np.zeros((1, T))}
        out = oracle.sweep(traj, _static_agent(20.0, 0.0, 0.0, T, T=T), VEH, 0.1)
        assert (out["pair_f"][0, 0, oracle.PF["ttc"]] == 0.0) == collide


def test_thresholds_and_metric_selection(oracle):
    traj = _straight_traj(31, 10.0)
    ag = _static_agent(15.0, 0.4, 0.0, 31, typ=4, dims=(0.3, 0.5))
    base = oracle.sweep(traj, ag, VEH, 0.1, thr={"harm": 1, "risk": 1})
    assert base["safe"][0] == 1  # harm <= 1 always
    c = base["cost"][0]
    assert c[oracle.COST["max_obst_harm_with_cp_all"]] > 0.1
    assert oracle.sweep(traj, ag, VEH, 0.1, thr={"harm": 0.1})["safe"][0] == 0
    assert oracle.sweep(traj, ag, VEH, 0.1, thr={"cp": 0.01})["safe"][0] == 0
    assert oracle.sweep(traj, ag, VEH, 0.1, thr={"ttc": 5.0})["safe"][0] == 0
    assert oracle.sweep(traj, ag, VEH, 0.1, thr={"dce": 0.5})["safe"][0] == 0
    # thresholds of metrics that are not activated are ignored (metric.py:61-98 test "'x' in results")
    only_cp = oracle.sweep(traj, ag, VEH, 0.1, metrics=("cp",), thr={"ttc": 5.0, "dce": 0.5, "harm": 0.0})
    assert only_cp["safe"][0] == 1
    assert np.isnan(only_cp["pair_f"][0, 0, oracle.PF["dce"]])
    # dependency closure (metric.py:125-147)
    L = oracle.lib()
    B = oracle.METRIC_BITS
    assert L.fo_oracle_required_metrics(B["wttc"]) == B["wttc"] | B["ttc"] | B["dce"]
    assert L.fo_oracle_required_metrics(B["hr"]) == B["hr"] | B["cp"]
    assert L.fo_oracle_required_metrics(B["ttce"]) == B["ttce"] | B["dce"]


def test_no_agents_is_safe(oracle):
    traj = _straight_traj(31, 10.0)
    ag = {"pos": np.zeros((0, 31, 2)), "yaw": np.zeros((0, 31)), "v": np.zeros((0, 31)), "cov": np.zeros((0, 31, 2, 2)),
          "shape": np.zeros((0, 2)), "raw_dims": np.zeros((0, 2)), "type": np.zeros(0, np.int32), "len": np.zeros(0, np.int32)}
    out = oracle.sweep(traj, ag, VEH, 0.1, thr={"harm": 0.0, "ttc": 99})
    assert out["safe"][0] == 1 and out["cost"][0, oracle.COST["wttc"]] == np.inf


def test_box_prob_closed_form_against_scipy_mvn(oracle):
    """F7: diagonal covariance -> product of 1-D normal box probabilities; compare with the Fortran MVNDST."""
    mvn = pytest.importorskip("scipy.stats._mvn")
    rng = np.random.default_rng(7)
    worst = 0.0
    for _ in range(500):
        mu = rng.uniform(-3, 3, 2)
        lo = mu + rng.uniform(-6, 2, 2)
        hi = lo + rng.uniform(0.1, 4, 2)
        var = rng.uniform(0.05, 0.5)
        ref = mvn.mvnun(lo, hi, mu, np.eye(2) * var)[0]
        worst = max(worst, abs(ref - oracle.box_prob(lo, hi, mu, var, var)))
    assert worst < 1e-15


# ---------------------------------------------------------------------------------------------- BE (metrics/be.py)
ALL7 = ("hr", "ttc", "ttce", "dce", "wttc", "cp", "be")


def _ped(x, y, T=31):
    return {"pos": np.tile(np.array([[x, y]]), (T, 1))[None], "yaw": np.zeros((1, T)), "v": np.zeros((1, T)),
            "cov": np.tile(0.1 * np.eye(2), (1, T, 1, 1)), "shape": np.array([[0.36, 0.65]]),
            "raw_dims": np.array([[0.3, 0.5]]), "type": np.array([4], dtype=np.int32), "len": np.array([T], dtype=np.int32)}


def test_be_bisection_finds_the_braking_that_just_avoids_a_standing_pedestrian(oracle):
    T, v0 = 31, 10.0
    t = np.arange(T) * 0.1
    traj = {"x": (v0 * t)[None], "y": np.zeros((1, T)), "theta": np.zeros((1, T)), "v": np.full((1, T), v0),
            "a": np.zeros((1, T))}
    out = oracle.sweep(traj, _ped(20.0, 0.0), VEH, 0.1, metrics=ALL7, thr={"be": 0.25})
    pf = out["pair_f"][0, 0]
    assert np.isfinite(pf[oracle.PF["ttc"]]) and pf[oracle.PF["ttc"]] > 0
    decel = pf[oracle.PF["be_decel"]]
    # front of the ego is wb + L/2 = 3.677 m ahead of the rear axle, the pedestrian's near face at 19.85 m: 16.17 m
    # of travel; the left-Riemann speed profile [v0, v0, v0 - d dt, ...] travels v0 dt + v0^2/(2 d) + v0 dt/2
    d_star = v0 * v0 / (2.0 * (16.173 - 1.0 - 0.5))
    assert d_star - 0.15 < decel < d_star + 0.15
    assert (decel * 1024) == int(decel * 1024)                       # a bisection midpoint of [0, 5]
    assert pf[oracle.PF["be_btn"]] == decel / VEH[4]
    assert out["cost"][0, oracle.COST["max_btn"]] == decel / VEH[4]
    assert out["safe"][0] == (0 if decel / VEH[4] > 0.25 else 1)
    # the bracket starts at round(|min(a)|, 2) (be.py:68): a trajectory that already brakes with 4.004 m/s^2 somewhere
    traj["a"][0, 7] = -4.004
    d2 = oracle.sweep(traj, _ped(20.0, 0.0), VEH, 0.1, metrics=ALL7)["pair_f"][0, 0, oracle.PF["be_decel"]]
    assert 4.0 <= d2 <= 5.0 and d2 != decel


def test_be_is_zero_without_a_collision_and_for_a_collision_at_t0(oracle):
    T = 31
    t = np.arange(T) * 0.1
    traj = {"x": (5.0 * t)[None], "y": np.zeros((1, T)), "theta": np.zeros((1, T)), "v": np.full((1, T), 5.0),
            "a": np.zeros((1, T))}
    far = oracle.sweep(traj, _ped(60.0, 0.0), VEH, 0.1, metrics=ALL7)["pair_f"][0, 0]
    assert far[oracle.PF["be_decel"]] == 0.0 and far[oracle.PF["be_btn"]] == 0.0 and np.isinf(far[oracle.PF["ttc"]])
    now = oracle.sweep(traj, _ped(2.0, 0.0), VEH, 0.1, metrics=ALL7)["pair_f"][0, 0]   # overlapping at t = 0: ttc = 0
    assert now[oracle.PF["ttc"]] == 0.0 and now[oracle.PF["be_decel"]] == 0.0           # be.py:50 "if ttc > 0"
    # 'be' alone pulls in ttc and dce (metric.py:135-139; be.py:39 needs results['ttc'])
    m = oracle.lib().fo_oracle_required_metrics(oracle.METRIC_BITS["be"])
    assert m & oracle.METRIC_BITS["dce"] and m & oracle.METRIC_BITS["ttc"]


# ---------------------------------------------------------------------------------------------- DCE geometry: properties
def _brute_distance(a, b, n=200):
    """independent check: densely sampled boundary-to-boundary distance (0 if a vertex of one lies inside the other)"""
    def boundary(q):
        t = np.linspace(0.0, 1.0, n, endpoint=False)[:, None]
        return np.concatenate([q[i] + t * (q[(i + 1) % 4] - q[i]) for i in range(4)])

    def inside(p, q):
        s = np.array([(q[(i + 1) % 4][0] - q[i][0]) * (p[1] - q[i][1]) - (q[(i + 1) % 4][1] - q[i][1]) * (p[0] - q[i][0])
                      for i in range(4)])
        return (s >= 0).all() or (s <= 0).all()
    if any(inside(p, b) for p in a) or any(inside(p, a) for p in b):
        return 0.0
    pa, pb = boundary(a), boundary(b)
    d = np.sqrt(((pa[:, None, :] - pb[None, :, :]) ** 2).sum(-1))
    return float(d.min())


def test_rect_distance_properties_randomised(oracle):
    rng = np.random.default_rng(7)
    for _ in range(150):
        la, wa, lb, wb = rng.uniform(0.2, 6.0, 4)
        ca, cb = rng.uniform(-8, 8, 2), rng.uniform(-8, 8, 2)
        ya, yb = rng.uniform(-4, 4, 2)
        if rng.random() < 0.3:
            yb = ya + rng.choice([0.0, math.pi / 2, math.pi])         # parallel / perpendicular boxes
        a, b = rect(oracle, ca[0], ca[1], ya, la, wa), rect(oracle, cb[0], cb[1], yb, lb, wb)
        d = oracle.quad_distance(a, b)
        assert d >= 0.0 and d == pytest.approx(oracle.quad_distance(b, a), abs=1e-12)          # symmetric
        # invariant under a common rigid motion
        phi, sh = rng.uniform(-3, 3), rng.uniform(-50, 50, 2)
        R = np.array([[math.cos(phi), -math.sin(phi)], [math.sin(phi), math.cos(phi)]])
        assert oracle.quad_distance(a @ R.T + sh, b @ R.T + sh) == pytest.approx(d, abs=1e-10)
        # scales with the scene
        assert oracle.quad_distance(2.5 * a, 2.5 * b) == pytest.approx(2.5 * d, abs=1e-10)
        # never larger than the centre distance, never smaller than centre distance minus the circumradii
        cd = float(np.linalg.norm(ca - cb))
        assert d <= cd + 1e-12 and d >= cd - 0.5 * (math.hypot(la, wa) + math.hypot(lb, wb)) - 1e-12
        # agrees with a dense boundary sampling (which can only overestimate, by less than one sample spacing)
        bd = _brute_distance(a, b)
        assert d <= bd + 1e-9 and bd - d < 0.1
        assert (d == 0.0) == (bd == 0.0) or bd < 0.1


def test_dce_is_independent_of_the_order_of_equal_minima_and_uses_the_earliest(oracle):
    """two timesteps with exactly the same rounded distance: time_dce is the earlier one (dce.py:82 strict <)"""
    T = 9
    x = np.array([0.0, 1.0, 2.0, 3.0, 4.0, 3.0, 2.0, 3.0, 4.0])       # approaches, recedes, approaches to the same x
    traj = {"x": x[None], "y": np.zeros((1, T)), "theta": np.zeros((1, T)), "v": np.ones((1, T)), "a": np.zeros((1, T))}
    ag = _ped(12.0, 0.0, T)
    out = oracle.sweep(traj, ag, VEH, 0.1)
    assert out["pair_i"][0, 0, oracle.PI["time_dce"]] == 4
    gap = 12.0 - 0.15 - (4.0 + VEH[2] + VEH[0] / 2)
    assert out["pair_f"][0, 0, oracle.PF["dce"]] == pytest.approx(round(gap, 3), abs=1e-12)
