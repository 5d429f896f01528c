"""An independent Python restatement of the reference's DCE module (ref metrics/dce.py:52-99 +
metrics/utils/convert_dynamic_obstacle.py:17-86) against the C oracle on random batches.  The reference measures the
distance between two shapely polygons (GEOS, absent here); this file measures it by brute force -- segment-to-segment
distances over all 16 side pairs, zero if the rectangles intersect or one contains the other -- and walks the time
steps exactly like dce.py does.  CPU only."""
import math

import numpy as np

VEH = (4.508, 1.610, 1.4227, 1093.3, 11.5)


def _rect(cx, cy, yaw, length, width):
    """commonroad Rectangle [ext]: vertices (+-l/2, +-w/2) rotated by yaw about the centre, translated"""
    l2, w2 = length / 2.0, width / 2.0
    v = np.array([[-l2, -w2], [-l2, w2], [l2, w2], [l2, -w2]])
    c, s = math.cos(yaw), math.sin(yaw)
    return v @ np.array([[c, s], [-s, c]]) + np.array([cx, cy])


def _pt_seg(p, a, b):
    ex, ey = b[0] - a[0], b[1] - a[1]
    t = min(1.0, max(0.0, ((p[0] - a[0]) * ex + (p[1] - a[1]) * ey) / (ex * ex + ey * ey)))
    return math.hypot(p[0] - (a[0] + t * ex), p[1] - (a[1] + t * ey))


def _seg_cross(a, b, c, d):
    def orient(p, q, r):
        return (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
    o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
    return (o1 * o2 < 0 and o3 * o4 < 0)


def _inside(p, poly):
    s = [(poly[(i + 1) % 4][0] - poly[i][0]) * (p[1] - poly[i][1]) - (poly[(i + 1) % 4][1] - poly[i][1]) * (p[0] - poly[i][0])
         for i in range(4)]
    return all(v >= 0 for v in s) or all(v <= 0 for v in s)


def polygon_distance(pa, pb):
    """what shapely's Polygon.distance returns for two convex quadrilaterals"""
    for i in range(4):
        for j in range(4):
            if _seg_cross(pa[i], pa[(i + 1) % 4], pb[j], pb[(j + 1) % 4]):
                return 0.0
    if any(_inside(p, pb) for p in pa) or any(_inside(p, pa) for p in pb):
        return 0.0
    best = math.inf
    for i in range(4):
        for j in range(4):
            a0, a1, b0, b1 = pa[i], pa[(i + 1) % 4], pb[j], pb[(j + 1) % 4]
            best = min(best, _pt_seg(a0, b0, b1), _pt_seg(a1, b0, b1), _pt_seg(b0, a0, a1), _pt_seg(b1, a0, a1))
    return best


def dce_like_the_reference(x, y, theta, agent_pos, agent_yaw, agent_len, raw_dims):
    """ref dce.py:70-92: dce = inf; for t ...: if the agent has no state at t: break; d = round(distance, 3);
    if d < dce: keep (d, t); if dce == 0: break"""
    dce, t_dce = math.inf, 0
    for t in range(len(x)):
        if t >= agent_len:
            break
        cx = x[t] + VEH[2] * math.cos(theta[t])                      # convert_dynamic_obstacle.py:73
        cy = y[t] + VEH[2] * math.sin(theta[t])
        d = round(polygon_distance(_rect(cx, cy, theta[t], VEH[0], VEH[1]),
                                   _rect(agent_pos[t, 0], agent_pos[t, 1], agent_yaw[t], raw_dims[0], raw_dims[1])), 3)
        if d < dce:
            dce, t_dce = d, t
        if dce == 0.0:
            break
    return dce, t_dce


def test_oracle_dce_equals_the_python_restatement_on_random_batches(oracle):
    from frenetix_occlusion import synthetic as SY
    n_pairs = n_zero = 0
    for seed in (1, 2, 3):
        traj = SY.make_trajectories(12, 31, 0.1, seed=seed)
        agents = SY.make_agents(10, 31, 0.1, seed=seed + 10)
        agents["len"][::4] = np.array([7, 19, 30])[: len(agents["len"][::4])]    # some predictions end early (Q3)
        out = oracle.sweep(traj, agents, VEH, 0.1, metrics=("dce", "ttc", "ttce"), want_lists=False)
        for m in range(12):
            for k in range(10):
                d, t = dce_like_the_reference(traj["x"][m], traj["y"][m], traj["theta"][m], agents["pos"][k],
                                              agents["yaw"][k], int(agents["len"][k]), agents["raw_dims"][k])
                assert out["pair_f"][m, k, oracle.PF["dce"]] == d, (seed, m, k)
                assert out["pair_i"][m, k, oracle.PI["time_dce"]] == t, (seed, m, k)
                ttc = round(t * 0.1, 3) if np.isclose(d, 0.0) else math.inf            # ttc.py:43-46
                assert out["pair_f"][m, k, oracle.PF["ttc"]] == ttc
                assert out["pair_f"][m, k, oracle.PF["ttce"]] == round(t * 0.1, 3)   # ttce.py:39
                n_pairs += 1
                n_zero += d == 0.0
    assert n_pairs == 360 and n_zero > 5
