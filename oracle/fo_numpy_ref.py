"""NumPy restatement of the metric evaluation in the REFERENCE'S OWN LOOP STRUCTURE -- CPU baseline "B0" of SURVEY 8d.

TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/ import it; the product never does).

Shape of the computation, as in the reference: ``FOInterface.trajectory_safety_assessment`` is called once per candidate
trajectory (interface.py:216-219); ``Metric.evaluate_metrics`` walks the activated metrics in dependency order
(metrics/metric.py:35-48, 125-147); every metric walks the agent predictions (dce.py:38-48, cp.py:31-40, hr.py:57-99);
and the per-pair functions walk the timesteps one by one where the reference does (dce.py:69-88: two rectangles and
one distance per step; collision_probability.py:69-122: nine box integrals per near step) or use NumPy vectors over the
horizon where the reference does (harm_model.py:80-107; the LR4S angle bins are a Python loop over the samples,
logistic_regression.py:28-42).  One thread, plain float64 NumPy / math / scipy's mvnun.  Where the reference builds two
shapely polygons and asks GEOS for their distance, this file has a NumPy routine over the 16 edge pairs; where it
builds commonroad state objects per trajectory, this file indexes arrays -- the figure is indicative of the
reference's own speed (~1e3 pair-evals/s per core), not a measurement of it.

Results equal oracle/fo_oracle.c (tests/test_numpy_baseline.py), which is pinned to the reference's golden vectors.
"""
import math

import numpy as np

try:                                   # the reference calls scipy.stats.mvn.mvnun (collision_probability.py:117)
    from scipy.stats import _mvn as _mvn_mod
    _mvnun = _mvn_mod.mvnun
except Exception:                      # pragma: no cover - other scipy builds: closed form for diagonal covariances
    _mvnun = None

T_A = 45.0 / 180.0 * math.pi
T_B = 3.0 * T_A
PROTECTED = {0, 1, 2, 5, 6, 7, 9}      # car, truck, bus, priority, parked, train, taxi (harm_model.py:15-32)
UNPROTECTED = {3, 4, 8, 10}            # bicycle, pedestrian, motorcycle, unknown


def _rect(cx, cy, yaw, length, width):
    c, s = math.cos(yaw), math.sin(yaw)
    lx = np.array([-0.5, -0.5, 0.5, 0.5]) * length
    ly = np.array([-0.5, 0.5, 0.5, -0.5]) * width
    return np.stack((c * lx - s * ly + cx, s * lx + c * ly + cy), axis=1)


def _pts_segs(p, a, b):
    """distances of points p [n,2] to segments a->b [m,2] -> [n,m]"""
    ab = b - a
    l2 = (ab * ab).sum(1)
    ap = p[:, None, :] - a[None, :, :]
    r = np.clip((ap * ab[None]).sum(2) / l2[None], 0.0, 1.0)
    d = ap - r[..., None] * ab[None]
    return np.sqrt((d * d).sum(2))


def _inside(p, q):
    e = np.roll(q, -1, axis=0) - q
    cr = e[:, 0] * (p[1] - q[:, 1]) - e[:, 1] * (p[0] - q[:, 0])
    return not ((cr > 0).any() and (cr < 0).any())


def rect_distance(qa, qb):
    """distance of two convex quadrilaterals (what shapely's Polygon.distance returns, dce.py:76-79): containment,
    then the 16 edge pairs -- crossing edges give 0, otherwise the closest vertex-to-edge distance"""
    if _inside(qa[0], qb) or _inside(qb[0], qa):
        return 0.0
    a, b = qa, np.roll(qa, -1, axis=0)
    c, d = qb, np.roll(qb, -1, axis=0)
    ab, cd = (b - a)[:, None, :], (d - c)[None, :, :]
    ac = a[:, None, :] - c[None, :, :]
    den = ab[..., 0] * cd[..., 1] - ab[..., 1] * cd[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (ac[..., 1] * cd[..., 0] - ac[..., 0] * cd[..., 1]) / den
        s_ = (ac[..., 1] * ab[..., 0] - ac[..., 0] * ab[..., 1]) / den
    if ((den != 0.0) & (r >= 0.0) & (r <= 1.0) & (s_ >= 0.0) & (s_ <= 1.0)).any():
        return 0.0
    return float(min(_pts_segs(qa, c, d).min(), _pts_segs(qb, a, b).min()))


def _box_prob(lo, hi, mu, cov):
    if _mvnun is not None:
        return float(_mvnun(lo, hi, mu, cov)[0])
    sx, sy = math.sqrt(cov[0, 0]), math.sqrt(cov[1, 1])
    q = lambda z: 0.5 * math.erfc(z / math.sqrt(2.0))
    l0, u0, l1, u1 = (lo[0] - mu[0]) / sx, (hi[0] - mu[0]) / sx, (lo[1] - mu[1]) / sy, (hi[1] - mu[1]) / sy
    return q(l0) * q(l1) - q(u0) * q(l1) - q(l0) * q(u1) + q(u0) * q(u1)


class DCE:
    def __init__(self, veh):
        self.veh = veh

    def evaluate(self, results, traj, agents):
        out = {}
        for k in agents["active"]:                                  # dce.py:38-48
            out[k] = self._calc(traj, agents, k)
        return out

    def _calc(self, traj, agents, k):
        length, width, wb = self.veh[0], self.veh[1], self.veh[2]
        L = int(agents["len"][k])
        dce, time_dce = math.inf, 0
        for i in range(len(traj["x"])):                             # dce.py:69-88
            if i >= L:
                break
            th = traj["theta"][i]
            qa = _rect(traj["x"][i] + wb * math.cos(th), traj["y"][i] + wb * math.sin(th), th, length, width)
            qb = _rect(agents["pos"][k, i, 0], agents["pos"][k, i, 1], agents["yaw"][k, i], *agents["raw_dims"][k])
            d = float(np.round(rect_distance(qa, qb), 3))
            if d < dce:
                time_dce, dce = i, d
            if dce == 0.0:
                break
        return {"dce": dce, "time_dce": time_dce}


class TTC:
    def __init__(self, dt):
        self.dt = dt

    def evaluate(self, results, traj, agents):
        return {k: (float(np.round(r["time_dce"] * self.dt, 3)) if np.isclose(r["dce"], 0.0) else math.inf)
                for k, r in results["dce"].items()}                 # ttc.py:36-47


class TTCE(TTC):
    def evaluate(self, results, traj, agents):
        return {k: float(np.round(r["time_dce"] * self.dt, 3)) for k, r in results["dce"].items()}   # ttce.py:34-41


class WTTC:
    def evaluate(self, results, traj, agents):
        return min(results["ttc"].values(), default=math.inf)       # wttc.py:32-42


class CP:
    def __init__(self, veh):
        self.veh = veh

    def evaluate(self, results, traj, agents):
        return {k: self._calc(traj, agents, k) for k in agents["active"]}   # cp.py:31-40

    def _calc(self, traj, agents, k):
        length, width = self.veh[0], self.veh[1]
        off = np.array([length / 6.0, width / 2.0])
        L, T = int(agents["len"][k]), len(traj["x"])
        pos, yaw, cov = agents["pos"][k], agents["yaw"][k], agents["cov"][k]
        half = agents["shape"][k, 0] / 2.0
        probs = np.zeros(T - 1)
        for i in range(1, T):                                       # collision_probability.py:69-122
            if i >= L:
                continue
            dev = np.array([math.cos(yaw[i]), math.sin(yaw[i])]) * half
            means = [pos[i - 1], pos[i - 1] + dev, pos[i - 1] - dev]
            ego = np.array([traj["x"][i], traj["y"][i]])
            if min(float(np.linalg.norm(m - ego)) for m in means) > 5.0:
                continue
            c = np.array(cov[i - 1], dtype=np.float64).reshape(2, 2)
            if not c.any():
                c = np.eye(2) * 0.1
            ax = np.array([math.cos(traj["theta"][i]), math.sin(traj["theta"][i])]) * (length / 2.0) * (2.0 / 3.0)
            centres = [ego, ego + ax, ego - ax]
            p = 0.0
            for m in means:
                for cc in centres:
                    p += _box_prob(cc - off, cc + off, m, c)
            probs[i - 1] = p / 3.0
        return probs


def _lr4s(vel, ang, hc):
    ang = np.array(ang, dtype=np.float64)
    coef = np.empty_like(ang)
    for i in range(len(ang)):                                       # logistic_regression.py:28-42 (a Python loop there too)
        a = ang[i]
        if -T_A < a < T_A:
            coef[i] = 0.0
        elif (T_A <= a < T_B) or (-T_A >= a > -T_B):
            coef[i] = hc["lr4s_side"]
        else:
            coef[i] = hc["lr4s_rear"]
    return 1.0 / (1.0 + np.exp(-hc["lr4s_const"] - hc["lr4s_speed"] * vel - coef))


def _mass(typ, size):
    if typ in (0, 5, 6, 9):
        return -1333.5 + 526.9 * size ** 0.8
    return {1: 25000.0, 2: 13000.0, 3: 90.0, 4: 75.0, 7: 118800.0, 8: 250.0}.get(typ, 0.0)


class HR:
    def __init__(self, veh, hc):
        self.veh, self.hc = veh, hc

    def _harm(self, traj, agents, k):
        T, L = len(traj["x"]), int(agents["len"][k])
        n = min(T - 1, L)                                           # harm_model.py:65-66
        x, y, th, v = (np.asarray(traj[q][:n]) for q in ("x", "y", "theta", "v"))
        pos, yaw, av = agents["pos"][k, :n], agents["yaw"][k, :n], agents["v"][k, :n]
        typ = int(agents["type"][k])
        m_obs = _mass(typ, float(agents["shape"][k, 0] * agents["shape"][k, 1]))
        pdof = yaw - th + np.pi
        rel = np.arctan2(pos[:, 1] - y, pos[:, 0] - x)
        dv = np.sqrt(np.power(v, 2) + np.power(av, 2) + 2 * v * av * np.cos(pdof))
        ego_dv, obs_dv = m_obs / (self.veh[3] + m_obs) * dv, self.veh[3] / (self.veh[3] + m_obs) * dv
        hc = self.hc
        if typ in PROTECTED:
            return _lr4s(ego_dv, rel - th, hc), _lr4s(obs_dv, np.pi + rel - yaw, hc)
        if typ in UNPROTECTED:
            return (1.0 / (1.0 + np.exp(-hc["lr1s_const"] - hc["lr1s_speed"] * ego_dv)),
                    1.0 / (1.0 + np.exp(hc["ped_const"] - hc["ped_speed"] * obs_dv)))
        return np.ones(n), np.ones(n)

    def evaluate(self, results, traj, agents):
        out = {}
        for k in agents["active"]:                                  # hr.py:57-99
            eh, oh = self._harm(traj, agents, k)
            if len(eh) == 0:
                continue
            cp = results["cp"][k]
            er = [eh[t] * cp[t] for t in range(len(eh))]            # hr.py:78-79: list comprehensions
            orr = [oh[t] * cp[t] for t in range(len(oh))]
            mcp = float(np.max(cp))
            out[k] = {"max_ego_risk": max(er), "max_obst_risk": max(orr), "max_obst_risk_index": int(np.argmax(orr)),
                      "max_obst_harm_with_cp": float(oh[int(np.argmax(cp))]) if mcp > 0.01 else 0.0,
                      "max_ego_harm": float(np.max(eh)), "max_obst_harm": float(np.max(oh)),
                      "ego_risk_traj": er, "obst_risk_traj": orr, "ego_harm_traj": eh, "obst_harm_traj": oh,
                      "collision_probability": cp, "max_collision_probability": mcp}
        for key, src in (("max_ego_risk_all", "max_ego_risk"), ("max_obst_risk_all", "max_obst_risk"),
                         ("max_ego_harm_all", "max_ego_harm"), ("max_obst_harm_all", "max_obst_harm"),
                         ("max_collision_probability_all", "max_collision_probability"),
                         ("max_obst_harm_with_cp_all", "max_obst_harm_with_cp")):
            out[key] = max([0.0] + [r[src] for r in out.values() if isinstance(r, dict)])   # hr.py:101-114
        return out


class Metric:
    """metric.py:18-100: ordered metrics, thresholds"""

    def __init__(self, vehicle, dt, metrics=("hr", "ttc", "ttce", "dce", "wttc", "cp"), thresholds=None, harm_coeff=None):
        from . import fo_oracle as O
        names = list(metrics)                                       # metric.py:125-147
        if "wttc" in names:
            names = ["ttc"] + [n for n in names if n != "ttc"]
        if any(n in names for n in ("ttc", "ttce")):
            names = ["dce"] + [n for n in names if n != "dce"]
        if "hr" in names:
            names = ["cp"] + [n for n in names if n != "cp"]
        hc = harm_coeff or O.HARM_COEFF
        table = {"dce": lambda: DCE(vehicle), "ttc": lambda: TTC(dt), "ttce": lambda: TTCE(dt), "wttc": WTTC,
                 "cp": lambda: CP(vehicle), "hr": lambda: HR(vehicle, hc)}
        self.metrics = [(n, table[n]()) for n in names]
        self.thr = thresholds or {}

    def evaluate_metrics(self, traj, agents):
        results = {}
        if not len(agents["active"]):
            return results, True
        for name, metric in self.metrics:                           # metric.py:46-48
            results[name] = metric.evaluate(results, traj, agents)
        ok, thr = True, self.thr
        if "hr" in results:
            if thr.get("harm") is not None and results["hr"]["max_obst_harm_with_cp_all"] > thr["harm"]:
                ok = False
            if thr.get("risk") is not None and results["hr"]["max_obst_risk_all"] > thr["risk"]:
                ok = False
            if thr.get("cp") is not None and results["hr"]["max_collision_probability_all"] > thr["cp"]:
                ok = False
        if "ttc" in results and thr.get("ttc") is not None and min(results["ttc"].values(), default=math.inf) < thr["ttc"]:
            ok = False
        if "dce" in results and thr.get("dce") is not None and any(r["dce"] < thr["dce"] for r in results["dce"].values()):
            ok = False
        return results, ok


def sweep(traj, agents, vehicle, dt, metrics=("hr", "ttc", "ttce", "dce", "wttc", "cp"), thr=None, harm_coeff=None):
    """per trajectory -> per metric -> per agent -> per timestep; returns (list of result dicts, safe [M])"""
    ag = dict(agents)
    ag["cov"] = np.asarray(agents["cov"]).reshape(len(agents["len"]), -1, 4)
    ag["active"] = [k for k in range(len(agents["len"])) if agents["len"][k] > 0]
    metric = Metric(vehicle, dt, metrics, thr, harm_coeff)
    M = len(traj["x"])
    results, safe = [], np.zeros(M, dtype=np.uint8)
    for m in range(M):                                              # interface.py:216-219, once per candidate
        t = {k: np.asarray(v[m]) for k, v in traj.items()}
        r, ok = metric.evaluate_metrics(t, ag)
        results.append(r)
        safe[m] = ok
    return results, safe
