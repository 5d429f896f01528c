#!/usr/bin/env python3
"""Golden-vector generator: runs the reference's OWN metric code and records inputs + outputs.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What is executed unmodified from /root/reference/frenetix_occlusion:
  metrics/cp.py  (CP.evaluate -> metrics/utils/collision_probability.py:14-126)
  metrics/hr.py  (HR.evaluate -> metrics/utils/harm_model.py:35-107, logistic_regression.py:11-75)
  metrics/ttc.py:28-49, metrics/ttce.py:28-43, metrics/wttc.py:29-44

Three third-party names those files import are absent from this image and are aliased before import
(SURVEY.md §8c).  They contain no algorithm of the hot path:
  * commonroad.scenario.obstacle.ObstacleType  -> an Enum of the type strings (harm_model.py:15-32 keys)
  * commonroad_dc.pycrcc.RectOBB               -> value holder with center()/r_x()/local_x_axis()
                                                  (collision_probability.py:149-156 uses only these)
  * scipy.stats.mvn                            -> scipy.stats._mvn (the same Fortran MVNDST wrapper; the public
                                                  alias was dropped in scipy 1.15)
SensorModel and SpawnLocator need shapely/commonroad proper and are NOT run here; for those the oracle is pinned by
analytic known-answer tests only ("parity unpinned").  DCE and BE (round 6; `gen_golden.py dce`, `gen_golden.py be`,
each in a process of its own): metrics/dce.py and metrics/be.py ARE executed unmodified -- the walk over the time steps with its
rounding, tie and stop rules; the deceleration profiles, scipy re-sampling and bisection -- with the ONE module they import
besides numpy / scipy, metrics/utils/convert_dynamic_obstacle.py (commonroad proper), replaced by duck-typed obstacles over this
repository's rectangle primitives (_install_rectangle_stub): the polygon `distance` / `intersects` under those loops is the
oracle's restatement of GEOS' (pinned to exact arithmetic by tests/test_dce_sympy.py), everything above it is the reference's.

Only arrays (inputs and the reference's outputs) are written, to tests/golden/*.npz.  No reference source
or bytecode is copied.
"""
import enum
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


# --------------------------------------------------------------------------- aliases for absent third-party names
def _install_aliases():
    class ObstacleType(enum.Enum):
        UNKNOWN = "unknown"
        CAR = "car"
        TRUCK = "truck"
        BUS = "bus"
        BICYCLE = "bicycle"
        PEDESTRIAN = "pedestrian"
        PRIORITY_VEHICLE = "priorityVehicle"
        PARKED_VEHICLE = "parkedVehicle"
        CONSTRUCTION_ZONE = "constructionZone"
        TRAIN = "train"
        ROAD_BOUNDARY = "roadBoundary"
        MOTORCYCLE = "motorcycle"
        TAXI = "taxi"
        BUILDING = "building"
        PILLAR = "pillar"
        MEDIAN_STRIP = "median_strip"

    cr = types.ModuleType("commonroad")
    cr_s = types.ModuleType("commonroad.scenario")
    cr_o = types.ModuleType("commonroad.scenario.obstacle")
    cr_o.ObstacleType = ObstacleType
    cr.scenario = cr_s
    cr_s.obstacle = cr_o
    sys.modules.update({"commonroad": cr, "commonroad.scenario": cr_s, "commonroad.scenario.obstacle": cr_o})

    class RectOBB:
        def __init__(self, r_x, r_y, yaw, x, y):
            self._rx, self._yaw, self._c = r_x, yaw, np.array([x, y], dtype=np.float64)

        def center(self):
            return self._c

        def r_x(self):
            return self._rx

        def local_x_axis(self):
            return np.array([np.cos(self._yaw), np.sin(self._yaw)])

    dc = types.ModuleType("commonroad_dc")
    pc = types.ModuleType("commonroad_dc.pycrcc")
    pc.RectOBB = RectOBB
    dc.pycrcc = pc
    sys.modules.update({"commonroad_dc": dc, "commonroad_dc.pycrcc": pc})

    import scipy.stats
    import scipy.stats._mvn as _mvn
    scipy.stats.mvn = _mvn
    sys.modules["scipy.stats.mvn"] = _mvn


# --------------------------------------------------------------------------- harness objects (duck-typed inputs)
class Cartesian:
    def __init__(self, x, y, theta, v, a):
        self.x, self.y, self.theta, self.v, self.a = (np.array(q, dtype=np.float64) for q in (x, y, theta, v, a))


class Traj:
    def __init__(self, *args):
        self.cartesian = Cartesian(*args)


class VehicleParams:
    # CommonRoad vehicle 2 (BMW 320i) as quoted in SURVEY.md §8
    length, width, wb_rear_axle, mass, a_max = 4.508, 1.610, 1.4227, 1093.3, 11.5


class Agent:
    def __init__(self, agent_id, agent_type):
        self.agent_id, self.agent_type = agent_id, agent_type


class AgentManager:
    def __init__(self, dt):
        self.dt, self.phantom_agents, self.predictions = dt, [], {}

    def agent_by_prediction_id(self, pid):
        aid = int(str(pid)[:5])
        for a in self.phantom_agents:
            if a.agent_id == aid:
                return a


AGENT_TYPES = ["Car", "Truck", "Bicycle", "Pedestrian"]
RAW_DIMS = {"Car": (4.8, 2.0), "Truck": (9.0, 2.5), "Bicycle": (2.0, 0.9), "Pedestrian": (0.3, 0.5),
            # types the phantom generator never produces but the harm model's tables know (harm_model.py:15-32,158-190)
            "Bus": (12.0, 2.6), "Train": (20.0, 3.0), "Motorcycle": (2.2, 0.8), "Taxi": (4.6, 1.9), "Unknown": (1.0, 1.0)}
SPEED = {"Car": 10.0, "Truck": 10.0, "Bicycle": 5.0, "Pedestrian": 1.4, "Bus": 8.0, "Train": 12.0, "Motorcycle": 11.0,
         "Taxi": 9.0, "Unknown": 2.0}


def make_traj(rng, T, dt, x0=0.0, y0=0.0, psi0=0.0):
    """Quintic-blend lateral offset + smoothly changing speed, in a frame rotated by psi0."""
    v0 = rng.uniform(3, 12)
    v1 = v0 * rng.uniform(0.3, 1.3)
    d1 = rng.uniform(-3, 3)
    tau = np.linspace(0, 1, T)
    blend = 10 * tau ** 3 - 15 * tau ** 4 + 6 * tau ** 5
    v = v0 + (v1 - v0) * blend
    s = np.concatenate(([0.0], np.cumsum(0.5 * (v[1:] + v[:-1]) * dt)))
    d = d1 * blend
    c, sn = np.cos(psi0), np.sin(psi0)
    x = x0 + c * s - sn * d
    y = y0 + sn * s + c * d
    theta = np.arctan2(np.gradient(y), np.gradient(x))
    a = np.gradient(v, dt)
    return x, y, theta, v, a


def make_prediction(rng, kind, L, dt, near_xy, var0=0.1, vf=1.05, zero_cov=False, curved=False, heading=None,
                    offset=None):
    raw_l, raw_w = RAW_DIMS[kind]
    fl, fw = (1.4, 2.5) if kind == "Bicycle" else (1.2, 1.3)
    speed = SPEED[kind] * rng.uniform(0.6, 1.2)
    psi = rng.uniform(-np.pi, np.pi) if heading is None else heading
    p0 = np.asarray(near_xy) + (rng.uniform(-6, 6, size=2) if offset is None else np.asarray(offset))
    t = np.arange(L) * dt
    if curved:
        om = rng.uniform(-0.4, 0.4)
        yaw = psi + om * t
        pos = p0 + np.cumsum(np.stack((np.cos(yaw), np.sin(yaw)), -1) * speed * dt, axis=0)
    else:
        yaw = np.full(L, psi)
        vx, vy = round(speed * np.cos(psi), 3), round(speed * np.sin(psi), 3)
        pos = p0 + t[:, None] * np.array([vx, vy])
    var = var0 * vf ** np.arange(L)
    cov = np.array([[[q, 0.0], [0.0, q]] for q in var])
    if zero_cov:
        cov[: L // 2] = 0.0
    return {"pos_list": pos, "v_list": np.full(L, speed), "orientation_list": yaw,
            "cov_list": cov, "shape": {"length": raw_l * fl, "width": raw_w * fw}}


def pad(arr_list, n, fill=np.nan):
    out = np.full((len(arr_list), n), fill)
    for i, a in enumerate(arr_list):
        out[i, : len(a)] = a
    return out


def run_case(name, trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC, dce_inputs=None):
    am = AgentManager(dt)
    keys = []
    for i, (kind, pred) in enumerate(zip(kinds, preds)):
        aid = 10000 + i
        am.phantom_agents.append(Agent(aid, kind))
        key = int(str(aid) + "0")
        am.predictions[key] = pred
        keys.append(key)
    vp = VehicleParams()
    cp_m, hr_m = CP(vp, am), HR(vp, am)
    M, A = len(trajs), len(keys)
    T = len(trajs[0].cartesian.x)
    Lmax = max(len(p["pos_list"]) for p in preds)
    out = {
        "dt": dt, "vehicle": np.array([vp.length, vp.width, vp.wb_rear_axle, vp.mass, vp.a_max]),
        "traj_x": np.stack([t.cartesian.x for t in trajs]), "traj_y": np.stack([t.cartesian.y for t in trajs]),
        "traj_theta": np.stack([t.cartesian.theta for t in trajs]), "traj_v": np.stack([t.cartesian.v for t in trajs]),
        "traj_a": np.stack([t.cartesian.a for t in trajs]),
        "agent_type": np.array(kinds), "agent_len": np.array([len(p["pos_list"]) for p in preds]),
        "agent_pos": np.stack([np.pad(p["pos_list"], ((0, Lmax - len(p["pos_list"])), (0, 0))) for p in preds]),
        "agent_yaw": pad([p["orientation_list"] for p in preds], Lmax, 0.0),
        "agent_v": pad([p["v_list"] for p in preds], Lmax, 0.0),
        "agent_cov": np.stack([np.pad(p["cov_list"], ((0, Lmax - len(p["cov_list"])), (0, 0), (0, 0))) for p in preds]),
        "agent_shape": np.array([[p["shape"]["length"], p["shape"]["width"]] for p in preds]),
        "agent_raw_dims": np.array([RAW_DIMS[k] for k in kinds]),
    }
    cp = np.zeros((M, A, T - 1))
    Lh = [min(T - 1, len(p["pos_list"])) for p in preds]
    lists = {k: np.full((M, A, T - 1), np.nan) for k in ("ego_harm", "obst_harm", "ego_risk", "obst_risk")}
    scal = {k: np.zeros((M, A)) for k in ("max_ego_risk", "max_obst_risk", "max_obst_harm_with_cp", "max_ego_harm",
                                          "max_obst_harm", "max_collision_probability")}
    ridx = np.zeros((M, A), dtype=np.int64)
    glob = {k: np.zeros(M) for k in ("max_ego_risk_all", "max_obst_risk_all", "max_ego_harm_all", "max_obst_harm_all",
                                     "max_collision_probability_all", "max_obst_harm_with_cp_all")}
    for m, tr in enumerate(trajs):
        res = {"cp": cp_m.evaluate(tr, {})}
        res["hr"] = hr_m.evaluate(tr, res)
        for j, key in enumerate(keys):
            cp[m, j] = res["cp"][key]
            h = res["hr"][key]
            lists["ego_harm"][m, j, : Lh[j]] = h["ego_harm_traj"]
            lists["obst_harm"][m, j, : Lh[j]] = h["obst_harm_traj"]
            lists["ego_risk"][m, j, : Lh[j]] = h["ego_risk_traj"]
            lists["obst_risk"][m, j, : Lh[j]] = h["obst_risk_traj"]
            for k in scal:
                scal[k][m, j] = h[k]
            ridx[m, j] = h["max_obst_risk_index"]
        for k in glob:
            glob[k][m] = res["hr"][k]
    out["ref_cp"] = cp
    out.update({"ref_" + k: v for k, v in lists.items()})
    out.update({"ref_" + k: v for k, v in scal.items()})
    out["ref_max_obst_risk_index"] = ridx
    out.update({"ref_" + k: v for k, v in glob.items()})

    # TTC / TTCE / WTTC are post-processing of a DCE result dict: feed chosen (dce, time_dce) pairs.
    if dce_inputs is not None:
        dce_v, dce_t = dce_inputs  # [M, A]
        ttc = np.zeros((M, A))
        ttce = np.zeros((M, A))
        wttc = np.zeros(M)
        for m in range(M):
            res = {"dce": {key: {"dce": float(dce_v[m, j]), "time_dce": int(dce_t[m, j])} for j, key in enumerate(keys)}}
            res["ttc"] = TTC(am).evaluate(None, res)
            res["ttce"] = TTCE(am).evaluate(None, res)
            w = WTTC.evaluate(None, res)
            for j, key in enumerate(keys):
                ttc[m, j] = res["ttc"][key]
                ttce[m, j] = res["ttce"][key]
            wttc[m] = w
        out.update({"in_dce": dce_v, "in_time_dce": dce_t, "ref_ttc": ttc, "ref_ttce": ttce, "ref_wttc": wttc})
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "M", M, "A", A, "T", T, "max cp", cp.max())


def _install_agent_stubs():
    """enough of commonroad / shapely / the reference's own frenetix- and route-planner wrappers for agent.py to import;
    none of it computes anything the pedestrian prediction uses when the spawn point brings its orientation"""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class Rectangle:
        def __init__(self, length, width, center=None, orientation=0.0):
            self.length, self.width, self.center, self.orientation = length, width, center, orientation

    class State:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class Dummy:
        def __init__(self, *a, **kw):
            pass

    mod("commonroad.geometry", shape=mod("commonroad.geometry.shape", Rectangle=Rectangle))
    mod("commonroad.scenario.state", InitialState=State, CustomState=State)
    sys.modules["commonroad.scenario.obstacle"].DynamicObstacle = Dummy
    mod("commonroad.prediction", prediction=mod("commonroad.prediction.prediction", TrajectoryPrediction=Dummy))
    mod("commonroad.scenario.trajectory", Trajectory=Dummy)
    mod("shapely", geometry=mod("shapely.geometry", Polygon=Dummy, Point=Dummy, MultiPolygon=Dummy, LineString=Dummy),
        affinity=mod("shapely.affinity", rotate=Dummy, translate=Dummy))
    mod("frenetix_occlusion.route_planner", FORoutePlanner=Dummy)
    mod("frenetix_occlusion.utils.frenetix_handler", FrenetixHandler=Dummy)


def run_pedestrian_predictions(name, dt):
    """predictions of the reference's own OAPPedestrianAgent (agent.py:429-536) for spawn points that bring their
    orientation (interface.py:192-198): rounding of the velocity components (Q12), horizon -> sample count, covariance
    growth (agent.py:260-280), shape inflation"""
    _install_agent_stubs()
    from frenetix_occlusion.agent import OAPPedestrianAgent
    rng = np.random.default_rng(20240138)
    n = 24
    cfg = {"prediction": {"size_factor_length_s": 1.2, "size_factor_width_s": 1.3, "variance_factor": 1.05}}
    pos0 = rng.uniform(-40.0, 40.0, size=(n, 2))
    yaw = np.concatenate(([0.0, np.pi / 2, np.pi, -np.pi / 2, np.pi / 3, 2.0943951023931953], rng.uniform(-np.pi, 2 * np.pi, n - 6)))
    speed = np.concatenate(([1.4, 1.4, 1.4, 1.4, 1.001, 0.9995], rng.uniform(0.3, 3.0, n - 6)))
    horizon = np.where(np.arange(n) % 5 == 4, 2.05, 3.0)
    out = {"dt": dt, "pos0": pos0, "yaw": yaw, "speed": speed, "horizon": horizon, "raw_length": 0.3, "raw_width": 0.5,
           **{"cfg_" + k: v for k, v in cfg["prediction"].items()}}
    Tmax = int(3.0 / dt) + 1
    P, V, Y, Cv = (np.full((n, Tmax) + sh, np.nan) for sh in ((2,), (), (), (2, 2)))
    L, shape = np.zeros(n, dtype=np.int64), np.zeros((n, 2))
    for i in range(n):
        a = OAPPedestrianAgent(pos0[i].copy(), float(speed[i]), "Pedestrian", {"agent_id": 10000 + i, "length": 0.3, "width": 0.5},
                               None, cfg, dt=dt, horizon=float(horizon[i]), ref_path=None, mode="ref_path", orientation=float(yaw[i]))
        p = a.predictions[0]
        L[i] = len(p["pos_list"])
        P[i, :L[i]], V[i, :L[i]], Y[i, :L[i]], Cv[i, :L[i]] = p["pos_list"], p["v_list"], p["orientation_list"], p["cov_list"]
        shape[i] = (p["shape"]["length"], p["shape"]["width"])
    out.update({"ref_pos": P, "ref_v": V, "ref_yaw": Y, "ref_cov": Cv, "ref_len": L, "ref_shape": shape})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "n", n, "lengths", sorted(set(L.tolist())))


def run_obstacle_states(name):
    """FOObstacles of the reference (utils/fo_obstacle.py:14-116 + helper_functions.calc_corner_points :99-112, imported
    unmodified; shapely's Point / Polygon are inert stubs, they only wrap what this fixture records): which state an
    obstacle has at a global time step -- initial state at its own first step, state_list[rel - 1] afterwards, nothing
    before it appears and after its trajectory ends, static obstacles always their initial state -- and its corner
    points.  The obstacle shape's vertices follow commonroad's Rectangle (the closed ring (-l/2,-w/2), (-l/2,w/2),
    (l/2,w/2), (l/2,-w/2), (-l/2,-w/2)): an assumption of this fixture, commonroad is not installable here."""
    _install_agent_stubs()
    mp = types.ModuleType("shapely.geometry.multipolygon")
    mp.MultiPolygon = lambda polys: list(polys)
    sys.modules["shapely.geometry.multipolygon"] = mp
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    rng = np.random.default_rng(20240139)
    NS = types.SimpleNamespace

    def ring(l, w):
        return np.array([[-l / 2, -w / 2], [-l / 2, w / 2], [l / 2, w / 2], [l / 2, -w / 2], [-l / 2, -w / 2]])

    spec = [("STATIC", 0, 0), ("DYNAMIC", 0, 10), ("DYNAMIC", 3, 6), ("STATIC", 2, 0), ("DYNAMIC", 5, 20), ("DYNAMIC", 1, 1)]
    obs, rec = [], []
    for i, (role, t0, n) in enumerate(spec):
        l, w = float(rng.uniform(1.5, 9.0)), float(rng.uniform(0.6, 2.6))
        init = np.array([rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(-4, 7), rng.uniform(0, 12)])
        states = np.column_stack((rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(-4, 7, n), rng.uniform(0, 12, n)))
        o = NS(obstacle_id=100 + i, obstacle_role=NS(name=role), obstacle_shape=NS(vertices=ring(l, w), length=l, width=w),
               initial_state=NS(time_step=t0, position=init[:2].copy(), orientation=float(init[2])),
               prediction=NS(trajectory=NS(state_list=[NS(position=st[:2].copy(), orientation=float(st[2])) for st in states])))
        obs.append(o)
        rec.append((role, t0, l, w, init, states))
    fo = FOObstacles(obs)
    steps = 16
    present = np.zeros((len(obs), steps), dtype=np.uint8)
    pos, yaw, corn = np.full((len(obs), steps, 2), np.nan), np.full((len(obs), steps), np.nan), np.full((len(obs), steps, 4, 2), np.nan)
    for t in range(steps):
        fo.update(t)
        for i, o in enumerate(fo):
            if o.current_pos is not None:
                present[i, t] = 1
                pos[i, t], yaw[i, t], corn[i, t] = o.current_pos, o.current_orientation, o.current_corner_points
    nmax = max(len(r[5]) for r in rec)
    st = np.full((len(obs), nmax, 4), np.nan)
    for i, r in enumerate(rec):
        st[i, :len(r[5])] = r[5]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), role=np.array([r[0] for r in rec]), t0=np.array([r[1] for r in rec]),
                        length=np.array([r[2] for r in rec]), width=np.array([r[3] for r in rec]),
                        initial=np.stack([r[4] for r in rec]), states=st, n_states=np.array([len(r[5]) for r in rec]),
                        ref_present=present, ref_pos=pos, ref_yaw=yaw, ref_corners=corn)
    print("wrote", name, "present per obstacle", present.sum(axis=1).tolist())


def run_route_enumeration(name):
    """FORoutePlanner._find_all_routes of the reference (route_planner.py:54-90, imported unmodified; the three
    commonroad_route_planner imports at its top are inert stubs): candidate lanelet-id routes from every start lanelet,
    depth 2, over successors and same-direction neighbours that have successors -- on the topology of the three example
    scenarios (read from their XML by this repository's loader; only ids and adjacency are stored) and on two random
    networks."""
    for modname, names in (("commonroad_route_planner", ()), ("commonroad_route_planner.route_planner", ("Route",)),
                           ("commonroad_route_planner.utility", ()), ("commonroad_route_planner.utility.route", ("lanelet_orientation_at_position",)),
                           ("commonroad_route_planner.route", ("RouteType",))):
        m = types.ModuleType(modname)
        for n in names:
            setattr(m, n, object)
        sys.modules[modname] = m
    sys.modules.pop("frenetix_occlusion.route_planner", None)      # (a stub of it may be installed by the agent case)
    from frenetix_occlusion.route_planner import FORoutePlanner
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(OUT)), "frenetix-occlusion_amd"))
    nets = []
    # the example scenarios' topology through this repository's loader (its own package shadows the reference's name:
    # load it from its file)
    import importlib.util
    spec = importlib.util.spec_from_file_location("fo_amd_scenario", os.path.join(os.path.dirname(os.path.dirname(OUT)),
                                                  "frenetix-occlusion_amd", "frenetix_occlusion", "scenario.py"))
    SC = importlib.util.module_from_spec(spec)
    sys.modules["fo_amd_scenario"] = SC
    spec.loader.exec_module(SC)
    for k in (1, 2, 3):
        sc = SC.load_commonroad_xml(os.path.join(REF, "example_scenarios", f"scenario{k}.xml"))
        nets.append([(ll.lanelet_id, list(ll.successors), ll.adj_left, bool(ll.adj_left_same_direction), ll.adj_right,
                      bool(ll.adj_right_same_direction)) for ll in sc.lanelets])
    rng = np.random.default_rng(20240140)
    for n in (30, 45):
        ids = list(range(100, 100 + n))
        net = []
        for i in ids:
            succ = [int(x) for x in rng.choice(ids, size=int(rng.integers(0, 4)), replace=False) if x != i]
            al = int(rng.choice(ids)) if rng.random() < 0.4 else None
            ar = int(rng.choice(ids)) if rng.random() < 0.4 else None
            net.append((i, succ, al, bool(rng.random() < 0.6), ar, bool(rng.random() < 0.6)))
        nets.append(net)
    NS = types.SimpleNamespace
    out = {"n_nets": len(nets)}
    for q, net in enumerate(nets):
        by = {i: NS(lanelet_id=i, successor=list(su), adj_left=al, adj_left_same_direction=als, adj_right=ar,
                    adj_right_same_direction=ars) for i, su, al, als, ar, ars in net}
        rp = FORoutePlanner(None, NS(find_lanelet_by_id=lambda i, by=by: by[i]), None, False)
        flat, off = [], [0]
        starts = []
        for i in by:
            for route in rp._find_all_routes(i, max_depth=2):
                flat.extend(route)
                off.append(len(flat))
                starts.append(i)
        succ_flat, succ_off = [], [0]
        for _, su, *_ in net:
            succ_flat.extend(su)
            succ_off.append(len(succ_flat))
        out.update({f"net{q}_id": np.array([t[0] for t in net]), f"net{q}_succ": np.array(succ_flat, dtype=np.int64),
                    f"net{q}_succ_off": np.array(succ_off), f"net{q}_adj_left": np.array([-1 if t[2] is None else t[2] for t in net]),
                    f"net{q}_adj_left_same": np.array([t[3] for t in net]),
                    f"net{q}_adj_right": np.array([-1 if t[4] is None else t[4] for t in net]),
                    f"net{q}_adj_right_same": np.array([t[5] for t in net]),
                    f"net{q}_route_start": np.array(starts), f"net{q}_route_flat": np.array(flat, dtype=np.int64),
                    f"net{q}_route_off": np.array(off)})
        print("net", q, "lanelets", len(net), "routes", len(starts))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


class _OracleDCE:
    """stands in for metrics/dce.py (shapely + commonroad_dc are not installable here): the reference's Metric class gets
    its 'dce' results from this repository's oracle, so that its OWN threshold logic (metric.py:50-100) and dependency
    closure (:125-147) can be run unmodified on top of its own CP / HR / TTC / TTCE / WTTC results."""
    table = {}          # id(trajectory) -> {prediction key: {"dce": float, "time_dce": int}}

    def __init__(self, vehicle_params, agent_manager):
        pass

    def evaluate(self, trajectory, results):
        return _OracleDCE.table[id(trajectory)]


class _NoBE:
    def __init__(self, vehicle_params, agent_manager):
        pass


def run_threshold_case(name, trajs, kinds, preds, dt, Metric, configs):
    """(results, safe) of the reference's Metric.evaluate_metrics for several metric / threshold configurations"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import fo_oracle as O
    O.build()
    am = AgentManager(dt)
    keys = []
    for i, (kind, pred) in enumerate(zip(kinds, preds)):
        am.phantom_agents.append(Agent(10000 + i, kind))
        key = int(str(10000 + i) + "0")
        am.predictions[key] = pred
        keys.append(key)
    vp = VehicleParams()
    M, A, T = len(trajs), len(keys), len(trajs[0].cartesian.x)
    Lmax = max(len(p["pos_list"]) for p in preds)
    codes = {"car": 0, "truck": 1, "bus": 2, "bicycle": 3, "pedestrian": 4}
    traj = {k: np.stack([getattr(t.cartesian, k) for t in trajs]) for k in ("x", "y", "theta", "v", "a")}
    agents = {
        "pos": np.stack([np.pad(p["pos_list"], ((0, Lmax - len(p["pos_list"])), (0, 0))) for p in preds]),
        "yaw": pad([p["orientation_list"] for p in preds], Lmax, 0.0), "v": pad([p["v_list"] for p in preds], Lmax, 0.0),
        "cov": np.stack([np.pad(p["cov_list"], ((0, Lmax - len(p["cov_list"])), (0, 0), (0, 0))) for p in preds]),
        "shape": np.array([[p["shape"]["length"], p["shape"]["width"]] for p in preds]),
        "raw_dims": np.array([RAW_DIMS[k] for k in kinds]), "type": np.array([codes[k.lower()] for k in kinds], dtype=np.int32),
        "len": np.array([len(p["pos_list"]) for p in preds], dtype=np.int32)}
    orc = O.sweep(traj, agents, (vp.length, vp.width, vp.wb_rear_axle, vp.mass, vp.a_max), dt)
    for m, tr in enumerate(trajs):
        _OracleDCE.table[id(tr)] = {key: {"dce": float(orc["pair_f"][m, j, O.PF["dce"]]),
                                           "time_dce": int(orc["pair_i"][m, j, O.PI["time_dce"]])} for j, key in enumerate(keys)}
    out = {"dt": dt, "vehicle": np.array([vp.length, vp.width, vp.wb_rear_axle, vp.mass, vp.a_max]),
           **{"traj_" + k: v for k, v in traj.items()},
           "agent_type": np.array(kinds), "agent_len": agents["len"], "agent_pos": agents["pos"], "agent_yaw": agents["yaw"],
           "agent_v": agents["v"], "agent_cov": agents["cov"], "agent_shape": agents["shape"], "agent_raw_dims": agents["raw_dims"],
           "n_configs": len(configs)}
    names = ("harm", "risk", "be", "cp", "ttc", "wttc", "ttce", "dce")
    # a threshold given as ("q", f) is put at the f-quantile of the batch's own values (finite ones), so that the
    # configuration has candidates on both sides of it
    per_traj = {"harm": orc["cost"][:, O.COST["max_obst_harm_with_cp_all"]], "risk": orc["cost"][:, O.COST["max_obst_risk_all"]],
                "cp": orc["cost"][:, O.COST["max_collision_probability_all"]], "ttc": orc["cost"][:, O.COST["wttc"]],
                "dce": orc["cost"][:, O.COST["min_dce"]]}
    for c, (activated, thr) in enumerate(configs):
        thr = dict(thr)
        for k, v in list(thr.items()):
            if isinstance(v, tuple):
                vals = per_traj[k][np.isfinite(per_traj[k])]
                thr[k] = float(np.round(np.quantile(vals, v[1]), 4))
        full = {k: thr.get(k) for k in names}
        met = Metric({"activated_metrics": list(activated), "metric_thresholds": full}, vp, am)
        safe = np.zeros(M, dtype=np.uint8)
        present = set()
        for m, tr in enumerate(trajs):
            res, ok = met.evaluate_metrics(tr)
            safe[m] = 1 if ok else 0
            present |= set(res)
        out[f"cfg{c}_activated"] = np.array(list(activated))
        out[f"cfg{c}_thr"] = np.array([np.nan if full[k] is None else float(full[k]) for k in names])
        out[f"cfg{c}_evaluated"] = np.array(sorted(present))         # metric names after the dependency closure
        out[f"cfg{c}_safe"] = safe
    out["thr_names"] = np.array(names)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "M", M, "A", A, "configs", len(configs), "safe fractions",
          [float(out[f"cfg{c}_safe"].mean()) for c in range(len(configs))])


def main():
    _install_aliases()
    sys.path.insert(0, REF)
    for modname, cls_name, cls in (("frenetix_occlusion.metrics.dce", "DCE", _OracleDCE), ("frenetix_occlusion.metrics.be", "BE", _NoBE)):
        mod = types.ModuleType(modname)
        setattr(mod, cls_name, cls)
        sys.modules[modname] = mod
    from frenetix_occlusion.metrics.cp import CP
    from frenetix_occlusion.metrics.hr import HR
    from frenetix_occlusion.metrics.ttc import TTC
    from frenetix_occlusion.metrics.ttce import TTCE
    from frenetix_occlusion.metrics.wttc import WTTC
    dt = 0.1

    # case 0: the SURVEY §8c probe (straight 8 m/s ego, pedestrian crossing at x = 15)
    T = 31
    t = np.arange(T) * dt
    tr = Traj(8.0 * t, np.zeros(T), np.zeros(T), np.full(T, 8.0), np.zeros(T))
    psi = np.pi / 2
    pos = np.array([15.0, -2.0]) + t[:, None] * np.array([round(1.4 * np.cos(psi), 3), round(1.4 * np.sin(psi), 3)])
    var = 0.1 * 1.05 ** np.arange(T)
    ped = {"pos_list": pos, "v_list": np.full(T, 1.4), "orientation_list": np.full(T, psi),
           "cov_list": np.array([[[q, 0.0], [0.0, q]] for q in var]), "shape": {"length": 0.3 * 1.2, "width": 0.5 * 1.3}}
    run_case("probe_ped_crossing", [tr], ["Pedestrian"], [ped], dt, CP, HR, TTC, TTCE, WTTC,
             dce_inputs=(np.array([[0.0]]), np.array([[19]])))

    # case 1: random batch, all four phantom types, equal lengths
    rng = np.random.default_rng(20240131)
    trajs = [Traj(*make_traj(rng, 31, dt, psi0=0.3)) for _ in range(24)]
    mid = np.array([trajs[0].cartesian.x[15], trajs[0].cartesian.y[15]])
    kinds = [AGENT_TYPES[i % 4] for i in range(8)]
    preds = [make_prediction(rng, k, 31, dt, mid, curved=(i % 3 == 0)) for i, k in enumerate(kinds)]
    dv = rng.choice([0.0, 0.0, 1e-9, 0.001, 0.004, 0.25, 3.217, 17.5], size=(24, 8))
    dtm = rng.integers(0, 31, size=(24, 8))
    run_case("random_equal_len", trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC, dce_inputs=(dv, dtm))

    # case 2: ragged predictions (shorter and longer than the trajectory), zero covariance blocks, headings
    # outside (-pi, pi] so that the un-wrapped angle binning of logistic_regression.py:28-42 is exercised
    rng = np.random.default_rng(20240132)
    trajs = []
    for i in range(16):
        x, y, th, v, a = make_traj(rng, 31, dt, psi0=2.9)
        trajs.append(Traj(x, y, th + (2 * np.pi if i % 4 == 1 else 0.0), v, a))
    mid = np.array([trajs[0].cartesian.x[12], trajs[0].cartesian.y[12]])
    kinds = ["Pedestrian", "Car", "Bicycle", "Truck", "Car", "Pedestrian"]
    lens = [31, 20, 40, 5, 1, 30]
    preds = [make_prediction(rng, k, L, dt, mid, zero_cov=(i == 1), curved=(i == 2)) for i, (k, L) in enumerate(zip(kinds, lens))]
    preds[4]["orientation_list"] = preds[4]["orientation_list"] + 2 * np.pi
    run_case("random_ragged", trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC)

    # case 3: short trajectories (T = 12) against full-length predictions
    rng = np.random.default_rng(20240133)
    trajs = [Traj(*make_traj(rng, 12, dt, psi0=-1.0)) for _ in range(8)]
    mid = np.array([trajs[0].cartesian.x[6], trajs[0].cartesian.y[6]])
    kinds = ["Car", "Pedestrian", "Bicycle"]
    preds = [make_prediction(rng, k, 31, dt, mid) for k in kinds]
    run_case("short_traj", trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC)

    # case 4: a ring of agents around the trajectories' mid-point -- every bearing x many headings, so that both
    # impact angles visit all front / side / rear bins of logistic_regression.py:28-42 -- with every obstacle type whose
    # enum value survives `.lower()` (harm_model.py:61,78): mass table and protection flags of :15-32,158-190
    rng = np.random.default_rng(20240134)
    trajs = [Traj(*make_traj(rng, 31, dt, psi0=0.7)) for _ in range(12)]
    mid = np.array([trajs[0].cartesian.x[14], trajs[0].cartesian.y[14]])
    ring_kinds = ["Car", "Truck", "Bus", "Taxi", "Train", "Motorcycle", "Unknown", "Bicycle", "Pedestrian"]
    kinds, preds = [], []
    for j in range(36):
        kind = ring_kinds[j % len(ring_kinds)]
        bearing = 2 * np.pi * j / 36
        rad = 4.0 + 9.0 * ((j * 7) % 5) / 4.0
        heading = bearing * 3.0 + 0.4 * j                      # decorrelated from the bearing, runs past 2 pi
        kinds.append(kind)
        preds.append(make_prediction(rng, kind, 31, dt, mid, heading=heading,
                                     offset=rad * np.array([np.cos(bearing), np.sin(bearing)])))
    run_case("ring_all_types", trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC)

    # case 5: full covariance matrices (what a prediction module hands over for real agents): correlation from -0.95 to
    # 0.985, unequal and growing variances -- scipy's mvnun integrates the bivariate normal exactly (MVNDST -> BVNMVN)
    rng = np.random.default_rng(20240135)
    trajs = [Traj(*make_traj(rng, 31, dt, psi0=-0.4)) for _ in range(10)]
    mid = np.array([trajs[0].cartesian.x[13], trajs[0].cartesian.y[13]])
    kinds = ["Car", "Pedestrian", "Bicycle", "Truck", "Pedestrian", "Car", "Bicycle", "Pedestrian", "Car", "Pedestrian"]
    rhos = [0.6, -0.8, 0.95, -0.95, 0.3, -0.15, 0.9, 0.0, 0.985, -0.9]
    preds = []
    for i, (k, rho) in enumerate(zip(kinds, rhos)):
        off = rng.uniform(-2.5, 2.5, size=2)
        p = make_prediction(rng, k, 31, dt, mid, offset=off if i < 8 else 0.15 * off)
        L = len(p["pos_list"])
        sx = np.sqrt(0.08 * 1.05 ** np.arange(L)) * (1.0 + 0.5 * (i % 3))
        sy = np.sqrt(0.12 * 1.04 ** np.arange(L))
        if i >= 8:                             # tight, strongly correlated: standard deviations of 4 to 9 cm, so the
            sx, sy = 0.15 * sx, 0.2 * sy       # boxes span dozens of them (hard for a quadrature in x)
        r = rho * np.cos(0.05 * np.arange(L)) if i % 2 else np.full(L, rho)      # some correlations change over time
        cov = np.zeros((L, 2, 2))
        cov[:, 0, 0], cov[:, 1, 1] = sx * sx, sy * sy
        cov[:, 0, 1] = cov[:, 1, 0] = r * sx * sy
        if i == 4:
            cov[:6] = 0.0                      # zero blocks fall back to 0.1 I (collision_probability.py:84-86)
        p["cov_list"] = cov
        preds.append(p)
    run_case("correlated_cov", trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC)

    # case 6: the impact-angle bins of the LR4S model (logistic_regression.py:28-42) and their boundaries.  The ego
    # stands at the origin with seven headings (one trajectory each); 240 cars sit on a circle of 12 m at 1.5 deg steps,
    # once with yaw 0 and once with yaw pi/2.  The positions at multiples of 45 deg are exact ((12, 0), (8.5, 8.5), ...),
    # so for the headings that are multiples of 45 deg the un-wrapped angle lands ON a boundary as a float (atan2(1, 1)
    # is the float pi/4 = 45/180*pi): what the reference's comparisons make of those is part of the contract, and
    # differs from what exact geometry says (cos of the float -pi/4 is one ulp above |sin|: "front", the reference:
    # "side").  Elsewhere the neighbouring floats of the boundaries are visited.
    T = 2
    heads = [0.0, np.pi / 2, -np.pi / 4, 0.7, 5.5, -3.0 * np.pi / 4, np.pi]
    trajs = [Traj(np.zeros(T), np.zeros(T), np.full(T, h), np.full(T, 0.5), np.zeros(T)) for h in heads]
    nb = 240
    exact = [(12.0, 0.0), (8.5, 8.5), (0.0, 12.0), (-8.5, 8.5), (-12.0, 0.0), (-8.5, -8.5), (0.0, -12.0), (8.5, -8.5)]
    kinds, preds = [], []
    for psi in (0.0, np.pi / 2):
        for i in range(nb):
            beta = 2.0 * np.pi * i / nb
            p0 = np.array(exact[i // 30]) if i % 30 == 0 else 12.0 * np.array([np.cos(beta), np.sin(beta)])
            kinds.append("Car")
            preds.append({"pos_list": np.tile(p0, (T, 1)), "v_list": np.full(T, 2.0), "orientation_list": np.full(T, psi),
                          "cov_list": np.tile(0.1 * np.eye(2), (T, 1, 1)),
                          "shape": {"length": RAW_DIMS["Car"][0] * 1.2, "width": RAW_DIMS["Car"][1] * 1.3}})
    run_case("angle_bins", trajs, kinds, preds, dt, CP, HR, TTC, TTCE, WTTC)

    # case 7: the safety decision itself -- the reference's Metric.evaluate_metrics (metric.py:35-100) with its own
    # threshold logic and dependency closure, for several activated-metric lists and threshold sets, on a batch that
    # has safe and unsafe candidates under each of them.  'dce' results are fed from this repository's oracle
    # (_OracleDCE above); everything else is the reference's own arithmetic.
    from frenetix_occlusion.metrics.metric import Metric
    rng = np.random.default_rng(20240137)
    trajs = [Traj(*make_traj(rng, 31, dt, psi0=0.2)) for _ in range(40)]
    mid = np.array([trajs[0].cartesian.x[12], trajs[0].cartesian.y[12]])
    kinds = ["Pedestrian", "Car", "Bicycle", "Pedestrian", "Truck", "Car"]
    # agents ahead of the slower candidates and to the side of the lane: part of the batch reaches them, part does not
    offs = [(6.0, 3.5), (14.0, -6.5), (9.0, -4.5), (3.0, -3.5), (25.0, 8.0), (11.0, 6.0)]
    preds = [make_prediction(rng, k, 31 if i != 4 else 19, dt, mid, offset=np.array(offs[i]),
                             heading=[-1.4, 0.2, 1.8, 1.5, 3.3, -2.0][i]) for i, k in enumerate(kinds)]
    allm = ("hr", "ttc", "ttce", "dce", "wttc", "cp")
    configs = [
        (allm, {"harm": 0.1, "risk": 1}),                                  # occlusion.yaml of the reference's example
        (allm, {"harm": 1, "risk": 1}),                                    # config.yaml defaults: nothing trips
        (("hr", "ttc"), {"harm": 0.3, "risk": 0.05}),                     # BASELINE configs[0]: closure adds cp, dce
        (allm, {"cp": ("q", 0.5)}),
        (allm, {"ttc": ("q", 0.5)}),
        (allm, {"dce": 1.0}),
        (("ttc",), {"ttc": ("q", 0.3), "harm": 0.0}),                      # harm threshold without 'hr': not checked
        (("wttc",), {"wttc": 9.0, "ttce": 9.0}),                           # thresholds the reference never checks
        (("cp",), {"cp": 0.0001}),                                         # cp threshold needs 'hr' (metric.py:79)
        (allm, {"harm": ("q", 0.8), "risk": ("q", 0.8), "cp": ("q", 0.8), "ttc": ("q", 0.2), "dce": 0.05}),
    ]
    run_threshold_case("thresholds", trajs, kinds, preds, dt, Metric, configs)

    # case 8: pedestrian predictions of the reference's own agent class
    run_pedestrian_predictions("ped_predictions", dt)

    # case 9: obstacle state cache of the reference
    run_obstacle_states("obstacle_states")

    # case 10: route enumeration of the reference's route planner
    run_route_enumeration("routes")


def gen_sampling_matrix():
    """the matrix the reference hands to the frenetix sampler for a phantom vehicle (utils/frenetix_handler.py:82-105),
    from its own, natively importable utils/sampling.py -- inputs and outputs only"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_sampling", os.path.join(REF, "frenetix_occlusion", "utils", "sampling.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sh = m.SamplingHandler(dt=0.1, max_sampling_number=1, t_min=2.0, horizon=3.0, delta_d_max=0.5, delta_d_min=-0.5)
    d1_range = np.array(list(sh.d_sampling.to_range(0)))
    cases = np.array([[12.5, 1.2, 8.33], [0.0, 0.0, 30 / 3.6], [40.0, -0.3, 5.0], [7.0, 0.25, 13.9], [3.0, -2.0, 10.0]])
    mats = []
    for s0, d0, v0 in cases:
        mats.append(m.generate_sampling_matrix(t0_range=0.0, t1_range=3.0, s0_range=s0, ss0_range=v0, sss0_range=0,
                                               ss1_range=np.array([v0 * 0.8, v0, v0 * 1.2]), sss1_range=0, d0_range=d0,
                                               dd0_range=0, ddd0_range=0, d1_range=d1_range, dd1_range=0.0, ddd1_range=0.0))
    np.savez(os.path.join(OUT, "sampling_matrix.npz"), cases=cases, d1_range=d1_range, matrices=np.array(mats))
    print("sampling_matrix.npz", np.array(mats).shape)


def gen_shadow_geometry():
    """The shadow geometry of the reference's sensor model, from its OWN arithmetic: utils/helper_functions.py
    (create_polygon_from_vertices :79-96, get_polygon_from_obstacle_occlusion / _identify_projection_points :139-176) and
    sensor_model.SensorModel._calc_relevant_sector (:201-209) are pure numpy up to the final shapely ``Polygon(...)``
    constructor.  Imported unmodified with ``shapely.geometry.Polygon`` replaced by a class that RECORDS its vertex list,
    they hand over the reference's own quad / wedge / sector vertices for random inputs -- inputs and outputs only go to
    shadow_geometry.npz.  (What GEOS then does with those polygons -- difference, intersection -- is not run: absent.)"""
    class RecPolygon:
        def __init__(self, shell=None, holes=None):
            self.vertices = np.array([np.asarray(v, dtype=np.float64) for v in shell], dtype=np.float64)

    class Dummy:
        def __init__(self, *a, **kw):
            pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    mp = mod("shapely.geometry.multipolygon", MultiPolygon=Dummy)
    mod("shapely", geometry=mod("shapely.geometry", Polygon=RecPolygon, Point=Dummy, MultiPolygon=Dummy, LineString=Dummy,
                                multipolygon=mp),
        affinity=mod("shapely.affinity", rotate=Dummy, translate=Dummy))
    sys.path.insert(0, REF)
    import frenetix_occlusion.utils.helper_functions as hf
    from frenetix_occlusion.sensor_model import SensorModel
    rng = np.random.default_rng(20240141)
    # (a) boundary shadows: quad of a vertex pair (helper_functions.py:79-96)
    n = 200
    ego_a = rng.uniform(-60.0, 60.0, size=(n, 2))
    ang = rng.uniform(0.0, 2.0 * np.pi, n)
    dist = rng.uniform(0.2, 55.0, n)
    v1 = ego_a + dist[:, None] * np.stack((np.cos(ang), np.sin(ang)), -1)
    v2 = v1 + rng.uniform(-4.0, 4.0, size=(n, 2))
    v2[:5] = v1[:5]                                    # degenerate pairs (a repeated vertex)
    v2[5:10] = ego_a[5:10] + 2.5 * (v1[5:10] - ego_a[5:10])     # collinear with the ego: zero-area quad
    quads = np.array([hf.create_polygon_from_vertices(v1[i], v2[i], ego_a[i]).vertices for i in range(n)])
    # (b) obstacle shadows: silhouette pair + wedge (helper_functions.py:139-176)
    m = 200
    ego_b = rng.uniform(-40.0, 40.0, size=(m, 2))
    yaw = rng.uniform(-np.pi, np.pi, m)
    length, width = rng.uniform(0.5, 12.0, m), rng.uniform(0.4, 3.0, m)
    cdist = rng.uniform(0.3, 50.0, m)
    cang = rng.uniform(0.0, 2.0 * np.pi, m)
    cen = ego_b + cdist[:, None] * np.stack((np.cos(cang), np.sin(cang)), -1)
    # special geometries: axis-aligned rectangles seen along an axis (ties among the 16 angles), the ego on the extension of
    # a side, the ego close to a corner, the ego almost touching a long side
    yaw[:8] = 0.0
    cen[:8] = ego_b[:8] + np.array([[10.0, 0.0], [0.0, 7.0], [-6.0, 0.0], [0.0, -3.0], [5.0, 5.0], [-5.0, 5.0], [12.0, 0.0], [0.0, 9.0]])
    length[:8], width[:8] = 4.0, 2.0
    yaw[8:12] = np.pi / 2
    cen[8:12] = ego_b[8:12] + np.array([[3.0, 0.0], [1.0 + 1e-9, 5.0], [-1.0, -7.0], [20.0, 2.0]])
    length[8:12], width[8:12] = 4.0, 2.0
    corners = np.zeros((m, 4, 2))
    for i in range(m):
        base = np.array([[-length[i] / 2, -width[i] / 2], [-length[i] / 2, width[i] / 2], [length[i] / 2, width[i] / 2],
                         [length[i] / 2, -width[i] / 2]])
        c, s_ = np.cos(yaw[i]), np.sin(yaw[i])
        corners[i] = base @ np.array([[c, s_], [-s_, c]]) + cen[i]
    for i in range(12, 20):                           # the ego 5 cm .. 1 m off a corner, outside the rectangle
        k = i % 4
        out_dir = corners[i, k] - cen[i]
        ego_b[i] = corners[i, k] + out_dir / np.linalg.norm(out_dir) * (0.05 + 0.12 * (i - 12))
    for i in range(20, 26):                           # the ego 0.3 .. 1.5 m off the middle of a side
        a, b = corners[i, 0], corners[i, 1]
        nrm = np.array([-(b - a)[1], (b - a)[0]])
        nrm = nrm / np.linalg.norm(nrm)
        if np.dot(nrm, a - cen[i]) < 0:
            nrm = -nrm
        ego_b[i] = 0.5 * (a + b) + nrm * (0.3 + 0.24 * (i - 20))
    wedges, c1s, c2s = np.zeros((m, 4, 2)), np.zeros((m, 2)), np.zeros((m, 2))
    for i in range(m):
        poly, c1, c2 = hf.get_polygon_from_obstacle_occlusion(ego_b[i], corners[i])
        wedges[i], c1s[i], c2s[i] = poly.vertices, c1, c2
    # (c) sectors (sensor_model.py:201-209): full / open fans, factors 1.0 and 1.5 (the occluded area's half disc, :85-87)
    cases = []
    for i in range(12):
        ego = rng.uniform(-50.0, 50.0, 2)
        yaw_e = rng.uniform(-np.pi, np.pi)
        r = [30.0, 50.0, 80.0][i % 3]
        fov = [90.0, 120.0, 180.0, 270.0, 359.0, 45.0][i % 6]
        half = np.radians(fov) / 2.0
        for a0, a1, fac in ((yaw_e - half, yaw_e + half, 1.0), (yaw_e - np.pi / 2, yaw_e + np.pi / 2, 1.5)):
            self_ = types.SimpleNamespace(ego_pos=ego, sensor_radius=r)
            sec = SensorModel._calc_relevant_sector(self_, a0, a1, factor=fac)
            cases.append((ego, r, a0, a1, fac, sec.vertices))
    np.savez_compressed(os.path.join(OUT, "shadow_geometry.npz"),
                        quad_ego=ego_a, quad_v1=v1, quad_v2=v2, quad_ref=quads,
                        wedge_ego=ego_b, wedge_corners=corners, wedge_ref=wedges, wedge_c1=c1s, wedge_c2=c2s,
                        sector_ego=np.array([c[0] for c in cases]), sector_r=np.array([c[1] for c in cases]),
                        sector_a0=np.array([c[2] for c in cases]), sector_a1=np.array([c[3] for c in cases]),
                        sector_factor=np.array([c[4] for c in cases]), sector_ref=np.array([c[5] for c in cases]))
    print("shadow_geometry.npz quads", quads.shape, "wedges", wedges.shape, "sectors", np.array([c[5] for c in cases]).shape)


def _install_rectangle_stub(O):
    """frenetix_occlusion.metrics.utils.convert_dynamic_obstacle (commonroad proper: not installable here) as a module whose two
    converters build duck-typed obstacles -- `.prediction.trajectory.state_list`, `.occupancy_at_time(i)` (None beyond the last
    state, as commonroad's) with `.shape.shapely_object` -- over this repository's rectangle primitives: vertices by
    oracle.rect_vertices (rear axle -> centre as convert_dynamic_obstacle.py:73), `distance` = oracle.quad_distance (the
    restatement of GEOS' polygon distance), `intersects` = that distance == 0.  Everything the reference's metric classes do ON
    TOP of those two predicates then runs unmodified."""
    class _Poly:
        def __init__(self, q):
            self.q = q

        def distance(self, other):
            return O.quad_distance(self.q, other.q)

        def intersects(self, other):
            return O.quad_distance(self.q, other.q) == 0.0

    class _NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class _Dyn:
        def __init__(self, quads):
            self._q = quads
            self.prediction = _NS(trajectory=_NS(state_list=[None] * len(quads)))

        def occupancy_at_time(self, i):
            return None if i >= len(self._q) else _NS(shape=_NS(shapely_object=_Poly(self._q[i])))

    def convert_traj_to_dyn_obstacle(trajectory, vehicle_params, obstacle_id=42):
        c = trajectory.cartesian
        wb = vehicle_params.wb_rear_axle
        return _Dyn([O.rect_vertices(x + wb * np.cos(t), y + wb * np.sin(t), t, vehicle_params.length, vehicle_params.width)
                     for x, y, t in zip(c.x, c.y, c.theta)])

    def convert_prediction_to_dyn_obstacle(agent, key, prediction):
        return _Dyn([O.rect_vertices(p[0], p[1], yaw, agent.shape.length, agent.shape.width)
                     for p, yaw in zip(prediction["pos_list"], prediction["orientation_list"])])

    mod = types.ModuleType("frenetix_occlusion.metrics.utils.convert_dynamic_obstacle")
    mod.convert_traj_to_dyn_obstacle = convert_traj_to_dyn_obstacle
    mod.convert_prediction_to_dyn_obstacle = convert_prediction_to_dyn_obstacle
    sys.modules["frenetix_occlusion.metrics.utils.convert_dynamic_obstacle"] = mod
    return _NS


def gen_dce():
    """tests/golden/dce_loop.npz: the reference's own, unmodified metrics/dce.py (DCE.evaluate / _calc_dce, dce.py:31-99: the
    walk over the time steps, np.round(distance, 3), 'first strict minimum', the stop at the first zero and at the end of a
    prediction) with metrics/ttc.py, ttce.py and wttc.py on top of ITS results -- over the stubbed rectangles of
    _install_rectangle_stub: the polygon distance under the loop is this repository's restatement of GEOS', the loop is the
    reference's.  Ragged predictions, overlaps at t = 0 and later, touching rectangles, stationary pairs (every step ties),
    symmetric pass-bys (two equal minima), pairs that never come near."""
    _install_aliases()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import fo_oracle as O
    O.build()
    _NS = _install_rectangle_stub(O)
    from frenetix_occlusion.metrics.dce import DCE         # the reference's own classes, unmodified
    from frenetix_occlusion.metrics.ttc import TTC
    from frenetix_occlusion.metrics.ttce import TTCE
    from frenetix_occlusion.metrics.wttc import WTTC
    dt, T = 0.1, 31
    t = np.arange(T) * dt
    rng = np.random.default_rng(20240207)
    vp = VehicleParams()
    trajs = [Traj(*make_traj(rng, T, dt, psi0=rng.uniform(-0.4, 0.4))) for _ in range(28)]
    trajs.append(Traj(np.full(T, -40.0), np.full(T, 10.0), np.zeros(T), np.zeros(T), np.zeros(T)))           # ego at rest, away from the others' start
    xs = np.concatenate((np.linspace(0, 9, 16), np.linspace(9, 0, 16)[1:]))                                # out and back: symmetric
    trajs.append(Traj(xs, np.zeros(T), np.zeros(T), np.full(T, 6.0), np.zeros(T)))
    trajs.append(Traj(6.0 * t, np.zeros(T), np.zeros(T), np.full(T, 6.0), np.zeros(T)))                     # straight
    mid = np.array([trajs[0].cartesian.x[15], trajs[0].cartesian.y[15]])
    kinds = [AGENT_TYPES[i % 4] for i in range(12)] + ["Car", "Car", "Car", "Pedestrian"]
    lens = [31, 31, 20, 31, 9, 31, 1, 40, 31, 15, 31, 2, 31, 31, 31, 31]
    preds = [make_prediction(rng, k, L, dt, mid, curved=(i % 3 == 0)) for i, (k, L) in enumerate(zip(kinds[:12], lens[:12]))]
    touch_x = vp.wb_rear_axle + vp.length / 2 + RAW_DIMS["Car"][0] / 2          # faces touch the ego at rest: distance 0
    for p0, L in (((30.0, 0.0), 31), ((15.0, 0.0), 31), ((-40.0 + touch_x, 10.0), 31), ((20.0, 6.0), 31)):
        kind = kinds[len(preds)]
        spd = 0.0 if kind == "Car" else 4.0
        psi = 0.0 if kind == "Car" else -np.pi / 2
        tt = np.arange(L) * dt
        pos = np.asarray(p0) + tt[:, None] * spd * np.array([np.cos(psi), np.sin(psi)])
        preds.append({"pos_list": pos, "v_list": np.full(L, spd), "orientation_list": np.full(L, psi),
                      "cov_list": np.tile(0.1 * np.eye(2), (L, 1, 1)),
                      "shape": {"length": RAW_DIMS[kind][0] * 1.2, "width": RAW_DIMS[kind][1] * 1.3}})
    am = AgentManager(dt)
    keys = []
    for i, (kind, pred) in enumerate(zip(kinds, preds)):
        ag = Agent(10000 + i, kind)
        ag.shape = _NS(length=RAW_DIMS[kind][0], width=RAW_DIMS[kind][1])      # agent.py:216: the RAW dimensions
        ag.obstacle_type = kind
        am.phantom_agents.append(ag)
        key = int(str(10000 + i) + "0")
        am.predictions[key] = pred
        keys.append(key)
    dce_m, ttc_m, ttce_m, wttc_m = DCE(vp, am), TTC(am), TTCE(am), WTTC()
    M, A = len(trajs), len(keys)
    out_d, out_t, out_ttc, out_ttce, out_w = (np.zeros((M, A)), np.zeros((M, A), dtype=np.int64), np.zeros((M, A)),
                                              np.zeros((M, A)), np.zeros(M))
    for m, tr in enumerate(trajs):
        res = {"dce": dce_m.evaluate(tr, {})}
        res["ttc"] = ttc_m.evaluate(tr, res)
        res["ttce"] = ttce_m.evaluate(tr, res)
        res["wttc"] = wttc_m.evaluate(tr, res)
        for j, k in enumerate(keys):
            out_d[m, j], out_t[m, j] = res["dce"][k]["dce"], res["dce"][k]["time_dce"]
            out_ttc[m, j], out_ttce[m, j] = res["ttc"][k], res["ttce"][k]
        out_w[m] = res["wttc"]
    Lmax = max(lens)
    codes = {"car": 0, "truck": 1, "bus": 2, "bicycle": 3, "pedestrian": 4}
    agents = {
        "pos": np.stack([np.pad(p["pos_list"], ((0, Lmax - len(p["pos_list"])), (0, 0))) for p in preds]),
        "yaw": pad([p["orientation_list"] for p in preds], Lmax, 0.0), "v": pad([p["v_list"] for p in preds], Lmax, 0.0),
        "cov": np.stack([np.pad(p["cov_list"], ((0, Lmax - len(p["cov_list"])), (0, 0), (0, 0))) for p in preds]),
        "shape": np.array([[p["shape"]["length"], p["shape"]["width"]] for p in preds]),
        "raw_dims": np.array([RAW_DIMS[k] for k in kinds]), "type": np.array([codes[k.lower()] for k in kinds], dtype=np.int32),
        "len": np.array(lens, dtype=np.int32)}
    outd = {"dt": dt, "vehicle": np.array([vp.length, vp.width, vp.wb_rear_axle, vp.mass, vp.a_max]),
            **{"traj_" + k: np.stack([getattr(tr.cartesian, k) for tr in trajs]) for k in ("x", "y", "theta", "v", "a")},
            "agent_type": np.array(kinds), **{"agent_" + k: v for k, v in agents.items() if k != "type"},
            "agent_type_code": agents["type"], "ref_dce": out_d, "ref_time_dce": out_t, "ref_ttc": out_ttc, "ref_ttce": out_ttce,
            "ref_wttc": out_w}
    path = os.path.join(OUT, "dce_loop.npz")
    np.savez_compressed(path, **outd)
    print("wrote", path, "pairs", M * A, "zero distances", int((out_d == 0).sum()), "time_dce > 0", int((out_t > 0).sum()),
          "finite ttc", int(np.isfinite(out_ttc).sum()), "distinct time_dce", len(np.unique(out_t)))


def gen_be():
    """tests/golden/be_bisection.npz: the reference's own, unmodified metrics/be.py (BE.evaluate -> find_minimum_deceleration ->
    _calc_deceleration_trajectory -> _collision_check, be.py:31-193) and metrics/ttc.py on top of it.  be.py's one import
    besides numpy / scipy, metrics/utils/convert_dynamic_obstacle.py, needs commonroad proper and is replaced by a module
    whose two converters build duck-typed obstacles: `.prediction.trajectory.state_list`, `.occupancy_at_time(i)` (None
    beyond the last state) with `.shape.shapely_object.intersects(other)`.  Only that predicate is NOT the reference's: it is
    this repository's rectangle predicate (oracle quad_distance == 0, the restatement of GEOS' distance; rear axle -> centre as
    convert_dynamic_obstacle.py:73).  Everything BE adds on top -- the speed profile of a constant deceleration, the
    re-sampling of the path by scipy's interp1d, the bisection with its bracket, rounding, iteration count and stop rule, the
    brake threat number, which pairs get a value at all (`ttc is not np.inf`, `ttc > 0`) -- is the reference's code."""
    _install_aliases()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import fo_oracle as O
    O.build()

    _NS = _install_rectangle_stub(O)
    from frenetix_occlusion.metrics.be import BE          # the reference's own class, unmodified
    from frenetix_occlusion.metrics.ttc import TTC

    dt, T = 0.1, 31
    t = np.arange(T) * dt
    rng = np.random.default_rng(20240206)
    vp = VehicleParams()

    def ego(v0, a_prof, psi0, kappa):
        """a path integrated sample by sample (chord = v dt along the heading), so that the travelled length is the sum of
        the speeds: a constant deceleration from v[1] on never runs ahead of a profile that brakes less hard than it"""
        v = np.maximum(v0 + np.concatenate(([0.0], np.cumsum(a_prof[:-1] * dt))), 0.5)
        th = psi0 + kappa * np.concatenate(([0.0], np.cumsum(v[:-1] * dt)))
        x = np.concatenate(([0.0], np.cumsum(v[:-1] * np.cos(th[:-1]) * dt)))
        y = np.concatenate(([0.0], np.cumsum(v[:-1] * np.sin(th[:-1]) * dt)))
        return Traj(x, y, th, v, np.gradient(v, dt))

    trajs = []
    for i in range(40):
        v0 = rng.uniform(4.0, 14.0)
        kind = i % 4
        a_prof = (np.zeros(T) if kind == 0 else np.full(T, -rng.uniform(0.2, 2.5)) if kind == 1 else
                  rng.uniform(0.2, 1.5) * np.ones(T) if kind == 2 else -rng.uniform(0.5, 3.0) * (t > rng.uniform(0.3, 2.0)))
        trajs.append(ego(v0, a_prof, rng.uniform(-0.2, 0.2), rng.uniform(-0.03, 0.03)))
    # agents: standing and slow vehicles ahead on the candidates' way (the bisection has something to find), crossing
    # pedestrians and a bicycle, an oncoming car, one far away (no collision: no value), one on top of the ego at t = 0
    # (ttc = 0: no value), short predictions
    kinds = ["Car", "Car", "Truck", "Pedestrian", "Pedestrian", "Bicycle", "Car", "Car", "Car", "Truck"]
    specs = [((14.0, 0.3), 0.0, 0.0, 31), ((22.0, -0.5), 0.1, 2.0, 31), ((30.0, 1.0), -0.1, 1.0, 31),
             ((12.0, -4.0), np.pi / 2, 1.4, 31), ((18.0, 5.0), -np.pi / 2, 1.6, 25), ((16.0, -6.0), np.pi / 2 - 0.2, 4.0, 31),
             ((45.0, 0.5), np.pi, 8.0, 31), ((200.0, 50.0), 0.3, 5.0, 31), ((1.0, 0.0), 0.0, 0.0, 31), ((10.0, 0.0), 0.05, 0.5, 12)]
    am = AgentManager(dt)
    preds, keys = [], []
    for i, (kind, (p0, psi, spd, L)) in enumerate(zip(kinds, specs)):
        ag = Agent(10000 + i, kind)
        ag.shape = _NS(length=RAW_DIMS[kind][0], width=RAW_DIMS[kind][1])      # agent.py:216: the RAW dimensions
        ag.obstacle_type = kind
        am.phantom_agents.append(ag)
        tt = np.arange(L) * dt
        pos = np.asarray(p0) + tt[:, None] * spd * np.array([np.cos(psi), np.sin(psi)])
        pred = {"pos_list": pos, "v_list": np.full(L, spd), "orientation_list": np.full(L, psi),
                "cov_list": np.tile(0.1 * np.eye(2), (L, 1, 1)), "shape": {"length": RAW_DIMS[kind][0] * 1.2, "width": RAW_DIMS[kind][1] * 1.3}}
        key = int(str(10000 + i) + "0")
        am.predictions[key] = pred
        preds.append(pred)
        keys.append(key)
    Lmax = T
    codes = {"car": 0, "truck": 1, "bus": 2, "bicycle": 3, "pedestrian": 4}
    keep, decel, btn, ttc_out = [], [], [], []
    be, ttc_m = BE(vp, am), TTC(am)
    for m, tr in enumerate(trajs):
        traj1 = {k: getattr(tr.cartesian, k)[None] for k in ("x", "y", "theta", "v", "a")}
        agents = {
            "pos": np.stack([np.pad(p["pos_list"], ((0, Lmax - len(p["pos_list"])), (0, 0))) for p in preds]),
            "yaw": pad([p["orientation_list"] for p in preds], Lmax, 0.0), "v": pad([p["v_list"] for p in preds], Lmax, 0.0),
            "cov": np.stack([np.pad(p["cov_list"], ((0, Lmax - len(p["cov_list"])), (0, 0), (0, 0))) for p in preds]),
            "shape": np.array([[p["shape"]["length"], p["shape"]["width"]] for p in preds]),
            "raw_dims": np.array([RAW_DIMS[k] for k in kinds]), "type": np.array([codes[k.lower()] for k in kinds], dtype=np.int32),
            "len": np.array([len(p["pos_list"]) for p in preds], dtype=np.int32)}
        orc = O.sweep(traj1, agents, (vp.length, vp.width, vp.wb_rear_axle, vp.mass, vp.a_max), dt, metrics=("dce", "ttc"))
        res = {"dce": {key: {"dce": float(orc["pair_f"][0, j, O.PF["dce"]]), "time_dce": int(orc["pair_i"][0, j, O.PI["time_dce"]])}
                       for j, key in enumerate(keys)}}
        res["ttc"] = ttc_m.evaluate(tr, res)                  # the reference's own TTC (np.inf by identity where no collision)
        try:
            out = be.evaluate(tr, res)
        except ValueError:      # scipy's interp1d refuses a re-sampled arc length beyond the path: the reference raises; not a case
            continue
        keep.append(m)
        decel.append([out[k]["required_constant_deceleration"] for k in keys])
        btn.append([out[k]["break_threat_number"] for k in keys])
        ttc_out.append([float(res["ttc"][k]) for k in keys])
    keep = np.array(keep)
    outd = {"dt": dt, "vehicle": np.array([vp.length, vp.width, vp.wb_rear_axle, vp.mass, vp.a_max]),
            **{"traj_" + k: np.stack([getattr(trajs[m].cartesian, k) for m in keep]) for k in ("x", "y", "theta", "v", "a")},
            "agent_type": np.array(kinds), **{"agent_" + k: v for k, v in agents.items() if k != "type"},
            "agent_type_code": agents["type"], "be_decel": np.array(decel), "be_btn": np.array(btn), "ttc": np.array(ttc_out),
            "dropped_by_the_reference": len(trajs) - len(keep)}
    path = os.path.join(OUT, "be_bisection.npz")
    np.savez_compressed(path, **outd)
    d = outd["be_decel"]
    print("wrote", path, "trajectories", len(keep), "of", len(trajs), "pairs with a value", int((d > 0).sum()), "of", d.size,
          "distinct decelerations", len(np.unique(np.round(d[d > 0], 6))), "range", float(d[d > 0].min()), float(d.max()))


def gen_relevant_lanelets():
    """The topology side of the dynamic-obstacle rule, from the reference's OWN SpawnLocator methods (spawn_locator.py, imported
    unmodified; commonroad_dc / shapely / commonroad_route_planner as inert stubs -- none of them is touched on this path):
    ``_find_relevant_intersection`` + ``_find_intersection_lanelets`` (:584-622: the first intersection in list order whose
    incoming or successor lanelets hold the ego's lanelet), ``_find_lanelets_along_reference`` (:624-635: every fifth vertex of
    the reference window, the first lanelet at each, no repeats) with ``_find_lanelet_by_position`` (:666-676) underneath,
    composed as lines :186-202 compose them, and ``_find_nearest_index`` / ``_find_ego_intention`` (:729-750; the curvature
    itself is commonroad_dc's and is an INPUT here).  The scenario side is duck-typed: random lanelet networks and
    intersections, 'which lanelets hold this point' from a table.  Inputs and outputs only go to relevant_lanelets.npz."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class Dummy:
        def __init__(self, *a, **kw):
            pass
    curv_box = {}
    mod("commonroad_dc", geometry=mod("commonroad_dc.geometry", util=mod(
        "commonroad_dc.geometry.util", compute_pathlength_from_polyline=lambda p: None,
        compute_curvature_from_polyline=lambda ref: curv_box["k"])))
    mp = mod("shapely.geometry.multipolygon", MultiPolygon=Dummy)
    mod("shapely", geometry=mod("shapely.geometry", Polygon=Dummy, Point=Dummy, MultiPolygon=Dummy, LineString=Dummy, multipolygon=mp),
        affinity=mod("shapely.affinity", rotate=Dummy, translate=Dummy), ops=mod("shapely.ops", unary_union=Dummy))
    mod("commonroad_route_planner", utility=mod("commonroad_route_planner.utility", route=mod(
        "commonroad_route_planner.utility.route", lanelet_orientation_at_position=Dummy)))
    sys.path.insert(0, REF)
    from frenetix_occlusion.spawn_locator import SpawnLocator
    NS = types.SimpleNamespace
    rng = np.random.default_rng(20240606)
    cases = []
    for c in range(80):
        n = int(rng.integers(6, 40))
        ids = [int(x) for x in rng.choice(np.arange(100, 400), size=n, replace=False)]
        adj = {i: (int(rng.choice(ids)) if rng.random() < 0.6 else None) for i in ids}
        lanelets = {i: NS(lanelet_id=i, adj_left=adj[i]) for i in ids}
        inters = []
        for _ in range(int(rng.integers(0, 4))):
            incs = []
            for _ in range(int(rng.integers(1, 5))):
                pick = lambda hi: set(int(x) for x in rng.choice(ids, size=int(rng.integers(0, hi)), replace=False))
                incs.append(NS(incoming_lanelets=pick(3) | {int(rng.choice(ids))}, successors_left=pick(3), successors_right=pick(3),
                               successors_straight=pick(3)))
            inters.append(NS(incomings=incs))
        # the reference window: points with a table of the lanelets that hold each (none, one or two; the first one counts)
        npts = int(rng.integers(3, 60))
        pts = [np.array([float(k), float(c)]) for k in range(npts)]
        at = {tuple(q): [int(x) for x in rng.choice(ids, size=int(rng.integers(0, 3)), replace=False)] for q in pts}
        ego = np.array([-1.0, float(c)])
        at[tuple(ego)] = [int(rng.choice(ids))] if rng.random() < 0.9 else []
        if c % 4 == 0 and inters:   # the ego on an intersection's lanelet for sure
            e0 = inters[int(rng.integers(len(inters)))].incomings[0]
            at[tuple(ego)] = [sorted(e0.incoming_lanelets)[0]]
        net = NS(intersections=inters, find_lanelet_by_id=lambda i: lanelets[i],
                 find_lanelet_by_position=lambda plist: [list(at[tuple(np.asarray(plist[0], dtype=np.float64))])])
        sl = object.__new__(SpawnLocator)
        sl.scenario = NS(lanelet_network=net)
        sl.ego_pos = ego
        sl.reference = pts
        ego_ll = sl._find_lanelet_by_position(sl.ego_pos)
        if ego_ll is None:
            found, rel, inner_ids = -1, [], []      # (the reference goes on to ego_lanelet.lanelet_id and raises: not a case)
        else:
            inter, inc, inner = sl._find_relevant_intersection(net.intersections, ego_ll)
            if inter:                                # lines :186-198 of the rule, verbatim
                import copy
                relevant = copy.deepcopy(inc)
                relevant.update(inner)
                relevant.remove(ego_ll.lanelet_id)
                found, rel, inner_ids = inters.index(inter), sorted(relevant), sorted(inner)
            else:                                    # :199-202
                along = sl._find_lanelets_along_reference(sl.reference, step=5)
                found, rel, inner_ids = -1, [(-1 if l.adj_left is None else l.adj_left) for l in along], []
        cases.append(dict(ids=ids, adj=[-1 if adj[i] is None else adj[i] for i in ids],
                          inters=[[[sorted(e.incoming_lanelets), sorted(e.successors_left), sorted(e.successors_right),
                                    sorted(e.successors_straight)] for e in it.incomings] for it in inters],
                          at_pts=[at[tuple(q)] for q in pts], at_ego=at[tuple(ego)], found=found, rel=rel, inner=inner_ids))
    # flat arrays (no pickles): every list of lists as values + offsets
    out = {"n_cases": len(cases)}
    for k, cs in enumerate(cases):
        out[f"c{k}_ids"] = np.array(cs["ids"]); out[f"c{k}_adj"] = np.array(cs["adj"])
        flat, meta = [], []          # meta rows: intersection, incoming element, kind (0 incoming 1 left 2 right 3 straight), count
        for a, it in enumerate(cs["inters"]):
            for b, el in enumerate(it):
                for kind, lst in enumerate(el):
                    meta.append((a, b, kind, len(lst))); flat.extend(lst)
        out[f"c{k}_inter_meta"] = np.array(meta, dtype=np.int64).reshape(-1, 4); out[f"c{k}_inter_flat"] = np.array(flat, dtype=np.int64)
        out[f"c{k}_n_inter"] = len(cs["inters"])
        out[f"c{k}_at_off"] = np.cumsum([0] + [len(a) for a in cs["at_pts"]]); out[f"c{k}_at_flat"] = np.array([x for a in cs["at_pts"] for x in a], dtype=np.int64)
        out[f"c{k}_at_ego"] = np.array(cs["at_ego"], dtype=np.int64)
        out[f"c{k}_found"] = cs["found"]; out[f"c{k}_rel"] = np.array(cs["rel"], dtype=np.int64); out[f"c{k}_inner"] = np.array(cs["inner"], dtype=np.int64)
    # nearest index and the intention thresholds
    path_s = np.cumsum(np.concatenate(([0.0], rng.uniform(0.05, 1.5, 300))))
    q_s = np.concatenate((rng.uniform(-5.0, path_s[-1] + 5.0, 200), 0.5 * (path_s[3:40] + path_s[4:41]), path_s[50:60]))   # incl. exact midpoints (ties) and vertices
    out["path_s"] = path_s; out["query_s"] = q_s
    out["nearest"] = np.array([int(SpawnLocator._find_nearest_index(path_s, q)) for q in q_s])
    curvs, names = [], []
    for _ in range(60):
        k = rng.normal(0.0, float(rng.choice([0.02, 0.04, 0.08])), int(rng.integers(3, 30)))
        if rng.random() < 0.3:
            k[int(rng.integers(len(k)))] = float(rng.choice([0.10, -0.10, 0.1000001, -0.1000001]))   # on and next to the thresholds
        curv_box["k"] = k
        curvs.append(k); names.append(("straight ahead", "left turn", "right turn").index(SpawnLocator._find_ego_intention(None)))
    out["curv_off"] = np.cumsum([0] + [len(k) for k in curvs]); out["curv_flat"] = np.concatenate(curvs); out["intention"] = np.array(names)
    # find_spawn_points itself (:80-143) with the three rule methods replaced by recorders: s_threshold (:113), the reference
    # window (:678-693), which rule runs under which intention and switch (:124-138), how their results are appended (:637-645)
    len_box = {}
    sys.modules["frenetix_occlusion.spawn_locator"].compute_pathlength_from_polyline = lambda p: len_box["s"]
    rows, toks, tok_off = [], [], [0]
    for c in range(80):
        n = int(rng.integers(20, 240))
        s_arr = np.cumsum(np.concatenate(([0.0], rng.uniform(0.2, 1.2, n - 1))))
        len_box["s"] = s_arr
        ego_s = float(rng.uniform(-2.0, s_arr[-1] + 2.0))
        ego_v = float(rng.choice([rng.uniform(0.0, 15.0), 6.25, 6.2500001, 0.0]))
        curv_box["k"] = rng.normal(0.0, float(rng.choice([0.02, 0.05, 0.09])), 12)
        sw = [bool(x) for x in rng.random(3) < 0.75]
        kind = [int(rng.integers(4)), int(rng.integers(4)), int(rng.integers(2))]
        ret = lambda base, k_: None if k_ == 0 else [] if k_ == 1 else [base] if k_ == 2 else [base, None, base + 1]
        calls = []
        sl = object.__new__(SpawnLocator)
        sl.spawn_points, sl.ref_path, sl.debug = [], np.stack((np.arange(n, dtype=np.float64), np.zeros(n)), -1), False
        sl.s_threshold_time, sl.min_s_threshold = 4, 25
        sl.spawn_point_behind_dynamic_obstacle, sl.spawn_point_behind_static_obstacle, sl.spawn_point_behind_turn = sw
        sl._find_spawn_point_behind_dynamic_obstacle = lambda: (calls.append(0), ret(100, kind[0]))[1]
        sl._find_spawn_point_behind_static_obstacle = lambda: (calls.append(1), ret(200, kind[1]))[1]
        sl._find_spawn_point_behind_turn = lambda intention: (calls.append(2), None if kind[2] == 0 else 300)[1]
        got = sl.find_spawn_points(np.array([0.0, 0.0]), 0.0, np.array([ego_s, 0.3]), ego_v)
        i0 = int(sl.reference[0][0]) if len(sl.reference) else -1
        inten = ("straight ahead", "left turn", "right turn").index(SpawnLocator._find_ego_intention(None))
        rows.append([n, ego_s, ego_v, *[float(x) for x in sw], *[float(x) for x in kind], sl.s_threshold, i0, len(sl.reference), inten,
                     float(0 in calls), float(1 in calls), float(2 in calls)])
        out[f"o{c}_s"] = s_arr; out[f"o{c}_k"] = curv_box["k"]
        toks.extend(int(t) for t in got); tok_off.append(len(toks))
    out["orch"] = np.array(rows); out["orch_tokens"] = np.array(toks, dtype=np.int64); out["orch_tok_off"] = np.array(tok_off)
    path = os.path.join(OUT, "relevant_lanelets.npz")
    np.savez_compressed(path, **out)
    print("orchestration cases", len(rows), "calls dyn / static / turn", [int(sum(r[-3 + q] for r in rows)) for q in range(3)],
          "windows of fewer than three vertices", sum(r[11] < 3 for r in rows))
    print("wrote", path, "cases", len(cases), "with an intersection", sum(c["found"] >= 0 for c in cases),
          "ego on no lanelet", sum(not c["at_ego"] for c in cases), "intentions", np.bincount(out["intention"]).tolist())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "relevant":     # only the relevant-lanelets fixture (own process: stub modules)
        gen_relevant_lanelets()
    elif len(sys.argv) > 1 and sys.argv[1] == "sampling":     # only the sampling-matrix fixture
        gen_sampling_matrix()
    elif len(sys.argv) > 1 and sys.argv[1] == "be":         # only the brake-evaluation fixture (own process: it replaces a module)
        gen_be()
    elif len(sys.argv) > 1 and sys.argv[1] == "dce":        # only the DCE-loop fixture (own process as well)
        gen_dce()
    elif len(sys.argv) > 1 and sys.argv[1] == "shadow":     # only the shadow-geometry fixture
        gen_shadow_geometry()
    else:
        main()
        gen_sampling_matrix()
        # (the shadow-geometry fixture needs its own process: its recording Polygon stub would shadow the inert stubs above)
        import subprocess
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "shadow"])
