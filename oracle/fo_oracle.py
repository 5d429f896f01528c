"""ctypes front-end of the CPU oracle (oracle/fo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (frenetix-occlusion_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("FO_ORACLE_LIB") or os.path.join(_HERE, "libfo_oracle.so")   # env: sanitizer build (make asan)

NPF, NPI, NL, NC = 12, 4, 5, 16
PF = {"dce": 0, "ttc": 1, "ttce": 2, "max_ego_risk": 3, "max_obst_risk": 4, "max_obst_harm_with_cp": 5,
      "max_ego_harm": 6, "max_obst_harm": 7, "max_collision_probability": 8, "be_decel": 9, "be_btn": 10}
PI = {"time_dce": 0, "max_obst_risk_index": 1, "cp_argmax": 2, "hr_valid": 3}
LST = {"cp": 0, "ego_harm": 1, "obst_harm": 2, "ego_risk": 3, "obst_risk": 4}
COST = {"wttc": 0, "min_dce": 1, "max_ego_risk_all": 2, "max_obst_risk_all": 3, "max_ego_harm_all": 4,
        "max_obst_harm_all": 5, "max_collision_probability_all": 6, "max_obst_harm_with_cp_all": 7, "min_ttce": 8,
        "argmin_dce": 9, "argmin_ttc": 10, "argmax_risk": 11, "safe": 12, "max_btn": 13}
METRIC_BITS = {"dce": 1, "cp": 2, "ttc": 4, "ttce": 8, "wttc": 16, "be": 32, "hr": 64}
TYPE_CODES = {"car": 0, "truck": 1, "bus": 2, "bicycle": 3, "pedestrian": 4, "priorityvehicle": 5,
              "parkedvehicle": 6, "train": 7, "motorcycle": 8, "taxi": 9, "unknown": 10}

# harm_params.json entries that the reference reads (harm_params.json:26-29,40-45,98-101)
HARM_COEFF = dict(lr4s_const=-4.457, lr4s_speed=0.177, lr4s_side=0.244, lr4s_rear=-0.431,
                  lr1s_const=-4.591, lr1s_speed=0.185, ped_const=3.164, ped_speed=0.288)


class Vehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("length", "width", "wb_rear_axle", "mass", "a_max")]


class HarmCoeff(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lr4s_const", "lr4s_speed", "lr4s_side", "lr4s_rear", "lr1s_const",
                                          "lr1s_speed", "ped_const", "ped_speed")]


class Thresholds(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("harm", "risk", "be", "cp", "ttc", "dce")]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("fo_oracle.c", "fo_oracle_scene.c", "fo_oracle.h")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.fo_oracle_quad_distance.restype = C.c_double
        _lib.fo_oracle_round3.restype = C.c_double
        _lib.fo_oracle_round3.argtypes = [C.c_double]
        _lib.fo_oracle_box_prob.restype = C.c_double
        _lib.fo_oracle_required_metrics.restype = C.c_uint32
        _lib.fo_oracle_required_metrics.argtypes = [C.c_uint32]
        _lib.fo_oracle_sweep.restype = C.c_int
    return _lib


def _p(arr, typ=C.c_double):
    return arr.ctypes.data_as(C.POINTER(typ)) if arr is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def metric_mask(names):
    m = 0
    for n in names:
        m |= METRIC_BITS[n]
    return m


def thresholds(d=None):
    d = d or {}
    return Thresholds(*[float("nan") if d.get(k) is None else float(d[k]) for k in ("harm", "risk", "be", "cp", "ttc", "dce")])


def quad_distance(qa, qb):
    qa, qb = _f64(qa).reshape(8), _f64(qb).reshape(8)
    return lib().fo_oracle_quad_distance(_p(qa), _p(qb))


def rect_vertices(cx, cy, yaw, length, width):
    q = np.zeros(8)
    lib().fo_oracle_rect_vertices(C.c_double(cx), C.c_double(cy), C.c_double(yaw), C.c_double(length),
                                  C.c_double(width), _p(q))
    return q.reshape(4, 2)


def box_prob(lo, hi, mu, sxx, syy):
    lo, hi, mu = _f64(lo), _f64(hi), _f64(mu)
    return lib().fo_oracle_box_prob(_p(lo), _p(hi), _p(mu), C.c_double(sxx), C.c_double(syy))


def sweep(traj, agents, vehicle, dt, metrics=("hr", "ttc", "ttce", "dce", "wttc", "cp"), thr=None,
          harm_coeff=None, want_lists=True, nthreads=1, out=None):
    """traj: dict x,y,theta,v,a [M,T]; agents: dict pos[A,Ta,2], yaw[A,Ta], v[A,Ta], cov[A,Ta,2,2], shape[A,2],
    raw_dims[A,2], type[A] int, len[A] int; vehicle: (length,width,wb_rear_axle,mass,a_max).
    Returns dict of numpy arrays in the oracle's [M,A,...] layout."""
    x, y, th, v = (_f64(traj[k]) for k in ("x", "y", "theta", "v"))
    a = _f64(traj.get("a", np.zeros_like(x)))
    M, T = x.shape
    pos, yaw, av = _f64(agents["pos"]), _f64(agents["yaw"]), _f64(agents["v"])
    A = pos.shape[0]
    Ta = pos.shape[1] if A else 1
    cov = _f64(agents["cov"]).reshape(A, Ta, 4) if A else np.zeros((0, 1, 4))
    shape, raw = _f64(agents["shape"]), _f64(agents["raw_dims"])
    typ = np.ascontiguousarray(agents["type"], dtype=np.int32)
    ln = np.ascontiguousarray(agents["len"], dtype=np.int32)
    veh = Vehicle(*[float(q) for q in vehicle])
    hc = HarmCoeff(**(harm_coeff or HARM_COEFF))
    th_s = thr if isinstance(thr, Thresholds) else thresholds(thr)
    Tm1 = max(T - 1, 0)
    if out is not None:  # reuse (already page-faulted) buffers of a previous call with the same shapes
        pair_f, pair_i, lists, cost, safe = out["pair_f"], out["pair_i"], out["lists"], out["cost"], out["safe"]
        assert pair_f.shape == (M, A, NPF) and cost.shape == (M, NC)
    else:
        pair_f = np.empty((M, A, NPF))
        pair_i = np.empty((M, A, NPI), dtype=np.int32)
        lists = np.empty((M, A, NL, Tm1)) if want_lists else None
        cost = np.empty((M, NC))
        safe = np.empty(M, dtype=np.uint8)
    rc = lib().fo_oracle_sweep(
        C.c_int(M), C.c_int(T), _p(x), _p(y), _p(th), _p(v), _p(a), C.c_int(A), C.c_int(Ta), _p(pos), _p(yaw),
        _p(av), _p(cov), _p(shape), _p(raw), _p(typ, C.c_int32), _p(ln, C.c_int32), C.byref(veh), C.byref(hc),
        C.c_double(dt), C.byref(th_s), C.c_uint32(metric_mask(metrics)), _p(pair_f), _p(pair_i, C.c_int32),
        _p(lists), _p(cost), _p(safe, C.c_uint8), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f"fo_oracle_sweep failed with code {rc}")
    return {"pair_f": pair_f, "pair_i": pair_i, "lists": lists, "cost": cost, "safe": safe}


# ---------------------------------------------------------------------------------------------- scene half
# (fo_oracle_scene.c; discretisation defined in DESIGN.md, "parity unpinned" against the reference's GEOS algebra)
def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def ray_dirs(n_rays, ego_yaw=0.0, fov_deg=360.0):
    """Unit ray directions, counter-clockwise.  Full circle: angle_i = yaw + 2 pi i / n (ray n == ray 0);
    open fan: n rays from yaw - fov/2 to yaw + fov/2 inclusive.  Shared definition (host layer uses the same)."""
    if fov_deg >= 359.9:
        ang = ego_yaw + 2.0 * np.pi * np.arange(n_rays) / n_rays
    else:
        half = np.radians(fov_deg) / 2.0
        ang = ego_yaw + np.linspace(-half, half, n_rays)
    return np.stack((np.cos(ang), np.sin(ang)), -1)


def road_raster(poly_off, poly_xy, x0, y0, cs, nx, ny):
    poly_off, poly_xy = _i32(poly_off), _f64(poly_xy)
    mask = np.zeros((ny, nx), dtype=np.uint8)
    lib().fo_oracle_road_raster(C.c_int(len(poly_off) - 1), _p(poly_off, C.c_int32), _p(poly_xy), C.c_double(x0),
                                C.c_double(y0), C.c_double(cs), C.c_int(nx), C.c_int(ny), _p(mask, C.c_uint8))
    return mask


def _opt(a, typ, conv):
    return _p(conv(a), typ) if a is not None else None


def raycast(edges, ocorn, oflags, ego, dirs, r, rmax=None, edge_skip=None):
    edges, ocorn, oflags = _f64(edges).reshape(-1, 4), _f64(ocorn).reshape(-1, 8), _u8(oflags)
    ego, dirs = _f64(ego), _f64(dirs)
    n = dirs.shape[0]
    rng, hid, ring = np.zeros(n), np.zeros(n, dtype=np.int32), np.zeros((n, 2))
    rmax = _f64(rmax) if rmax is not None else None
    edge_skip = _u8(edge_skip) if edge_skip is not None else None
    lib().fo_oracle_raycast(C.c_int(edges.shape[0]), _p(edges), _opt(edge_skip, C.c_uint8, _u8),
                            C.c_int(ocorn.shape[0]), _p(ocorn), _p(oflags, C.c_uint8), _p(ego), C.c_int(n), _p(dirs),
                            C.c_double(r), _opt(rmax, C.c_double, _f64), _p(rng), _p(hid, C.c_int32), _p(ring))
    return rng, hid, ring


class _Exact(C.Structure):
    _fields_ = [("hit_id", C.c_void_p), ("rmax", C.c_void_p), ("E", C.c_int), ("edges", C.c_void_p),
                ("edge_skip", C.c_void_p), ("O", C.c_int), ("ocorn", C.c_void_p), ("oflags", C.c_void_p),
                ("half_dirs", C.c_void_p), ("edge_line", C.c_void_p), ("shadow_length", C.c_double)]


def grid(raster, rx0, ry0, cs, ix0, iy0, nx, ny, ego, hdir, r, full, dirs, rng, exact=None, return_n_exact=False):
    """exact: None (fan rule only) or dict(hit_id=, edges=, ocorn=, oflags=, rmax=None, edge_skip=None,
    half_dirs=None, edge_line=None, shadow_length=100.0): shadow_length is where an obstacle's occlusion polygon ends
    (helper_functions.py:145-146); math.inf = the wedge never ends"""
    raster = _u8(raster)
    rny, rnx = raster.shape
    ego, hdir, dirs, rng = _f64(ego), _f64(hdir), _f64(dirs), _f64(rng)
    cls = np.zeros((ny, nx), dtype=np.uint8)
    occ = np.zeros(nx * ny, dtype=np.int32)
    n_occ, n_ex = C.c_int32(0), C.c_int32(0)
    ex, keep = None, []
    if exact is not None:
        hid = _i32(exact["hit_id"])
        edges = _f64(exact["edges"]).reshape(-1, 4)
        ocorn = _f64(exact["ocorn"]).reshape(-1, 8)
        oflags = _u8(exact["oflags"])
        rmax = _f64(exact["rmax"]) if exact.get("rmax") is not None else None
        skip = _u8(exact["edge_skip"]) if exact.get("edge_skip") is not None else None
        half = _f64(exact["half_dirs"]) if exact.get("half_dirs") is not None else None
        assert half is None or half.shape == (100, 2)
        line = _i32(exact["edge_line"]) if exact.get("edge_line") is not None else None
        keep = [hid, edges, ocorn, oflags, rmax, skip, half, line]
        adr = lambda a: a.ctypes.data if a is not None and a.size else None
        ex = _Exact(adr(hid), adr(rmax), edges.shape[0], adr(edges), adr(skip), ocorn.shape[0], adr(ocorn), adr(oflags),
                     adr(half), adr(line), float(exact.get("shadow_length", 100.0)))
    lib().fo_oracle_grid(_p(raster, C.c_uint8), C.c_int(rnx), C.c_int(rny), C.c_double(rx0), C.c_double(ry0),
                         C.c_double(cs), C.c_int(ix0), C.c_int(iy0), C.c_int(nx), C.c_int(ny), _p(ego), _p(hdir),
                         C.c_double(r), C.c_int(1 if full else 0), C.c_int(dirs.shape[0]), _p(dirs), _p(rng),
                         _p(cls, C.c_uint8), _p(occ, C.c_int32), C.byref(n_occ),
                         C.byref(ex) if ex is not None else None, C.byref(n_ex))
    del keep
    if return_n_exact:
        return cls, occ[:n_occ.value].copy(), int(n_ex.value)
    return cls, occ[:n_occ.value].copy()


def wedge_far(ego, corners, length=100.0):
    """silhouette corner pair of a rectangle seen from ego and the far-chord half-plane of its occlusion polygon
    (helper_functions.py:139-176): returns (c1, c2, abc or None)"""
    ego, corners = _f64(ego), _f64(corners).reshape(4, 2)
    c12, abc = np.zeros(4), np.zeros(3)
    lib().fo_oracle_wedge_far.restype = C.c_int
    ok = lib().fo_oracle_wedge_far(_p(ego), _p(corners), C.c_double(length), _p(c12), _p(abc))
    return c12[:2].copy(), c12[2:].copy(), (abc if ok else None)


def obstacle_visibility(edges, ocorn, ocen, oflags, ego, r, full, dirs, edge_skip=None, hit_id=None):
    edges, ocorn, ocen, oflags = _f64(edges).reshape(-1, 4), _f64(ocorn).reshape(-1, 8), _f64(ocen), _u8(oflags)
    ego, dirs = _f64(ego), _f64(dirs)
    O = ocorn.shape[0]
    vis = np.zeros(O, dtype=np.uint8)
    edge_skip = _u8(edge_skip) if edge_skip is not None else None
    lib().fo_oracle_obstacle_visibility(C.c_int(edges.shape[0]), _p(edges), _opt(edge_skip, C.c_uint8, _u8), C.c_int(O),
                                        _p(ocorn), _p(ocen),
                                        _p(oflags, C.c_uint8), _p(ego), C.c_double(r), C.c_int(1 if full else 0),
                                        C.c_int(dirs.shape[0]), _p(dirs),
                                        _p(_i32(hit_id), C.c_int32) if hit_id is not None else None, _p(vis, C.c_uint8))
    return vis


def future_visibility(x, y, t_stride, dirs, r, edges, ocorn, oflags, occ_idx, rx0, ry0, cs, ix0, iy0, nx):
    """extension: (revealed [M,K] int32, area [M,K]) -- see fo_oracle_scene.c"""
    x, y, dirs = _f64(x), _f64(y), _f64(dirs)
    edges, ocorn, oflags = _f64(edges).reshape(-1, 4), _f64(ocorn).reshape(-1, 8), _u8(oflags)
    occ_idx = _i32(occ_idx)
    M, T = x.shape
    K = (T + t_stride - 1) // t_stride
    rev = np.zeros((M, K), dtype=np.int32)
    area = np.zeros((M, K))
    rc = lib().fo_oracle_future_visibility(C.c_int(M), C.c_int(T), _p(x), _p(y), C.c_int(t_stride), C.c_int(dirs.shape[0]),
                                           _p(dirs), C.c_double(r), C.c_int(edges.shape[0]), _p(edges),
                                           C.c_int(ocorn.shape[0]), _p(ocorn), _p(oflags, C.c_uint8),
                                           C.c_int(len(occ_idx)), _p(occ_idx, C.c_int32), C.c_double(rx0),
                                           C.c_double(ry0), C.c_double(cs), C.c_int(ix0), C.c_int(iy0), C.c_int(nx),
                                           _p(rev, C.c_int32), _p(area))
    assert rc == 0
    return rev, area


def spawn_cells(cls, rx0, ry0, cs, ix0, iy0, ego, hdir, min_ahead, max_dist, max_agents, all_occluded=False):
    cls = _u8(cls)
    ny, nx = cls.shape
    ego, hdir = _f64(ego), _f64(hdir)
    cell = np.full(max_agents, -1, dtype=np.int32)
    pos = np.zeros((max_agents, 2))
    n, nc = C.c_int32(0), C.c_int32(0)
    lib().fo_oracle_spawn_cells(_p(cls, C.c_uint8), C.c_int(nx), C.c_int(ny), C.c_double(rx0), C.c_double(ry0),
                                C.c_double(cs), C.c_int(ix0), C.c_int(iy0), _p(ego), _p(hdir), C.c_double(min_ahead),
                                C.c_double(max_dist), C.c_int(max_agents), C.c_int(1 if all_occluded else 0),
                                _p(cell, C.c_int32), _p(pos), C.byref(n),
                                C.byref(nc))
    return cell, pos, n.value, nc.value


def spawn_headings(pos, types, path, lane_yaw_at=None):
    pos, types, path = _f64(pos), _i32(types), _f64(path)
    n = pos.shape[0]
    lya = _f64(lane_yaw_at) if lane_yaw_at is not None else None
    yaw = np.zeros(n)
    lib().fo_oracle_spawn_headings(C.c_int(n), _p(pos), _p(types, C.c_int32), C.c_int(path.shape[0]), _p(path),
                                   _p(lya), _p(yaw))
    return yaw


def cv_predictions(pos0, yaw, speed, T, dt, var0=0.1, factor=1.05):
    pos0, yaw, speed = _f64(pos0), _f64(yaw), _f64(speed)
    n = pos0.shape[0]
    pos, yl, vl, cov = np.zeros((n, T, 2)), np.zeros((n, T)), np.zeros((n, T)), np.zeros((n, T, 4))
    lib().fo_oracle_cv_predictions(C.c_int(n), _p(pos0), _p(yaw), _p(speed), C.c_int(T), C.c_double(dt),
                                   C.c_double(var0), C.c_double(factor), _p(pos), _p(yl), _p(vl), _p(cov))
    return pos, yl, vl, cov.reshape(n, T, 2, 2)


def route_predictions(pos0, types, speed, lanelet, R, first, count, xy, s, yaw_fallback, T, dt, var0=0.1, factor=1.05):
    pos0, speed, yaw_fallback = _f64(pos0), _f64(speed), _f64(yaw_fallback)
    types, lanelet, first, count = _i32(types), _i32(lanelet), _i32(first), _i32(count)
    xy, s = _f64(xy).reshape(-1, 2), _f64(s)
    n = pos0.shape[0]
    pos, yl, vl = np.zeros((n * R, T, 2)), np.zeros((n * R, T)), np.zeros((n * R, T))
    cov, ln = np.zeros((n * R, T, 4)), np.zeros(n * R, dtype=np.int32)
    lib().fo_oracle_route_predictions(C.c_int(n), _p(pos0), _p(types, C.c_int32), _p(speed), _p(lanelet, C.c_int32),
                                      C.c_int(R), _p(first, C.c_int32), _p(count, C.c_int32), _p(xy), _p(s),
                                      _p(yaw_fallback), C.c_int(T), C.c_double(dt), C.c_double(var0), C.c_double(factor),
                                      _p(pos), _p(yl), _p(vl), _p(cov), _p(ln, C.c_int32))
    return pos, yl, vl, cov.reshape(n * R, T, 2, 2), ln
