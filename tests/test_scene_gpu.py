"""Parity of the HIP scene kernels (ray fan, cell grid, obstacle visibility, phantom spawn + predictions; through the
C ABI) against the CPU restatement in oracle/.  Integer outputs (hit ids, cell classes, occluded-cell indices, spawn
cells) must be bit-exact; ranges are float64 with identical operation order on both sides (no FMA contraction) and are
compared exactly as well.  Needs a real MI355X: `pytest -m gpu`."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    return torch


def _default_config():
    import yaml
    from frenetix_occlusion import interface
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    return cfg


def _check_step(torch, oracle, sc, ego, v_ego, timestep, sensor_angle=360.0, n_rays=720, radius=50.0, max_agents=32,
                ref_path=None, all_occluded=False, max_dist=None, shadow_length=100.0):
    from frenetix_occlusion.sensor_model import SensorModel, ray_dirs
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    ego = np.asarray(ego, dtype=np.float64)
    yaw = float(ego[2])
    if ref_path is None:
        s = np.linspace(0.0, 60.0, 61)
        ref_path = ego[None, :2] + s[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    sm = SensorModel(sc.lanelets, ref_path, sensor_radius=radius, sensor_angle=sensor_angle, n_rays=n_rays,
                     shadow_length=shadow_length)
    obst = FOObstacles(sc.obstacles)
    obst.update(timestep)
    sm.calc_visible_and_occluded_area(timestep, ego[:2], yaw, obst)
    torch.cuda.synchronize()
    geo = sm.map_geometry
    (x0, y0), (rnx, rny) = sm.raster_origin, sm.raster_dims
    cs = sm.cell_size

    # one-off raster
    raster_ref = oracle.road_raster(geo.poly_off, geo.poly_xy, x0, y0, cs, rnx, rny)
    assert np.array_equal(sm.road_raster(), raster_ref)

    # ray fan
    # the fan is written by the device (fo_scene_fan): it must agree with the host statement of its definition to
    # rounding; the oracle is then fed the very same directions so that everything downstream is bit-comparable
    assert np.array_equal(ray_dirs(n_rays, yaw, sensor_angle), oracle.ray_dirs(n_rays, yaw, sensor_angle))
    dirs = sm.dirs.cpu().numpy()
    np.testing.assert_allclose(dirs, ray_dirs(n_rays, yaw, sensor_angle), rtol=0, atol=4e-15)
    corn, cen, flags = obst.arrays()
    from frenetix_occlusion.sensor_model import HoleIndex, footprint_ranges
    rmax = sm.rmax.cpu().numpy()
    np.testing.assert_allclose(rmax, footprint_ranges(n_rays, yaw, sensor_angle, radius), rtol=0, atol=1e-11)
    hi = HoleIndex(geo)
    rings = hi.enclosed(ego[:2], yaw, sensor_angle, radius)
    skip = hi.edge_skip(rings) if rings else None
    assert (sm.edge_skip is None) == (skip is None)
    if skip is not None:
        assert np.array_equal(sm.edge_skip.cpu().numpy(), skip)
    rng_ref, hid_ref, ring_ref = oracle.raycast(geo.edges, corn, flags, ego[:2], dirs, radius, rmax=rmax, edge_skip=skip)
    assert np.array_equal(sm.hit_id.cpu().numpy(), hid_ref)
    assert np.array_equal(sm.range.cpu().numpy(), rng_ref)
    assert np.array_equal(sm.visible_area.ring.cpu().numpy(), ring_ref)

    # cell classes + occluded-cell indices (bit-exact, north_star)
    w = sm.window
    full = sensor_angle >= 359.9
    hd = np.array([math.cos(yaw), math.sin(yaw)])
    from frenetix_occlusion.sensor_model import half_fan_dirs
    half = sm.half_dirs.cpu().numpy()
    np.testing.assert_allclose(half, half_fan_dirs(yaw), rtol=0, atol=4e-15)
    ex = dict(hit_id=hid_ref, edges=geo.edges, ocorn=corn, oflags=flags, rmax=rmax, edge_skip=skip, half_dirs=half, edge_line=geo.edge_line,
              shadow_length=shadow_length)
    cls_ref, occ_ref, n_exact = oracle.grid(raster_ref, x0, y0, cs, w.ix0, w.iy0, w.nx, w.ny, ego[:2], hd, radius, full,
                                            dirs, rng_ref, exact=ex, return_n_exact=True)
    assert np.array_equal(sm.cell_class.cpu().numpy(), cls_ref)
    assert np.array_equal(sm.occluded_cells().cpu().numpy(), occ_ref)

    # obstacle visibility
    if len(flags):
        vis_ref = oracle.obstacle_visibility(geo.edges, corn, cen, flags, ego[:2], radius, full, dirs, edge_skip=skip,
                                             hit_id=hid_ref)
        got = np.array([o.current_visible for o in obst], dtype=np.uint8)
        assert np.array_equal(got, vis_ref)
        assert sm.visible_objects_timestep == [o.obstacle_id for o, v in zip(obst, vis_ref) if v]

    # phantom sampling + predictions
    cfg = _default_config()
    cfg["accelerator"]["spawn"]["max_agents"] = max_agents
    cfg["accelerator"]["spawn"]["all_occluded"] = all_occluded
    cfg["accelerator"]["spawn"]["max_dist"] = max_dist
    sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1)
    pts = sl.find_spawn_points(ego[:2], yaw, None, v_ego)
    torch.cuda.synchronize()
    b = sl.batch
    cell_ref, pos_ref, n_ref, n_cand = oracle.spawn_cells(cls_ref, x0, y0, cs, w.ix0, w.iy0, ego[:2], hd, sl.min_ahead,
                                                          sl.max_distance(v_ego), max_agents, all_occluded)
    assert int(b.n.item()) == n_ref == len(pts)
    assert np.array_equal(b.cell.cpu().numpy(), cell_ref)
    assert np.array_equal(b.pos0.cpu().numpy()[:n_ref], pos_ref[:n_ref])
    types = np.array([sl._t4[j % 4] for j in range(n_ref)], dtype=np.int32)
    assert np.array_equal(b.type.cpu().numpy()[:n_ref], types)
    ln = b.len.cpu().numpy()
    assert (ln[:n_ref] == sl.T).all() and (ln[n_ref:] == 0).all()
    if n_ref:
        lya = None
        if sm.lane_yaw is not None:
            ci = cell_ref[:n_ref]
            wx, wy = w.ix0 + ci % w.nx, w.iy0 + ci // w.nx
            ok = (wx >= 0) & (wx < rnx) & (wy >= 0) & (wy < rny)
            lya = np.where(ok, sm.lane_yaw[np.clip(wy, 0, rny - 1), np.clip(wx, 0, rnx - 1)], np.nan)
        yaw_ref = oracle.spawn_headings(pos_ref[:n_ref], types, ref_path, lya)
        np.testing.assert_allclose(b.yaw0.cpu().numpy()[:n_ref], yaw_ref, rtol=0, atol=1e-12)
        speed = np.array([sl._s4[j % 4] for j in range(n_ref)])
        p, yl, vl, cov = oracle.cv_predictions(pos_ref[:n_ref], yaw_ref, speed, sl.T, 0.1, 0.1, sl.var_factor)
        np.testing.assert_allclose(b.pos.cpu().numpy()[:n_ref], p, rtol=0, atol=1e-9)
        np.testing.assert_allclose(b.yaw.cpu().numpy()[:n_ref], yl, rtol=0, atol=1e-12)
        assert np.array_equal(b.v.cpu().numpy()[:n_ref], vl)
        np.testing.assert_allclose(b.cov.cpu().numpy()[:n_ref], cov, rtol=1e-13, atol=0)
    return dict(n_spawn=n_ref, n_cand=n_cand, n_occ=len(occ_ref), vis_cells=int(((cls_ref & 2) != 0).sum()),
                skipped=0 if skip is None else int(skip.sum()), n_exact=n_exact)


@pytest.mark.parametrize("timestep", [0, 8, 25, 60])
def test_scenario1_steps_match_the_oracle(torch_cuda, oracle, timestep):
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial.copy()
    ego[0] += 0.7 * timestep * math.cos(ego[2])      # the ego advances ~7 m/s along its heading
    ego[1] += 0.7 * timestep * math.sin(ego[2])
    st = _check_step(torch_cuda, oracle, sc, ego, 7.63, timestep)
    assert st["vis_cells"] > 100 and st["n_occ"] > 0
    assert st["n_exact"] > 20           # cells settled by the exact rule (shadow edges between two rays)
    if timestep == 0:
        assert st["skipped"] == 7      # the sliver hole behind the ego lies inside the footprint: no shadow (Q9)


@pytest.mark.parametrize("k", [2, 3])
def test_scenario2_and_3_match_the_oracle(torch_cuda, oracle, k):
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, f"scenario{k}_geometry.npz"))
    st = _check_step(torch_cuda, oracle, sc, sc.ego_initial, float(sc.ego_initial[3]), 0)
    assert st["vis_cells"] > 100


def test_open_fan_and_odd_ray_counts(torch_cuda, oracle):
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    _check_step(torch_cuda, oracle, sc, sc.ego_initial, 7.63, 0, sensor_angle=90.0, n_rays=181, radius=30.0)
    _check_step(torch_cuda, oracle, sc, sc.ego_initial, 7.63, 3, sensor_angle=200.0, n_rays=400, radius=40.0)
    _check_step(torch_cuda, oracle, sc, sc.ego_initial, 7.63, 3, sensor_angle=360.0, n_rays=97, radius=25.0, max_agents=5)


def test_synthetic_urban_grid_config3(torch_cuda, oracle):
    """BASELINE configs[2] scene: O(10^4) boundary edges, 64 parked cars, 720 rays @ 0.5 deg, 256 phantom slots."""
    from frenetix_occlusion import scenario as S
    sc = S.synthetic_urban_grid()
    st = _check_step(torch_cuda, oracle, sc, sc.ego_initial, 8.0, 0, max_agents=256)
    assert st["n_occ"] > 500 and st["n_cand"] > 0
    # the bench configuration: every occluded cell within 45 m is a candidate -> all 256 phantom slots are filled
    st = _check_step(torch_cuda, oracle, sc, sc.ego_initial, 8.0, 0, max_agents=256, all_occluded=True, max_dist=45.0)
    assert st["n_spawn"] == 256 and st["n_cand"] > 256


def test_no_obstacles_and_ego_off_the_raster_edge(torch_cuda, oracle):
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario2_geometry.npz"))
    sc.obstacles = []
    ego = sc.ego_initial.copy()
    _check_step(torch_cuda, oracle, sc, ego, 5.0, 0)
    ego[:2] = sc.lanelets[0].center[0]      # window hangs over the raster border
    _check_step(torch_cuda, oracle, sc, ego, 5.0, 0)


def test_map_without_obstacles_or_with_everything_absent(torch_cuda, oracle):
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    # time step far beyond every obstacle's state list: all flags 0
    _check_step(torch_cuda, oracle, sc, sc.ego_initial, 7.63, 10_000)
    # a bicycle in front of the ego must not cast a shadow (Q10) but is itself visible
    bike = S.Obstacle(555, "static", "bicycle", 2.0, 0.9, 0, np.array([8.0, 0.0, 0.0, 0.0]), np.zeros((0, 4)))
    sc.obstacles = [bike]
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    sm = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0)
    ob = FOObstacles(sc.obstacles)
    ob.update(0)
    sm.calc_visible_and_occluded_area(0, sc.ego_initial[:2], float(sc.ego_initial[2]), ob)
    E = len(sm.map_geometry.edges)
    assert (sm.hit_id.cpu().numpy() < E).all() and sm.visible_objects_timestep == [555]
    _check_step(torch_cuda, oracle, sc, sc.ego_initial, 7.63, 0)


def test_obstacle_lit_between_its_probe_points_is_visible(torch_cuda, oracle):
    """corners and centre of a long obstacle hidden behind three blockers, its side lit through the gaps: visible
    because fan rays stop at it (tests/test_scene_kat.py has the same scene on the oracle alone)"""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    xs = np.linspace(-5.0, 40.0, 2)
    room = S.Lanelet(1, np.stack((xs, np.full(2, 15.0)), -1), np.stack((xs, np.full(2, -15.0)), -1))
    mk = lambda i, x, y, l, w: S.Obstacle(i, "static", "car", l, w, 0, np.array([x, y, 0.0, 0.0]), np.zeros((0, 4)))
    sc = S.Scenario(0.1, [room], [mk(1, 30, 0, 2.0, 20.0), mk(2, 10, -3.3, 1.0, 1.2), mk(3, 10, 0.0, 1.0, 1.0),
                                  mk(4, 10, 3.3, 1.0, 1.2)])
    ego = np.array([0.0, 0.0, 0.0, 5.0])
    _check_step(torch_cuda, oracle, sc, ego, 5.0, 0)
    sm = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0)
    ob = FOObstacles(sc.obstacles)
    ob.update(0)
    sm.calc_visible_and_occluded_area(0, ego[:2], 0.0, ob)
    assert sm.visible_objects_timestep == [1, 2, 3, 4]
    assert len(sm.obstacle_occlusions[1]) > 10          # the rays its side stops


def test_shadow_of_an_obstacle_ends_where_the_references_polygon_does(torch_cuda, oracle):
    """helper_functions.py:145-146 (the occlusion polygon ends 100 m along the two silhouette sight lines): a truck across
    the road ~1 m ahead of the ego subtends so wide an angle that the far chord passes ~18 m away; the road beyond it is
    visible in the reference (tests/test_scene_pointwise.py proves the oracle equal to the reference's set algebra there).
    Device == oracle bit for bit with the default length, with other lengths and with the physical shadow (inf), on
    scenario 1 and in an open room; plus several obstacles at once, one of them a bicycle (casts no shadow)."""
    from frenetix_occlusion import scenario as S
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    yaw = float(ego[2])
    fwd, left = np.array([math.cos(yaw), math.sin(yaw)]), np.array([-math.sin(yaw), math.cos(yaw)])
    seen = {}
    for gap, shift in ((1.0, 0.0), (0.6, 2.0)):
        cen = ego[:2] + (gap + 1.25) * fwd + shift * left
        truck = S.Obstacle(900, "static", "truck", 12.0, 2.5, 0, np.array([cen[0], cen[1], yaw + 0.5 * math.pi, 0.0]),
                           np.zeros((0, 4)))
        sc2 = S.Scenario(sc.dt, sc.lanelets, [truck] + list(sc.obstacles), sc.intersections, sc.ego_initial, sc.benchmark_id)
        for L in (100.0, 30.0, math.inf):
            seen[(gap, L)] = _check_step(torch_cuda, oracle, sc2, ego, 7.63, 0, shadow_length=L)["vis_cells"]
        assert seen[(gap, 30.0)] > seen[(gap, 100.0)] > seen[(gap, math.inf)]
    xs = np.linspace(-60.0, 60.0, 2)
    room = S.Lanelet(1, np.stack((xs, np.full(2, 60.0)), -1), np.stack((xs, np.full(2, -60.0)), -1))
    mk = lambda i, x, y, l, w, yw, typ="car": S.Obstacle(i, "static", typ, l, w, 0, np.array([x, y, yw, 0.0]), np.zeros((0, 4)))
    sc3 = S.Scenario(0.1, [room], [mk(1, 2.0, 0.3, 2.5, 14.0, 0.1), mk(2, -1.5, 0.0, 2.0, 9.0, -0.2),
                                   mk(3, 0.5, 2.2, 6.0, 1.0, 0.0, "bicycle"), mk(4, 0.0, -9.0, 4.5, 1.8, 0.4)])
    e3 = np.array([0.0, 0.0, 0.0, 5.0])
    a = _check_step(torch_cuda, oracle, sc3, e3, 5.0, 0)["vis_cells"]
    b = _check_step(torch_cuda, oracle, sc3, e3, 5.0, 0, shadow_length=math.inf)["vis_cells"]
    assert a > b + 200


def test_shadow_length_argument_checks(torch_cuda):
    import math
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.sensor_model import SensorModel
    ctx = N.Context(0)
    with pytest.raises(Exception, match="NaN"):
        ctx.call("fo_scene_set_shadow_length", float("nan"))
    ctx.call("fo_scene_set_shadow_length", math.inf)
    ctx.call("fo_scene_set_shadow_length", 100.0)
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    for bad in (0.0, -5.0, float("nan")):
        with pytest.raises(ValueError):
            SensorModel(sc.lanelets, None, shadow_length=bad)


def test_footprint_and_hole_options(torch_cuda, oracle):
    """footprint="circle" / enclosed_holes="occlude" restore the exact-radius, every-piece-occludes variant"""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.sensor_model import SensorModel, ray_dirs
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    sm = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0, footprint="circle",
                     enclosed_holes="occlude", cell_visibility="fan")
    sm.calc_visible_and_occluded_area(0, ego[:2], float(ego[2]), None)
    assert sm.rmax is None and sm.edge_skip is None and sm.half_dirs is None
    dirs = sm.dirs.cpu().numpy()
    np.testing.assert_allclose(dirs, ray_dirs(720, float(ego[2]), 360.0), rtol=0, atol=4e-15)
    rng_ref, hid_ref, _ = oracle.raycast(sm.map_geometry.edges, np.zeros((0, 8)), np.zeros(0, np.uint8), ego[:2], dirs, 50.0)
    assert np.array_equal(sm.range.cpu().numpy(), rng_ref) and np.array_equal(sm.hit_id.cpu().numpy(), hid_ref)
    w = sm.window
    (x0, y0), (rnx, rny) = sm.raster_origin, sm.raster_dims
    raster = oracle.road_raster(sm.map_geometry.poly_off, sm.map_geometry.poly_xy, x0, y0, sm.cell_size, rnx, rny)
    hd = np.array([math.cos(ego[2]), math.sin(ego[2])])
    cls_ref, occ_ref = oracle.grid(raster, x0, y0, sm.cell_size, w.ix0, w.iy0, w.nx, w.ny, ego[:2], hd, 50.0, True, dirs,
                                   rng_ref)
    assert np.array_equal(sm.cell_class.cpu().numpy(), cls_ref)
    assert np.array_equal(sm.occluded_cells().cpu().numpy(), occ_ref)
    # with the reference semantics more of the road behind the ego is visible (the sliver hole no longer blocks it)
    vis_all = int(((sm.cell_class & 2) != 0).sum().item())
    sm2 = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0)
    sm2.calc_visible_and_occluded_area(0, ego[:2], float(ego[2]), None)
    assert int(((sm2.cell_class & 2) != 0).sum().item()) > vis_all + 100
    with pytest.raises(ValueError):
        SensorModel(sc.lanelets, None, footprint="square")


def test_phantom_vehicle_predictions_follow_lanelet_routes(torch_cuda, oracle):
    """routes > 0: one prediction per candidate route for vehicles (slot j * R + r), compared with the oracle"""
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    for name, R in (("scenario3_geometry.npz", 3), ("scenario1_geometry.npz", 2)):
        sc = S.load_geometry_npz(os.path.join(GOLDEN, name))
        ego = sc.ego_initial
        yaw = float(ego[2])
        ref_path = ego[None, :2] + np.linspace(0.0, 60.0, 61)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
        sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, routes=R)
        obst = FOObstacles(sc.obstacles)
        obst.update(0)
        sm.calc_visible_and_occluded_area(0, ego[:2], yaw, obst)
        cfg = _default_config()
        cfg["accelerator"]["spawn"].update(max_agents=24, all_occluded=True, max_dist=45.0, routes=R,
                                           pattern=["Car", "Bicycle", "Pedestrian", "Car"])
        sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1)
        assert sl.R == R
        sl.find_spawn_points(ego[:2], yaw, None, float(ego[3]))
        torch_cuda.cuda.synchronize()
        b = sl.batch
        n = int(b.n.item())
        assert n > 4 and b.pos.shape[0] == 24 * R
        w = sm.window
        ci = b.cell.cpu().numpy()[:n]
        wx, wy = w.ix0 + ci % w.nx, w.iy0 + ci // w.nx
        lan = sm.lanelet_raster[wy, wx]
        types = np.array([sl._t4[j % 4] for j in range(n)], dtype=np.int32)
        speed = np.array([sl._s4[j % 4] for j in range(n)])
        tab = sm.route_table
        pos, yl, vl, cov, ln = oracle.route_predictions(b.pos0.cpu().numpy()[:n], types, speed, lan, R, tab.first,
                                                        tab.count, tab.xy, tab.s, b.yaw0.cpu().numpy()[:n], sl.T, 0.1,
                                                        0.1, sl.var_factor)
        got_len = b.len.cpu().numpy()
        assert np.array_equal(got_len[:n * R], ln) and (got_len[n * R:] == 0).all()
        np.testing.assert_allclose(b.pos.cpu().numpy()[:n * R], pos, rtol=0, atol=1e-9)
        np.testing.assert_allclose(b.yaw.cpu().numpy()[:n * R], yl, rtol=0, atol=1e-12)
        np.testing.assert_allclose(b.v.cpu().numpy()[:n * R], vl, rtol=0, atol=1e-12)
        np.testing.assert_allclose(b.cov.cpu().numpy()[:n * R], cov, rtol=1e-13, atol=0)
        assert np.array_equal(b.type.cpu().numpy()[:n * R], np.repeat(types, R))
        veh_routes = ln.reshape(n, R)[types != 4]
        assert (veh_routes[:, 0] > 0).all() and (veh_routes > 0).sum() > len(veh_routes)    # some vehicle has 2 routes
        assert (ln.reshape(n, R)[types == 4][:, 1:] == 0).all()


def test_static_map_shared_between_contexts(torch_cuda):
    """BASELINE configs[4] ('shared occlusion map in HBM'): a second ego's SensorModel reads the first one's static map
    by reference (fo_scene_share_map).  Same outputs as with a map of its own, at its own pose, also after the owner
    is gone; a context on which fo_scene_set_map is called again detaches without disturbing the other."""
    import gc
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import scenario as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    torch = torch_cuda
    sc = S.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    poses = [(ego[0], ego[1], float(ego[2])), (ego[0] + 6.0, ego[1] - 0.3, float(ego[2]) + 0.2)]
    ref = ego[None, :2] + np.linspace(0.0, 60.0, 61)[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
    cfg = _default_config()
    cfg["accelerator"]["spawn"].update(max_agents=16, routes=3)

    def outputs(sm, pose):
        ob = FOObstacles(sc.obstacles)
        ob.update(0)
        sm.calc_visible_and_occluded_area(0, np.array(pose[:2]), pose[2], ob)
        b = SpawnLocator(None, ref, cfg, sm, dt=0.1, horizon=3.0).sample(np.array(pose[:2]), pose[2], 7.0)
        torch.cuda.synchronize()
        return [t.cpu().numpy().copy() for t in (sm.range, sm.hit_id, sm.cell_class, sm.occluded_cells(), b.pos, b.yaw, b.len,
                                                  b.type, b.n)] + [sm.road_raster()]

    mk = lambda **kw: SensorModel(sc.lanelets, ref, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, routes=3, **kw)
    own = [outputs(mk(), p) for p in poses]                       # every ego with a map of its own
    first = mk()
    second = mk(share_map_with=first)
    assert second.map_geometry is first.map_geometry
    got = [outputs(first, poses[0]), outputs(second, poses[1])]
    for a, b in zip(own, got):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
    del first                                                     # the map is reference counted
    gc.collect()
    again = outputs(second, poses[1])
    for x, y in zip(own[1], again):
        assert np.array_equal(x, y, equal_nan=True)
    third = mk(share_map_with=second)
    third._set_map(S.load_geometry_npz(os.path.join(GOLDEN, "scenario2_geometry.npz")).lanelets)   # detaches
    for x, y in zip(own[1], outputs(second, poses[1])):
        assert np.array_equal(x, y, equal_nan=True)
    other_dev_free = N.Context(0)
    with pytest.raises(N.NativeError):
        other_dev_free.call("fo_scene_share_map", N.Context(0)._h)    # the owner has no map


def test_tables_of_a_shared_map_cannot_be_replaced(torch_cuda):
    """fo_scene_share_map: the route table and the edge-line labels belong to the map.  While two contexts read one map,
    fo_scene_set_routes / fo_scene_set_edge_lines on either fail with FO_E_STATE instead of freeing tables under the
    other's kernels; fo_scene_set_map detaches the caller and is always allowed."""
    import ctypes as C
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import scenario as SC
    from frenetix_occlusion.sensor_model import SensorModel
    sc = SC.load_geometry_npz(os.path.join(GOLDEN, "scenario1_geometry.npz"))
    ego = sc.ego_initial
    owner = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0, n_rays=360)
    owner.upload_obstacles(sc.obstacle_arrays(0)[:3])
    owner.launch(ego[:2], float(ego[2]))
    torch_cuda.cuda.synchronize()
    before = (owner.range.cpu().numpy().copy(), owner.cell_class.cpu().numpy().copy())
    sharer = SensorModel(sc.lanelets, None, sensor_radius=50.0, sensor_angle=360.0, n_rays=360, share_map_with=owner)
    E = len(owner.map_geometry.edges)
    line = np.zeros(E, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.c_void_p)
    for ctx in (sharer.ctx, owner.ctx):
        with pytest.raises(N.NativeError) as e:
            ctx.call("fo_scene_set_edge_lines", E, ip(line))
        assert e.value.code == N.FO_E_STATE
        first, count = np.zeros(len(sc.lanelets), dtype=np.int32), np.zeros(len(sc.lanelets), dtype=np.int32)
        xy, s_ = np.zeros((1, 2)), np.zeros(1)
        rast = np.full(int(np.prod(owner.raster_dims)), -1, dtype=np.int32)
        with pytest.raises(N.NativeError) as e:
            ctx.call("fo_scene_set_routes", len(sc.lanelets), 1, ip(first), ip(count), 1, ip(xy), ip(s_), ip(rast))
        assert e.value.code == N.FO_E_STATE
    owner.launch(ego[:2], float(ego[2]))                     # the owner's map is what it was
    torch_cuda.cuda.synchronize()
    assert np.array_equal(owner.range.cpu().numpy(), before[0]) and np.array_equal(owner.cell_class.cpu().numpy(), before[1])
    del sharer                                               # one reader again: the labels may be replaced
    import gc
    gc.collect()
    owner.ctx.call("fo_scene_set_edge_lines", E, ip(np.arange(E, dtype=np.int32)))
