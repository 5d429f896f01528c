"""Parity of the HIP sweep (through the C ABI, libfo_hip.so) against the CPU oracle and the reference's golden
vectors.  Needs a real MI355X: run with `pytest -m gpu`."""
import os

import numpy as np
import pytest

from golden_util import CASES, load_case

pytestmark = pytest.mark.gpu

# north_star tolerance: 1e-5 on float metrics, exact on integer outputs.  The HIP path is float64, so the tests
# hold it to ATOL below (far tighter) and report the worst deviation.
ATOL = 1e-9
NORTH_STAR_ATOL = 1e-5


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    return torch


@pytest.fixture(autouse=True, params=["auto", "one-agent-per-wave"])
def sweep_kernel_mode(request, monkeypatch):
    """small batches take the kernel variant that splits an agent's horizon over the waves of a workgroup; every test of
    this module also runs with that variant switched off, so that both code paths meet every case"""
    if request.param != "auto":
        monkeypatch.setenv("FO_SWEEP_SPLIT", "0")
    return request.param


def _hip_sweep(torch, traj, agents, veh, dt, metrics=None, thr=None, mode="full", lists="f64"):
    from frenetix_occlusion.sweep import DEFAULT_METRICS, MetricSweep
    sw = MetricSweep(veh, dt, metrics=metrics or DEFAULT_METRICS, thresholds=thr)
    sw.set_agents(agents["pos"], agents["yaw"], agents["v"], agents["cov"], agents["shape"], agents["raw_dims"],
                  agents["type"], agents["len"])
    out = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], traj.get("a"), mode=mode, lists=lists)
    torch.cuda.synchronize()
    res = {"cost": out.cost.cpu().numpy(), "safe": out.safe.cpu().numpy(), "launch": sw.ctx.last_launch()}
    if out.pair_f is not None:
        res["pair_f"] = out.pair_f.permute(2, 1, 0).cpu().numpy()   # -> [M,A,NPF] (oracle layout)
        res["pair_i"] = out.pair_i.permute(2, 1, 0).cpu().numpy()
    if out.lists is not None:
        res["lists"] = out.lists.permute(3, 1, 0, 2).cpu().numpy()  # -> [M,A,NL,T-1]
    return res


def _compare(oracle, ref, got, atol=ATOL):
    """ref = oracle output dict, got = HIP output dict (both in oracle layout)."""
    PF, PI, C = oracle.PF, oracle.PI, oracle.COST
    worst = 0.0
    # Plateaus of the collision probability: several samples within 2 atol of the pair's maximum (a tight covariance
    # whose box probabilities saturate; float noise of the two implementations then decides np.argmax's "first
    # maximum", hr.py:81).  harm_with_cp = obst_harm[argmax cp] is discontinuous there: on such pairs it is checked
    # against the oracle's harm at the index the HIP side picked instead.
    have_lists = "lists" in ref and ref["lists"] is not None and ref["lists"].shape[-1] > 0
    plateau = np.zeros(ref["pair_f"].shape[:2], dtype=bool)
    if have_lists:
        cpv, mxc = ref["lists"][:, :, oracle.LST["cp"], :], ref["pair_f"][..., PF["max_collision_probability"]]
        plateau = (np.nan_to_num(mxc) > 0.01 + atol) & ((np.abs(cpv - mxc[..., None]) <= 2 * atol).sum(axis=-1) > 1)
        if plateau.any():
            gi = got["pair_i"][..., PI["cp_argmax"]].astype(np.int64)
            oh = np.take_along_axis(ref["lists"][:, :, oracle.LST["obst_harm"], :], gi[..., None], axis=-1)[..., 0]
            hw = got["pair_f"][..., PF["max_obst_harm_with_cp"]]
            assert np.all(np.abs(hw - oh)[plateau] <= atol), "max_obst_harm_with_cp on a cp plateau"
    # float pair scalars
    for name in ("dce", "ttc", "ttce", "max_ego_risk", "max_obst_risk", "max_obst_harm_with_cp", "max_ego_harm",
                 "max_obst_harm", "max_collision_probability"):
        a, b = ref["pair_f"][..., PF[name]], got["pair_f"][..., PF[name]]
        if name == "max_obst_harm_with_cp" and plateau.any():
            a, b = np.where(plateau, 0.0, a), np.where(plateau, 0.0, b)
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        assert np.array_equal(np.isinf(a), np.isinf(b)), name
        fin = np.isfinite(a)
        if fin.any():
            worst = max(worst, float(np.abs(a[fin] - b[fin]).max()))
            np.testing.assert_allclose(b[fin], a[fin], rtol=0, atol=atol, err_msg=name)
    # integer outputs: exact
    assert np.array_equal(ref["pair_i"][..., PI["time_dce"]], got["pair_i"][..., PI["time_dce"]])
    assert np.array_equal(ref["pair_i"][..., PI["hr_valid"]], got["pair_i"][..., PI["hr_valid"]])
    # argmax-type indices: equal wherever the maximum is numerically significant; elsewhere the HIP index must at
    # least point at a value within atol of the oracle's maximum
    if "lists" in ref and ref["lists"] is not None and ref["lists"].shape[-1] > 0:
        for idx_name, lst, mx in (("max_obst_risk_index", oracle.LST["obst_risk"], "max_obst_risk"),
                                  ("cp_argmax", oracle.LST["cp"], "max_collision_probability")):
            ri, gi = ref["pair_i"][..., PI[idx_name]], got["pair_i"][..., PI[idx_name]]
            vals = ref["lists"][:, :, lst, :]
            picked = np.take_along_axis(vals, gi[..., None].astype(np.int64), axis=-1)[..., 0]
            mxv = ref["pair_f"][..., PF[mx]]
            ok = np.isnan(mxv) | (np.abs(picked - mxv) <= atol)
            assert ok.all(), idx_name
            # ... and unique: on a plateau (several samples within 2 atol of the maximum -- e.g. a tight covariance
            # whose box probabilities saturate) the first-maximum rule may pick another sample of the plateau
            sig = (np.nan_to_num(mxv) > 1e-9) & ((np.abs(vals - mxv[..., None]) <= 2 * atol).sum(axis=-1) == 1)
            assert np.array_equal(ri[sig], gi[sig]), idx_name
        a, b = ref["lists"], got["lists"]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        fin = np.isfinite(a)
        if fin.any():
            worst = max(worst, float(np.abs(a[fin] - b[fin]).max()))
            np.testing.assert_allclose(b[fin], a[fin], rtol=0, atol=atol)
    # per-trajectory cost vector + flags
    for name in ("wttc", "min_dce", "max_ego_risk_all", "max_obst_risk_all", "max_ego_harm_all", "max_obst_harm_all",
                 "max_collision_probability_all", "max_obst_harm_with_cp_all", "min_ttce"):
        a, b = ref["cost"][:, C[name]], got["cost"][:, C[name]]
        assert np.array_equal(np.isinf(a), np.isinf(b)), name
        fin = np.isfinite(a)
        if name == "max_obst_harm_with_cp_all":
            fin &= ~plateau.any(axis=1)
        np.testing.assert_allclose(b[fin], a[fin], rtol=0, atol=atol, err_msg=name)
    for name in ("argmin_dce", "argmin_ttc"):
        assert np.array_equal(ref["cost"][:, C[name]], got["cost"][:, C[name]]), name
    assert np.array_equal(ref["safe"], got["safe"])
    assert np.array_equal(ref["cost"][:, C["safe"]], got["cost"][:, C["safe"]])
    return worst


@pytest.mark.parametrize("name", CASES)
def test_hip_matches_oracle_on_golden_inputs(torch_cuda, oracle, name):
    g, traj, agents, veh, dt = load_case(name)
    ref = oracle.sweep(traj, agents, veh, dt, thr={"harm": 0.1, "risk": 1})
    got = _hip_sweep(torch_cuda, traj, agents, veh, dt, thr={"harm": 0.1, "risk": 1})
    worst = _compare(oracle, ref, got)
    assert worst < NORTH_STAR_ATOL


@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_golden_outputs(torch_cuda, name):
    """Directly against what the reference's own CP/HR code produced (no oracle in between)."""
    from frenetix_occlusion import _native as N
    g, traj, agents, veh, dt = load_case(name)
    got = _hip_sweep(torch_cuda, traj, agents, veh, dt)
    for key, slot in (("cp", 0), ("ego_harm", 1), ("obst_harm", 2), ("ego_risk", 3), ("obst_risk", 4)):
        ref = g["ref_" + key]
        out = got["lists"][:, :, slot, :]
        assert np.array_equal(np.isnan(ref), np.isnan(out)), key
        np.testing.assert_allclose(out, ref, rtol=0, atol=ATOL, equal_nan=True, err_msg=key)
    for key in ("max_ego_risk", "max_obst_risk", "max_obst_harm_with_cp", "max_ego_harm", "max_obst_harm",
                "max_collision_probability"):
        np.testing.assert_allclose(got["pair_f"][..., N.PF[key]], g["ref_" + key], rtol=0, atol=ATOL, err_msg=key)
    for key in ("max_ego_risk_all", "max_obst_risk_all", "max_ego_harm_all", "max_obst_harm_all",
                "max_collision_probability_all", "max_obst_harm_with_cp_all"):
        np.testing.assert_allclose(got["cost"][:, N.COST[key]], g["ref_" + key], rtol=0, atol=ATOL, err_msg=key)
    sig = g["ref_max_obst_risk"] > 1e-9
    assert np.array_equal(got["pair_i"][..., N.PI["max_obst_risk_index"]][sig], g["ref_max_obst_risk_index"][sig])


@pytest.mark.parametrize("M,A,cfg", [(300, 16, 1), (2000, 32, 2), (130, 5, 3), (64, 1, 4), (1, 3, 5)])
def test_hip_matches_oracle_on_synthetic_batches(torch_cuda, oracle, M, A, cfg):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(M, A, config_id=cfg)
    thr = {"harm": 0.3, "risk": 0.2, "ttc": 1.0, "dce": 0.05, "cp": 0.8}
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, nthreads=8)
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr)
    worst = _compare(oracle, ref, got)
    assert worst < NORTH_STAR_ATOL
    if M >= 300:
        assert 0 < ref["safe"].mean() < 1  # both verdicts occur


def test_generic_kernel_and_long_horizon(torch_cuda, oracle, monkeypatch):
    """The queue kernel serves T-1 <= 30; longer horizons (and FO_SWEEP_GENERIC=1) take the generic kernel."""
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(200, 12, config_id=11)
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.3})
    monkeypatch.setenv("FO_SWEEP_GENERIC", "1")
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.3})
    _compare(oracle, ref, got)
    monkeypatch.delenv("FO_SWEEP_GENERIC")
    traj = S.make_trajectories(150, T=45, seed=5)
    agents = S.make_agents(7, T=45, seed=5)
    agents["len"][:] = [45, 31, 40, 44, 45, 2, 45]
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1)
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1)
    _compare(oracle, ref, got)


def test_output_modes_agree(torch_cuda):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(500, 24, config_id=6)
    full = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, mode="full")
    pair = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, mode="pair")
    red = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, mode="reduced")
    assert np.array_equal(full["cost"], pair["cost"]) and np.array_equal(full["cost"], red["cost"])
    assert np.array_equal(full["pair_f"], pair["pair_f"], equal_nan=True)
    assert np.array_equal(full["safe"], red["safe"])


# float32 list storage (fo_sweep_set_list_format(FO_LISTS_F32), SURVEY 8d's 648 B per pair): the lists are held to
# LIST32_ATOL against the float64 oracle and the reference's goldens; everything else must not move by a bit
LIST32_ATOL = 1e-6


def _compare_f32_lists(ref_lists, got32, got64):
    assert got32["lists"].dtype == np.float32
    for k in ("cost", "safe", "pair_i"):
        assert np.array_equal(got32[k], got64[k]), k
    assert np.array_equal(got32["pair_f"], got64["pair_f"], equal_nan=True)
    assert np.array_equal(np.isnan(ref_lists), np.isnan(got32["lists"]))
    fin = np.isfinite(ref_lists)
    worst = float(np.abs(ref_lists[fin] - got32["lists"][fin].astype(np.float64)).max()) if fin.any() else 0.0
    assert worst < LIST32_ATOL, worst
    # consistent among themselves: risk = harm x cp holds in the stored lists up to float32 rounding
    cp, eh, oh, er, orr = (got32["lists"][:, :, i, :].astype(np.float64) for i in range(5))
    f = np.isfinite(er)
    assert np.abs(er[f] - (eh * cp)[f]).max() < 3e-7 and np.abs(orr[f] - (oh * cp)[f]).max() < 3e-7
    return worst


@pytest.mark.parametrize("M,A,cfg", [(300, 16, 1), (2000, 32, 2), (130, 5, 3), (70, 9, 8)])
def test_float32_lists_match_oracle(torch_cuda, oracle, M, A, cfg):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(M, A, config_id=cfg)
    if cfg == 8:
        agents["len"] = np.array([31, 1, 2, 30, 17, 31, 5, 29, 0], dtype=np.int32)   # ragged + an inactive slot
    thr = {"harm": 0.3, "risk": 0.2, "ttc": 1.0, "dce": 0.05, "cp": 0.8}
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, nthreads=8)
    g64 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr)
    g32 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, lists="f32")
    _compare_f32_lists(ref["lists"], g32, g64)


@pytest.mark.parametrize("name", CASES)
def test_float32_lists_match_reference_goldens(torch_cuda, name):
    """the float32 lists directly against what the reference's own CP / HR code produced"""
    g, traj, agents, veh, dt = load_case(name)
    g64 = _hip_sweep(torch_cuda, traj, agents, veh, dt)
    g32 = _hip_sweep(torch_cuda, traj, agents, veh, dt, lists="f32")
    ref = np.stack([g["ref_" + k] for k in ("cp", "ego_harm", "obst_harm", "ego_risk", "obst_risk")], axis=2)
    _compare_f32_lists(ref, g32, g64)


def test_float32_lists_generic_kernel_and_metric_subset(torch_cuda, oracle, monkeypatch):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(200, 12, config_id=11)
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"))
    g64 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"))
    g32 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), lists="f32")
    _compare_f32_lists(ref["lists"], g32, g64)
    monkeypatch.setenv("FO_SWEEP_GENERIC", "1")    # (the generic kernel's libm route differs from the queue kernel's in the last bits)
    g64 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"))
    g32 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), lists="f32")
    _compare_f32_lists(ref["lists"], g32, g64)
    # a metric set that writes no lists at all: NaN-filled in either format
    monkeypatch.delenv("FO_SWEEP_GENERIC")
    g32 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("dce",), lists="f32")
    assert g32["lists"].dtype == np.float32 and np.isnan(g32["lists"]).all()


def _assert_rounded_float64(got32, want64, tol=1e-9):
    """every float32 entry is a float64 value within `tol` of `want64`, rounded to float32 (half a float32 ulp + tol)"""
    assert got32.dtype == np.float32
    assert np.array_equal(np.isnan(got32), np.isnan(want64))
    f = np.isfinite(want64)
    half_ulp = 0.5 * np.spacing(np.abs(want64[f]).astype(np.float32)).astype(np.float64)
    err = np.abs(got32[f].astype(np.float64) - want64[f])
    assert (err <= half_ulp + tol).all(), float((err - half_ulp).max())
    return float(np.mean(got32[f] != want64[f].astype(np.float32)))


@pytest.mark.parametrize("M,A,cfg", [(300, 16, 1), (2000, 32, 2), (70, 9, 8), (3300, 64, 3)])
def test_float32_exact_lists_are_the_float64_lists_rounded(torch_cuda, oracle, monkeypatch, M, A, cfg):
    """fo_sweep_set_list_format(FO_LISTS_F32_EXACT) / lists='f32x' -- the format bench.py's headline runs: float32 storage of
    the float64 results.  Every list entry equals the float64-list mode's entry rounded to float32, everything else is
    bit-identical, and against the oracle every entry is a float64 value within 1e-9 of the oracle's, rounded to float32 --
    in EVERY form of the queue kernel (round 6: the format has all the instantiations of the other formats): the full grid
    (3 300 x 64 and, forced, the small ones), the horizon-split form small batches take by themselves (2 000 x 32 = BASELINE
    configs[1]'s shape), a metric subset such as the reference's configs[0] (['hr', 'ttc'], metric.py:125-147), and both at
    once; the generic kernel (FO_SWEEP_GENERIC=1) stays the same arithmetic converted at the store."""
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(M, A, config_id=cfg)
    if cfg == 8:
        agents["len"] = np.array([31, 1, 2, 30, 17, 31, 5, 29, 0], dtype=np.int32)
    thr = {"harm": 0.3, "risk": 0.2, "ttc": 1.0, "dce": 0.05, "cp": 0.8}
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, nthreads=8)
    ref_sub = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), nthreads=8)
    seen = set()
    for split in (None, "1", "0"):            # the library's own choice, then both forms forced
        if split is None:
            monkeypatch.delenv("FO_SWEEP_SPLIT", raising=False)
        else:
            monkeypatch.setenv("FO_SWEEP_SPLIT", split)
        g64 = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr)
        gx = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr, lists="f32x")
        assert gx["lists"].dtype == np.float32
        for k in ("cost", "safe", "pair_i"):
            assert np.array_equal(gx[k], g64[k]), (split, k)
        assert np.array_equal(gx["pair_f"], g64["pair_f"], equal_nan=True)
        assert np.array_equal(gx["lists"], g64["lists"].astype(np.float32), equal_nan=True), split
        _assert_rounded_float64(gx["lists"], ref["lists"], tol=1e-9)
        # metric subset: the queue kernel's instantiation without the compile-time metric set, float64 twin of the same form
        g64s = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"))
        gxs = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), lists="f32x")
        for k in ("cost", "safe", "pair_i"):
            assert np.array_equal(gxs[k], g64s[k]), (split, k)
        assert np.array_equal(gxs["pair_f"], g64s["pair_f"], equal_nan=True)
        assert np.array_equal(gxs["lists"], g64s["lists"].astype(np.float32), equal_nan=True), split
        _assert_rounded_float64(gxs["lists"], ref_sub["lists"], tol=1e-9)
        seen.add((gx["launch"]["grid"], gxs["launch"]["grid"]))
        assert gx["launch"] == g64["launch"] and gxs["launch"] == g64s["launch"]      # the same form for both formats
    assert len(seen) == 2       # the two forms really are two launches (split: one workgroup per (tile, agent))
    # the generic kernel: the same arithmetic from libm, converted at the store
    monkeypatch.delenv("FO_SWEEP_SPLIT", raising=False)
    monkeypatch.setenv("FO_SWEEP_GENERIC", "1")
    g64g = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"))
    gxg = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), lists="f32x")
    _compare_f32_lists(ref_sub["lists"], gxg, g64g)
    assert np.array_equal(gxg["lists"], g64g["lists"].astype(np.float32), equal_nan=True)


def test_autotune_measures_and_keeps_a_setting_without_changing_results(torch_cuda):
    """fo_sweep_autotune: the library times the sweep kernel's agents-per-wave settings on the caller's batch, remembers the
    best for the shape, and leaves a complete result; later runs of that shape use it, other shapes keep the static rule"""
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    traj, agents = S.make_batch(4000, 64, config_id=4)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1})
    sw.set_agents(*[agents[k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")])
    args = [traj[k] for k in ("x", "y", "theta", "v", "a")]
    plain = sw.run(*args, mode="pair")
    torch_cuda.cuda.synchronize()
    want = [t.cpu().numpy().copy() for t in (plain.cost, plain.safe, plain.pair_f, plain.pair_i)]
    tuned = sw.run(*args, mode="pair", autotune=5)
    torch_cuda.cuda.synchronize()
    best = sw.last_autotune["agents_per_wave"]
    assert best in (1, 2, 4, 8) and set(sw.last_autotune["ms"]) == {1, 2, 4, 8} and all(v > 0 for v in sw.last_autotune["ms"].values())
    assert sw.last_autotune["ms"][best] == min(sw.last_autotune["ms"].values())
    for a, b in zip(want, (tuned.cost, tuned.safe, tuned.pair_f, tuned.pair_i)):
        assert np.array_equal(a, b.cpu().numpy(), equal_nan=True)
    again = sw.run(*args, mode="pair")
    torch_cuda.cuda.synchronize()
    if os.environ.get("FO_SWEEP_SPLIT") is None:
        assert sw.ctx.last_launch()["agents_per_wave"] == best
    assert np.array_equal(again.cost.cpu().numpy(), want[0], equal_nan=True)
    sw.run(*[q[:1000] for q in args], mode="pair")                 # another shape: not in the table
    torch_cuda.cuda.synchronize()


def test_metric_subset_config1(torch_cuda, oracle):
    """BASELINE config 1: activated_metrics = ['hr', 'ttc'] -> cp, dce, ttc, hr (metric.py:125-147)."""
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(200, 6, config_id=1)
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), thr={"harm": 0.1})
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=("hr", "ttc"), thr={"harm": 0.1})
    _compare(oracle, ref, got)
    assert np.isnan(got["pair_f"][..., oracle.PF["ttce"]]).all()


def test_ragged_and_edge_cases(torch_cuda, oracle):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(70, 9, config_id=8)
    agents["len"] = np.array([31, 1, 2, 30, 17, 31, 5, 29, 3], dtype=np.int32)
    agents["cov"][2] = 0.0  # zero covariance -> 0.1 I
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1)
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1)
    _compare(oracle, ref, got)
    # T = 2 and T = 1 trajectories
    for T in (2, 1):
        t2 = {k: v[:, :T].copy() for k, v in traj.items()}
        ref = oracle.sweep(t2, agents, S.VEHICLE_BMW320I, 0.1)
        got = _hip_sweep(torch_cuda, t2, agents, S.VEHICLE_BMW320I, 0.1)
        _compare(oracle, ref, got)


def test_no_agents_everything_safe(torch_cuda):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(100, 4, config_id=9)
    empty = {k: v[:0] for k, v in agents.items()}
    got = _hip_sweep(torch_cuda, traj, empty, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.0, "ttc": 9}, mode="reduced")
    assert got["safe"].all() and np.isinf(got["cost"][:, 0]).all()


def test_off_diagonal_covariance_fails_loudly(torch_cuda):
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    traj, agents = S.make_batch(10, 2, config_id=10)
    agents["cov"][1, :, 0, 1] = 0.01          # asymmetric: no covariance matrix (a symmetric one with |rho| <= 0.99
    agents["cov"][1, :, 1, 0] = 0.02          # is integrated numerically, test_correlated_covariances_*)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1)
    with pytest.raises(N.NativeError) as e:
        sw.set_agents(agents["pos"], agents["yaw"], agents["v"], agents["cov"], agents["shape"], agents["raw_dims"],
                      agents["type"], agents["len"])
    assert e.value.code == N.FO_E_UNSUPPORTED_COV
    # the condition belongs to that agent set only: a clean set on the same context passes, a bad one fails again
    good = S.make_batch(10, 2, config_id=10)[1]
    sw.set_agents(good["pos"], good["yaw"], good["v"], good["cov"], good["shape"], good["raw_dims"], good["type"],
                  good["len"])
    sw.set_agents(good["pos"], good["yaw"], good["v"], good["cov"], good["shape"], good["raw_dims"], good["type"],
                  good["len"], check=False)
    sw.ctx.call("fo_sweep_check", torch_cuda.cuda.current_stream().cuda_stream)
    with pytest.raises(N.NativeError):
        sw.set_agents(agents["pos"], agents["yaw"], agents["v"], agents["cov"], agents["shape"], agents["raw_dims"],
                      agents["type"], agents["len"])


def test_unusable_covariance_reads_nan_per_pair(torch_cuda):
    """an agent whose covariance is no covariance: its collision probabilities -- pair scalar, list, risks -- are NaN (not
    0 = "no risk"), the other agents' are untouched, and no trajectory reads as safe (include/fo_hip.h, fo_sweep_check)"""
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    traj, agents = S.make_batch(128, 3, config_id=10)
    agents["pos"][1, :, :] = np.stack((traj["x"][5], traj["y"][5]), -1)     # agent 1 sits on trajectory 5: inside the gate
    good = {k: v.copy() for k, v in agents.items()}
    agents["cov"][1, :, 0, 1] = 0.01
    agents["cov"][1, :, 1, 0] = 0.02
    res = {}
    for name, ag in (("bad", agents), ("good", good)):
        sw = MetricSweep(S.VEHICLE_BMW320I, 0.1)
        sw.set_agents(*[ag[k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")], check=False)
        out = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], mode="full")
        torch_cuda.cuda.synchronize()
        res[name] = (out.pair_f.cpu().numpy(), out.lists.cpu().numpy(), out.safe.cpu().numpy())
    pf, ls, safe = res["bad"]
    gpf, gls, _ = res["good"]
    assert gpf[N.PF["max_collision_probability"], 1, 5] > 0.01              # the clean run does see a probability there
    assert np.isnan(pf[N.PF["max_collision_probability"], 1, 5]) and np.isnan(pf[N.PF["max_obst_risk"], 1, 5])
    ingate = gls[N.LST["cp"], 1, :, 5] > 0
    assert ingate.any() and np.isnan(ls[N.LST["cp"], 1, :, 5][ingate]).all()
    for k in (0, 2):                                                         # the other agents: bit-identical
        assert np.array_equal(pf[:, k], gpf[:, k], equal_nan=True) and np.array_equal(ls[:, k], gls[:, k], equal_nan=True)
    assert not safe.any()


def test_full_size_properties_10k_x_256(torch_cuda):
    """BASELINE full size (10 000 x 256): size-independent properties instead of an oracle run.
    (1) permuting the agents permutes pair outputs and leaves the per-trajectory maxima unchanged;
    (2) a trajectory slice evaluated alone gives bit-identical rows (what sharding over GPUs relies on);
    (3) reduced mode == maxima of the pair outputs."""
    torch = torch_cuda
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    M, A = 10000, 256
    traj, agents = S.make_batch(M, A, config_id=3)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds={"harm": 0.1, "risk": 1})
    args = [agents[k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")]
    sw.set_agents(*args)
    base = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], mode="pair")
    torch.cuda.synchronize()
    cost, pf = base.cost.cpu().numpy(), base.pair_f.cpu().numpy()
    # (3)
    assert np.array_equal(cost[:, N.COST["min_dce"]], pf[N.PF["dce"]].min(axis=0))
    assert np.array_equal(cost[:, N.COST["max_obst_risk_all"]], np.maximum(pf[N.PF["max_obst_risk"]].max(axis=0), 0))
    assert np.array_equal(cost[:, N.COST["wttc"]], pf[N.PF["ttc"]].min(axis=0))
    # (1)
    perm = np.random.default_rng(0).permutation(A)
    sw.set_agents(*[a[perm] for a in args])
    p = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], mode="pair")
    torch.cuda.synchronize()
    assert np.array_equal(p.pair_f.cpu().numpy(), pf[:, perm], equal_nan=True)
    for name in ("wttc", "min_dce", "max_obst_risk_all", "max_ego_harm_all", "max_collision_probability_all",
                 "max_obst_harm_with_cp_all", "safe"):
        assert np.array_equal(p.cost.cpu().numpy()[:, N.COST[name]], cost[:, N.COST[name]]), name
    # (2)
    sw.set_agents(*args)
    sl = slice(1250 * 3, 1250 * 4)
    part = sw.run(traj["x"][sl], traj["y"][sl], traj["theta"][sl], traj["v"][sl], mode="reduced")
    torch.cuda.synchronize()
    assert np.array_equal(part.cost.cpu().numpy(), cost[sl])


def test_be_metric_matches_oracle(torch_cuda, oracle):
    """metrics/be.py (optional metric): required constant deceleration + brake threat number per colliding pair."""
    from frenetix_occlusion import synthetic as S
    metrics = ("hr", "ttc", "ttce", "dce", "wttc", "cp", "be")
    for M, A, cfg, thr in ((300, 16, 1, {"be": 0.2, "harm": 0.9}), (130, 9, 8, {"be": 0.05})):
        traj, agents = S.make_batch(M, A, config_id=cfg)
        traj["a"][::3, 4] -= 1.237                               # some trajectories already brake
        if cfg == 8:
            agents["len"] = np.array([31, 1, 2, 30, 17, 31, 5, 29, 3], dtype=np.int32)
        ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr)
        got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr)
        _compare(oracle, ref, got)
        for name in ("be_decel", "be_btn"):
            a, b = ref["pair_f"][..., oracle.PF[name]], got["pair_f"][..., oracle.PF[name]]
            assert np.array_equal(np.isnan(a), np.isnan(b)), name
            np.testing.assert_allclose(np.nan_to_num(b), np.nan_to_num(a), rtol=0, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(got["cost"][:, oracle.COST["max_btn"]], ref["cost"][:, oracle.COST["max_btn"]], atol=1e-12)
        assert (ref["pair_f"][..., oracle.PF["be_decel"]] > 0).sum() > 5
        red = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr, mode="reduced")
        assert np.array_equal(red["safe"], ref["safe"]) and np.array_equal(red["cost"], got["cost"])
    # without the acceleration profile the call fails loudly
    from frenetix_occlusion import _native as N
    from frenetix_occlusion.sweep import MetricSweep
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, metrics=metrics)
    sw.set_agents(*[agents[k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")])
    with pytest.raises(N.NativeError):
        sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], None)


def test_dce_matches_the_references_own_walk_over_the_time_steps(torch_cuda):
    """the device's DCE / TTC / TTCE / WTTC against tests/golden/dce_loop.npz -- what the reference's own, unmodified DCE class
    and the three metrics on top of it returned (gen_golden.py dce; only the rectangle distance under the loop is this
    repository's): distances to 1e-9 (they are whole millimetres), time_dce, ttc, ttce, wttc exact"""
    from golden_util import load_dce_case
    from oracle import fo_oracle as O
    g, traj, agents, veh, dt = load_dce_case()
    for mode in ("full", "reduced"):
        got = _hip_sweep(torch_cuda, traj, agents, veh, dt, metrics=("dce", "ttc", "ttce", "wttc"), mode=mode)
        assert np.array_equal(got["cost"][:, O.COST["wttc"]], g["ref_wttc"])
        np.testing.assert_allclose(got["cost"][:, O.COST["min_dce"]], g["ref_dce"].min(axis=1), rtol=0, atol=1e-9)
        if mode == "full":
            np.testing.assert_allclose(got["pair_f"][..., O.PF["dce"]], g["ref_dce"], rtol=0, atol=1e-9)
            assert np.array_equal(got["pair_i"][..., O.PI["time_dce"]], g["ref_time_dce"])
            assert np.array_equal(got["pair_f"][..., O.PF["ttc"]], g["ref_ttc"])
            assert np.array_equal(got["pair_f"][..., O.PF["ttce"]], g["ref_ttce"])


def test_be_metric_matches_the_references_own_bisection(torch_cuda):
    """the device's BE against tests/golden/be_bisection.npz -- what the reference's own, unmodified BE class returned
    (gen_golden.py be; only the rectangle `intersects` predicate under it is this repository's): every required deceleration
    and brake threat number, in full and in reduced mode (max_btn)"""
    from golden_util import load_be_case
    g, traj, agents, veh, dt = load_be_case()
    metrics = ("dce", "ttc", "be")
    got = _hip_sweep(torch_cuda, traj, agents, veh, dt, metrics=metrics, thr={"be": 0.2})
    from oracle import fo_oracle as O
    np.testing.assert_allclose(got["pair_f"][..., O.PF["be_decel"]], g["be_decel"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got["pair_f"][..., O.PF["be_btn"]], g["be_btn"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got["cost"][:, O.COST["max_btn"]], g["be_btn"].max(axis=1), rtol=0, atol=1e-12)
    assert np.array_equal(got["safe"].astype(bool), ~(g["be_btn"].max(axis=1) > 0.2))       # metric.py:54-61
    red = _hip_sweep(torch_cuda, traj, agents, veh, dt, metrics=metrics, thr={"be": 0.2}, mode="reduced")
    assert np.array_equal(red["cost"], got["cost"]) and np.array_equal(red["safe"], got["safe"])


def test_empty_trajectory_batch_and_single_lane(torch_cuda, oracle):
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(65, 3, config_id=12)
    empty = {k: v[:0] for k, v in traj.items()}
    got = _hip_sweep(torch_cuda, empty, agents, S.VEHICLE_BMW320I, 0.1, mode="full")
    assert got["cost"].shape == (0, 16) and got["safe"].shape == (0,) and got["lists"].shape == (0, 3, 5, 30)
    # 65 trajectories: one full tile + one tile with a single valid lane
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.2})
    _compare(oracle, ref, _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.2}))
    # agents standing exactly on the ego's start pose (atan2(0, 0) branch of the angle classes, overlap at t = 0)
    agents["pos"][0] = np.array([traj["x"][0, 0], traj["y"][0, 0]])
    agents["type"][0] = 0
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1)
    _compare(oracle, ref, _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1))


def test_every_obstacle_type_and_many_agents(torch_cuda, oracle):
    """all ObstacleType codes (mass / protection tables of harm_model.py:15-32,158-190) and A > one agent chunk"""
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(96, 70, config_id=13)
    agents["type"] = (np.arange(70) % 12).astype(np.int32)
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"risk": 0.3}, nthreads=8)
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"risk": 0.3})
    _compare(oracle, ref, got)


def test_gate_of_agents_that_jump_between_samples(torch_cuda, oracle, monkeypatch):
    """The sweep's coarse gate test works on the distance ego centre t -> agent mean t, although the gate of sample t-1
    pairs the ego with the agent mean of sample t-1: the agent's longest step is part of the coarse radius
    (fo_agent_rows.hpp).  Agents whose mean jumps by up to 40 m from sample to sample -- into the gate and out of it
    again -- must give the oracle's collision probabilities; both kernel variants, two agent sets in a row on ONE
    context (the longest step sits in a generation-tagged slot that is never reset)."""
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import DEFAULT_METRICS, MetricSweep
    torch = torch_cuda
    rng = np.random.default_rng(77)
    M, A, T = 200, 24, 31
    traj = S.make_trajectories(M, T, 0.1, seed=3)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, metrics=DEFAULT_METRICS, ctx=N.Context(0))
    for rnd in range(2):
        agents = S.make_agents(A, T, 0.1, seed=40 + rnd, lateral=4.0)
        pos = agents["pos"]
        ex, ey = traj["x"], traj["y"]
        for k in range(A):
            kind = k % 4
            if kind == 0:      # sits far away and visits an ego position for single samples
                pos[k] = pos[k] + np.array([300.0, 0.0])
                for t in rng.choice(np.arange(1, T - 1), 4, replace=False):
                    m = int(rng.integers(M))
                    pos[k, t] = (ex[m, min(t + 1, T - 1)] + rng.uniform(-2, 2), ey[m, min(t + 1, T - 1)] + rng.uniform(-2, 2))
            elif kind == 1:    # random walk with steps of up to 40 m around the trajectories
                c = np.array([ex[:, T // 2].mean(), ey[:, T // 2].mean()])
                pos[k] = c + rng.uniform(-20, 20, (T, 2))
            elif kind == 2 and rnd == 1:   # smaller steps than the set before: the old maximum must not survive
                pos[k] = pos[k, :1] + 0.01 * np.arange(T)[:, None]
        agents["len"] = rng.integers(2, T + 1, A).astype(np.int32)
        ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, nthreads=8)
        assert (ref["pair_f"][..., N.PF["max_collision_probability"]] > 0).any()
        for split in ("0", "1"):
            monkeypatch.setenv("FO_SWEEP_SPLIT", split)
            sw.set_agents(agents["pos"], agents["yaw"], agents["v"], agents["cov"], agents["shape"], agents["raw_dims"],
                          agents["type"], agents["len"])
            out = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], traj.get("a"), mode="full")
            torch.cuda.synchronize()
            got = {"cost": out.cost.cpu().numpy(), "safe": out.safe.cpu().numpy(),
                   "pair_f": out.pair_f.permute(2, 1, 0).cpu().numpy(), "pair_i": out.pair_i.permute(2, 1, 0).cpu().numpy(),
                   "lists": out.lists.permute(3, 1, 0, 2).cpu().numpy()}
            _compare(oracle, ref, got)


def test_harm_maxima_with_unusual_coefficient_signs(torch_cuda, oracle):
    """Without lists the harm maxima of the two-coefficient models come from the largest relative speed of a pair --
    valid while both speed coefficients are positive (the logistic argument then falls with the relative speed); with
    a negative coefficient the kernel must take the per-sample route again.  Reduced / pair / full outputs against the
    oracle and against each other for the usual signs, one negative and both negative."""
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import DEFAULT_HARM_COEFF, DEFAULT_METRICS, MetricSweep
    torch = torch_cuda
    traj, agents = S.make_batch(200, 12, config_id=7)
    agents["type"] = np.array([4, 3, 0, 4, 3, 4, 4, 3, 3, 4, 0, 4], dtype=np.int32)   # pedestrians, bicycles, two cars
    for flip in ((), ("lr1s_speed",), ("lr1s_speed", "ped_speed")):
        hc = dict(DEFAULT_HARM_COEFF)
        for k in flip:
            hc[k] = -hc[k]
        ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, harm_coeff=hc, nthreads=8)
        res = {}
        for mode in ("full", "pair", "reduced"):
            sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, metrics=DEFAULT_METRICS, harm_coeff=hc)
            sw.set_agents(agents["pos"], agents["yaw"], agents["v"], agents["cov"], agents["shape"], agents["raw_dims"],
                          agents["type"], agents["len"])
            out = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], traj.get("a"), mode=mode)
            torch.cuda.synchronize()
            got = {"cost": out.cost.cpu().numpy(), "safe": out.safe.cpu().numpy()}
            if mode != "reduced":
                got["pair_f"] = out.pair_f.permute(2, 1, 0).cpu().numpy()
                got["pair_i"] = out.pair_i.permute(2, 1, 0).cpu().numpy()
            if mode == "full":
                got["lists"] = out.lists.permute(3, 1, 0, 2).cpu().numpy()
                _compare(oracle, ref, got)
            res[mode] = got
        # the modes without lists (where the shortcut lives) give the bits of the full mode
        for mode in ("pair", "reduced"):
            assert np.array_equal(res[mode]["cost"], res["full"]["cost"], equal_nan=True), (flip, mode)
            assert np.array_equal(res[mode]["safe"], res["full"]["safe"]), (flip, mode)
        assert np.array_equal(res["pair"]["pair_f"], res["full"]["pair_f"], equal_nan=True), flip
        assert np.array_equal(res["pair"]["pair_i"], res["full"]["pair_i"]), flip


def test_dce_ties_stationary_and_random_geometry(torch_cuda, oracle):
    """the order-independent DCE scan (probe + lower-bound skips) must reproduce 'first strict minimum' exactly:
    stationary pairs (every timestep ties), symmetric pass-bys (two equal minima), touching rectangles, and a
    randomised batch with arbitrary headings, speeds, sizes and prediction lengths"""
    from frenetix_occlusion import synthetic as S
    T = 31
    t = np.arange(T) * 0.1
    rows = []
    rows.append((np.zeros(T), np.zeros(T), np.zeros(T), np.zeros(T)))                       # ego at rest
    rows.append((6.0 * t, np.zeros(T), np.zeros(T), np.full(T, 6.0)))                        # straight
    x = np.concatenate((np.linspace(0, 9, 16), np.linspace(9, 0, 16)[1:]))                   # out and back: symmetric
    rows.append((x, np.zeros(T), np.zeros(T), np.full(T, 6.0)))
    rows.append((np.full(T, 11.2), np.full(T, 0.4), np.full(T, 0.3), np.zeros(T)))           # at rest next to agent 1
    traj = {k: np.stack([r[i] for r in rows]) for i, k in enumerate(("x", "y", "theta", "v"))}
    traj["a"] = np.zeros_like(traj["x"])
    pos = np.zeros((5, T, 2))
    pos[0] = [30.0, 0.0]                                                                     # at rest ahead
    pos[1] = [15.0, 0.0]                                                                     # at rest, will be hit
    pos[2, :, 0], pos[2, :, 1] = 20.0, 6.0 - 4.0 * t                                          # crossing
    pos[3, :, 0], pos[3, :, 1] = 12.0 + 0.0 * t, -3.0                                         # beside the path
    pos[4] = [VEH_TOUCH_X, 0.0]                                                              # touching the ego at rest
    yaw = np.zeros((5, T))
    yaw[2] = -math.pi / 2
    agents = {"pos": pos, "yaw": yaw, "v": np.zeros((5, T)), "cov": np.tile(0.1 * np.eye(2), (5, T, 1, 1)),
              "shape": np.tile([5.76, 2.6], (5, 1)), "raw_dims": np.tile([4.8, 2.0], (5, 1)),
              "type": np.array([0, 0, 4, 3, 0], dtype=np.int32), "len": np.array([T, T, T, 20, T], dtype=np.int32)}
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1)
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1)
    _compare(oracle, ref, got)
    td = ref["pair_i"][..., oracle.PI["time_dce"]]
    assert td[0, 0] == 0 and td[2, 0] == 15 and td[0, 4] == 0                 # ties -> earliest; symmetric -> first
    assert ref["pair_f"][0, 4, oracle.PF["dce"]] == 0.0                        # touching counts as distance 0
    # Coincident reference points (harm_model.py:86-90: atan2(0, 0) = 0, then the UN-wrapped angle pi + 0 - yaw decides the
    # obstacle's impact class -- "rear" for every negative yaw) and zero relative speed at every sample: vehicles parked exactly on
    # the trajectory point of the ego at rest, headings on both sides of zero and beyond pi; one of them drives off later
    yaws = np.array([-0.5, 1.5, 3.5, -3.5, 0.0, 2.8])        # pi - yaw: rear, side, front, rear (un-wrapped 6.64), rear, front
    A2 = len(yaws)
    pos2 = np.zeros((A2, T, 2))
    pos2[2, 10:, 0] = 0.8 * (t[10:] - t[10])
    v2 = np.zeros((A2, T))
    v2[2, 10:] = 0.8
    agents2 = {"pos": pos2, "yaw": np.repeat(yaws[:, None], T, 1), "v": v2, "cov": np.tile(0.1 * np.eye(2), (A2, T, 1, 1)),
               "shape": np.tile([5.76, 2.6], (A2, 1)), "raw_dims": np.tile([4.8, 2.0], (A2, 1)),
               "type": np.zeros(A2, dtype=np.int32), "len": np.full(A2, T, dtype=np.int32)}
    for lists in ("f64", "f32x"):
        ref2 = oracle.sweep(traj, agents2, S.VEHICLE_BMW320I, 0.1)
        got2 = _hip_sweep(torch_cuda, traj, agents2, S.VEHICLE_BMW320I, 0.1, lists=lists)
        _compare(oracle, ref2, got2, atol=ATOL if lists == "f64" else 1e-6)
    oh = ref2["lists"][0, :, oracle.LST["obst_harm"], 0]                       # ego at rest on top of them, first sample
    assert len({oh[0], oh[1], oh[2]}) == 3 and oh[0] == oh[3] == oh[4] and oh[2] == oh[5]      # three classes, by the un-wrapped angle
    # Centres that coincide to within ~1e-40 m (advisor, round 5): the offsets are nonzero in float64 -- atan2 sees their
    # direction -- but denormal (1e-40) or zero (1e-60) as float32, where the kernel's cheap angle estimate is 0 * inf; such
    # samples must take the float64 route.  Eight directions x two magnitudes around the ego at rest at the origin, LR4S cars,
    # headings chosen so that the three classes all occur on both sides.
    dirs8 = np.array([[1, 2], [-2, 1], [-1, -2], [2, -1], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=np.float64)
    pos3 = np.concatenate([dirs8 * 1e-40, dirs8 * 1e-60])[:, None, :].repeat(T, 1)
    A3 = pos3.shape[0]
    yaw3 = np.tile(np.array([0.3, -2.9, 2.0, -1.2, 3.3, -0.4, 1.1, -3.4]), 2)
    agents3 = {"pos": pos3, "yaw": np.repeat(yaw3[:, None], T, 1), "v": np.zeros((A3, T)), "cov": np.tile(0.1 * np.eye(2), (A3, T, 1, 1)),
               "shape": np.tile([5.76, 2.6], (A3, 1)), "raw_dims": np.tile([4.8, 2.0], (A3, 1)),
               "type": np.zeros(A3, dtype=np.int32), "len": np.full(A3, T, dtype=np.int32)}
    traj3 = {k: v.copy() for k, v in traj.items()}
    traj3["theta"][0, :] = 0.4                      # (ego at rest at the origin, heading 0.4: its own classes vary with the direction)
    ref3 = oracle.sweep(traj3, agents3, S.VEHICLE_BMW320I, 0.1)
    eh = ref3["lists"][0, :, oracle.LST["ego_harm"], 0]
    assert len(set(np.round(eh[:8], 12))) == 3 and np.array_equal(eh[:8], eh[8:])         # all three classes; magnitude is irrelevant
    for lists in ("f64", "f32x"):
        got3 = _hip_sweep(torch_cuda, traj3, agents3, S.VEHICLE_BMW320I, 0.1, lists=lists)
        _compare(oracle, ref3, got3, atol=ATOL if lists == "f64" else 1e-6)
    red3 = _hip_sweep(torch_cuda, traj3, agents3, S.VEHICLE_BMW320I, 0.1, mode="reduced")
    assert np.array_equal(red3["cost"], got3["cost"], equal_nan=True)
    rng = np.random.default_rng(123)
    for trial in range(3):
        M, A = 256, 24
        traj = S.make_trajectories(M, T, 0.1, seed=1000 + trial, order="random")
        th0 = rng.uniform(-math.pi, math.pi, (M, 1))
        c, s_ = np.cos(th0), np.sin(th0)
        traj["x"], traj["y"] = c * traj["x"] - s_ * traj["y"], s_ * traj["x"] + c * traj["y"]
        traj["theta"] = traj["theta"] + th0 + rng.choice([0.0, 2 * math.pi, -4 * math.pi], (M, 1))   # un-wrapped headings
        p0 = rng.uniform(-25, 25, (A, 2))
        ya = rng.uniform(-7, 7, A)
        sp = rng.uniform(0, 12, A) * (rng.random(A) > 0.25)
        pos = p0[:, None, :] + t[None, :, None] * (sp[:, None] * np.stack((np.cos(ya), np.sin(ya)), -1))[:, None, :]
        raw = np.stack((rng.uniform(0.3, 12, A), rng.uniform(0.3, 3, A)), -1)
        agents = {"pos": pos, "yaw": np.repeat(ya[:, None], T, 1), "v": np.repeat(sp[:, None], T, 1),
                  "cov": np.tile(np.eye(2), (A, T, 1, 1)) * rng.uniform(0.05, 0.8, (A, 1, 1, 1)),
                  "shape": raw * 1.25, "raw_dims": raw, "type": rng.integers(0, 12, A).astype(np.int32),
                  "len": rng.integers(1, T + 1, A).astype(np.int32)}
        ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.4}, nthreads=8)
        got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.4})
        _compare(oracle, ref, got)


VEH_TOUCH_X = 1.4227 + 4.508 / 2 + 2.4   # ego centre offset + half ego length + half agent length: faces touch
import math  # noqa: E402


def test_single_sample_trajectories(torch_cuda, oracle):
    """T = 1: no collision-probability or harm sample exists (cp.py loops from 1, harm_model.py:66 takes T-1 = 0),
    DCE sees the one sample; list outputs have length 0"""
    from frenetix_occlusion import synthetic as S
    for M, A in ((1, 1), (70, 5)):
        traj = {k: v[:, :1].copy() for k, v in S.make_trajectories(M, 2, 0.1, seed=3).items()}
        agents = S.make_agents(A, 2, 0.1, seed=4)
        for k in ("pos", "yaw", "v", "cov"):
            agents[k] = agents[k][:, :1].copy()
        agents["len"] = np.minimum(agents["len"], 1).astype(np.int32)
        ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1)
        got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1)
        _compare(oracle, ref, got)
        assert got["lists"].shape[-1] == 0 and np.array_equal(ref["safe"], got["safe"])


def test_off_diagonal_covariance_poisons_the_outputs_without_the_check(torch_cuda):
    """set_agents(check=False) and a caller that never runs fo_sweep_check: the unsupported covariance must still
    show in cost / safe (fmax would otherwise drop the poisoned probabilities and report 'safe')"""
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    traj, agents = S.make_batch(130, 3, config_id=10)
    bad = {k: v.copy() for k, v in agents.items()}
    bad["cov"][2, :, 0, 1] = 0.9999 * bad["cov"][2, :, 0, 0]      # |rho| > 0.99: degenerate
    bad["cov"][2, :, 1, 0] = 0.9999 * bad["cov"][2, :, 0, 0]
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1)
    args = lambda a: (a["pos"], a["yaw"], a["v"], a["cov"], a["shape"], a["raw_dims"], a["type"], a["len"])
    sw.set_agents(*args(bad), check=False)
    out = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], traj["a"], mode="reduced")
    torch_cuda.cuda.synchronize()
    cost, safe = out.cost.cpu().numpy(), out.safe.cpu().numpy()
    assert not safe.any() and (cost[:, N.COST["safe"]] == 0.0).all()
    for name in ("max_collision_probability_all", "max_obst_risk_all", "max_ego_risk_all", "max_obst_harm_with_cp_all"):
        assert np.isnan(cost[:, N.COST[name]]).all(), name
    assert np.isfinite(cost[:, N.COST["min_dce"]]).all()        # geometry does not depend on the covariance
    # a clean set on the same context is not affected by the earlier one
    sw.set_agents(*args(agents), check=False)
    out = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], traj["a"], mode="reduced")
    torch_cuda.cuda.synchronize()
    assert np.isfinite(out.cost.cpu().numpy()[:, N.COST["max_collision_probability_all"]]).all()


def test_prediction_length_is_clamped_to_the_table(torch_cuda, oracle):
    """len[k] > Ta cannot be validated through the ABI (device array): the table build clamps it, so the result is
    the one of len = Ta and no row of the next agent is read"""
    from frenetix_occlusion import synthetic as S
    traj, agents = S.make_batch(96, 5, config_id=12)
    Ta = agents["pos"].shape[1]
    over = {k: v.copy() for k, v in agents.items()}
    over["len"][:] = Ta
    ref = _hip_sweep(torch_cuda, traj, over, S.VEHICLE_BMW320I, 0.1)
    over["len"][1], over["len"][4] = Ta + 7, 10 * Ta
    neg = over["len"].copy()
    got = _hip_sweep(torch_cuda, traj, over, S.VEHICLE_BMW320I, 0.1)
    for k in ("cost", "safe", "pair_f", "pair_i", "lists"):
        assert np.array_equal(ref[k], got[k], equal_nan=True), k
    neg[2] = -3                                                    # negative = unused slot, like 0
    over["len"] = neg
    zero = {k: v.copy() for k, v in over.items()}
    zero["len"][2] = 0
    a, b = (_hip_sweep(torch_cuda, traj, d, S.VEHICLE_BMW320I, 0.1) for d in (over, zero))
    for k in ("cost", "safe", "pair_f", "pair_i", "lists"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_reused_result_buffers_are_validated(torch_cuda):
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    traj, agents = S.make_batch(70, 4, config_id=13)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1)
    args = lambda a: (a["pos"], a["yaw"], a["v"], a["cov"], a["shape"], a["raw_dims"], a["type"], a["len"])
    sw.set_agents(*args(agents))
    run = lambda t, **kw: sw.run(t["x"], t["y"], t["theta"], t["v"], t["a"], **kw)
    out = run(traj, mode="full")
    again = run(traj, mode="full", out=out)                         # same shapes: reused in place
    assert again is out
    with pytest.raises(ValueError):
        run(traj, mode="reduced", out=out)                          # buffers the mode does not write
    with pytest.raises(ValueError):
        run({k: v[:50] for k, v in traj.items()}, mode="full", out=out)     # other M
    with pytest.raises(ValueError):
        run({k: v[:, :20] for k, v in traj.items()}, mode="full", out=out)  # other T
    fewer = {k: v[:3] for k, v in agents.items()}
    sw.set_agents(*args(fewer))
    with pytest.raises(ValueError):
        run(traj, mode="full", out=out)                             # other A
    with pytest.raises(ValueError):
        run(traj, mode="everything")
    torch_cuda.cuda.synchronize()


def test_horizon_split_variant_is_bit_identical(torch_cuda, monkeypatch):
    """FO_SWEEP_SPLIT=1 (four waves share an agent, a quarter of the horizon each) against =0 (one agent per wave):
    every output bit for bit, for full and short horizons, ragged predictions, every obstacle type"""
    from frenetix_occlusion import synthetic as S
    rng = np.random.default_rng(5)
    for M, A, T in ((130, 7, 31), (64, 3, 32), (200, 12, 10), (65, 5, 2), (300, 9, 17), (1, 1, 31)):
        traj = S.make_trajectories(M, T, 0.1, seed=100 + T)
        agents = S.make_agents(A, T, 0.1, seed=200 + T, lateral=3.0)
        agents["len"] = rng.integers(0, T + 1, A).astype(np.int32)
        agents["type"] = rng.integers(0, 11, A).astype(np.int32)
        outs = []
        # (FO_SWEEP_SPLIT_APW: agents a workgroup of the split form takes one after the other -- 1 unless set; a ragged last
        # workgroup with 3 and 5)
        for flag, per_wg in (("1", "1"), ("0", "1"), ("1", "2"), ("1", "3"), ("1", "5")):
            monkeypatch.setenv("FO_SWEEP_SPLIT", flag)
            monkeypatch.setenv("FO_SWEEP_SPLIT_APW", per_wg)
            outs.append(_hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.2, "risk": 0.1, "ttc": 1.0}))
            red = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.2, "risk": 0.1, "ttc": 1.0},
                             mode="reduced")
            assert np.array_equal(red["cost"], outs[-1]["cost"], equal_nan=True) and np.array_equal(red["safe"], outs[-1]["safe"])
            if per_wg == "1":      # the headline list format (float64 results stored as float32) in both forms: the float64 lists, rounded
                gx = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.2, "risk": 0.1, "ttc": 1.0}, lists="f32x")
                for k in ("cost", "safe", "pair_f", "pair_i"):
                    assert np.array_equal(gx[k], outs[-1][k], equal_nan=True), (M, A, T, flag, k)
                assert np.array_equal(gx["lists"], outs[-1]["lists"].astype(np.float32), equal_nan=True), (M, A, T, flag)
        for o in outs[1:]:
            for k in ("cost", "safe", "pair_f", "pair_i", "lists"):
                assert np.array_equal(outs[0][k], o[k], equal_nan=True), (M, A, T, k)


def test_correlated_covariances_against_the_oracle_and_the_diagonal_limit(torch_cuda, oracle):
    """full covariance matrices (real agents from a prediction module; the reference hands any matrix to mvnun,
    collision_probability.py:117): the numerical box integral of the kernel against the oracle's (itself pinned to the
    reference by tests/golden/correlated_cov.npz), mixed with diagonal agents in one batch, high correlations (the
    20- and 24-node rules of the correlation-angle integral), and the limit rho -> 0 against the closed form"""
    from frenetix_occlusion import synthetic as S
    rng = np.random.default_rng(4)
    traj, agents = S.make_batch(150, 6, config_id=31)
    T = agents["pos"].shape[1]
    for k, rho in enumerate((0.5, -0.97, 0.0, 0.85, -0.2, 0.985)):
        sx = np.sqrt(0.08 * 1.05 ** np.arange(T)) * (1.0 + 0.4 * k)
        sy = np.sqrt(0.15 * 1.03 ** np.arange(T))
        agents["cov"][k, :, 0, 0], agents["cov"][k, :, 1, 1] = sx * sx, sy * sy
        agents["cov"][k, :, 0, 1] = agents["cov"][k, :, 1, 0] = rho * sx * sy
    agents["pos"][:, :, :] = traj["x"][0, 12], traj["y"][0, 12]              # park them on the candidates' way
    agents["pos"] += rng.uniform(-2.0, 2.0, size=(6, 1, 2))
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1)
    got = _hip_sweep(torch_cuda, traj, agents, S.VEHICLE_BMW320I, 0.1)
    assert ref["lists"][:, :, oracle.LST["cp"], :].max() > 0.1               # the gate is really entered
    _compare(oracle, ref, got)
    # rho -> 0: the numerical integral meets the closed form of the diagonal case
    tiny = {k: v.copy() for k, v in agents.items()}
    diag = {k: v.copy() for k, v in agents.items()}
    tiny["cov"][:, :, 0, 1] = tiny["cov"][:, :, 1, 0] = 1e-13
    diag["cov"][:, :, 0, 1] = diag["cov"][:, :, 1, 0] = 0.0
    a, b = (_hip_sweep(torch_cuda, traj, d, S.VEHICLE_BMW320I, 0.1) for d in (tiny, diag))
    cpa, cpb = a["lists"][:, :, oracle.LST["cp"], :], b["lists"][:, :, oracle.LST["cp"], :]
    assert np.nanmax(np.abs(cpa - cpb)) < 1e-11


def test_correlated_and_diagonal_agent_sets_alternate_on_one_context(torch_cuda, oracle):
    """the sweep kernel picks its body from a generation-tagged device flag written by the agent preparation: a
    context that saw correlated agents must go back to the closed form -- bit for bit what a fresh context computes --
    when the next agent set is diagonal, and forth again"""
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import MetricSweep
    traj, diag = S.make_batch(130, 5, config_id=33)
    T = diag["pos"].shape[1]
    diag["pos"][:, :, :] = traj["x"][0, 10], traj["y"][0, 10]
    diag["pos"] += np.random.default_rng(8).uniform(-2.0, 2.0, size=(5, 1, 2))
    corr = {k: v.copy() for k, v in diag.items()}
    s = np.sqrt(corr["cov"][:, :, 0, 0] * corr["cov"][:, :, 1, 1])
    corr["cov"][:, :, 0, 1] = corr["cov"][:, :, 1, 0] = np.array([0.6, -0.8, 0.3, 0.95, -0.4])[:, None] * s
    keys = ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")
    fresh = _hip_sweep(torch_cuda, traj, diag, S.VEHICLE_BMW320I, 0.1)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1)
    outs = []
    for agents in (corr, diag, corr, diag, diag):
        sw.set_agents(*[agents[k] for k in keys])
        o = sw.run(traj["x"], traj["y"], traj["theta"], traj["v"], traj["a"], mode="full")
        torch_cuda.cuda.synchronize()
        outs.append((o.cost.cpu().numpy().copy(), o.lists.permute(3, 1, 0, 2).cpu().numpy().copy()))
    for i in (1, 3, 4):
        assert np.array_equal(outs[i][0], fresh["cost"]) and np.array_equal(outs[i][1], fresh["lists"], equal_nan=True)
    assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1], equal_nan=True)
    ref = oracle.sweep(traj, corr, S.VEHICLE_BMW320I, 0.1)
    f = np.isfinite(ref["lists"])
    np.testing.assert_allclose(outs[0][1][f], ref["lists"][f], rtol=0, atol=ATOL)
    assert np.abs(outs[0][1][f] - fresh["lists"][f]).max() > 1e-3        # and the correlation does change the numbers


def test_safety_decision_matches_the_references_metric_class(torch_cuda):
    """the `safe` flags of the sweep against what the reference's own Metric.evaluate_metrics decided (tests/golden/
    thresholds.npz; see tests/test_oracle_golden.py), in full and in reduced output mode"""
    from golden_util import load_threshold_case
    traj, agents, veh, dt, configs = load_threshold_case()
    for activated, thr, _, safe_ref in configs:
        t = {k: v for k, v in thr.items() if k in ("harm", "risk", "be", "cp", "ttc", "dce") and v is not None}
        for mode in ("full", "reduced"):
            got = _hip_sweep(torch_cuda, traj, agents, veh, dt, metrics=tuple(activated), thr=t, mode=mode)
            assert np.array_equal(got["safe"].astype(bool), safe_ref), (activated, thr, mode)


def test_library_loaded_before_torch_still_gets_the_device():
    """__graft_entry__.build() loads libfo_hip.so before anything imports torch; torch ships its own HIP runtime, and
    with two of them in one process fo_create used to fail (-3).  _native.load() therefore imports torch first."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from frenetix_occlusion import _native as N; N.load(); import torch; "
            "c = N.Context(0); print('ctx ok', N.load().fo_abi_version())" % os.path.join(root, "frenetix-occlusion_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ctx ok" in r.stdout, r.stderr[-2000:]
