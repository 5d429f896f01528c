"""The NumPy restatement in the reference's loop structure (oracle/fo_numpy_ref.py, CPU baseline B0 of the bench line)
equals the C oracle -- so the number bench.py prints for it is the cost of the same computation."""
import numpy as np


def test_numpy_baseline_equals_c_oracle(oracle):
    from frenetix_occlusion import synthetic as S
    from oracle import fo_numpy_ref as R
    traj, agents = S.make_batch(12, 7, config_id=4)
    agents["len"] = np.array([31, 5, 0, 31, 17, 31, 2], dtype=np.int32)
    thr = {"harm": 0.1, "risk": 0.05, "ttc": 1.0, "dce": 0.2}
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr)
    res, safe = R.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr=thr)
    assert np.array_equal(safe, ref["safe"])
    PF, PI, L = oracle.PF, oracle.PI, oracle.LST
    worst = 0.0
    for m, r in enumerate(res):
        for k in range(7):
            if agents["len"][k] == 0:
                assert k not in r["dce"]
                continue
            assert r["dce"][k]["dce"] == ref["pair_f"][m, k, PF["dce"]]
            assert r["dce"][k]["time_dce"] == ref["pair_i"][m, k, PI["time_dce"]]
            assert r["ttc"][k] == ref["pair_f"][m, k, PF["ttc"]] and r["ttce"][k] == ref["pair_f"][m, k, PF["ttce"]]
            cp = ref["lists"][m, k, L["cp"]]
            worst = max(worst, np.abs(r["cp"][k] - cp).max())
            h = r["hr"][k]
            n = len(h["ego_harm_traj"])
            for name, key in (("ego_harm", "ego_harm_traj"), ("obst_harm", "obst_harm_traj"), ("ego_risk", "ego_risk_traj"),
                              ("obst_risk", "obst_risk_traj")):
                worst = max(worst, np.abs(np.asarray(h[key]) - ref["lists"][m, k, L[name], :n]).max())
                assert np.isnan(ref["lists"][m, k, L[name], n:]).all()
            for name in ("max_ego_risk", "max_obst_risk", "max_obst_harm_with_cp", "max_ego_harm", "max_obst_harm",
                         "max_collision_probability"):
                worst = max(worst, abs(h[name] - ref["pair_f"][m, k, PF[name]]))
        for name in ("max_ego_risk_all", "max_obst_risk_all", "max_obst_harm_with_cp_all", "max_collision_probability_all"):
            worst = max(worst, abs(r["hr"][name] - ref["cost"][m, oracle.COST[name]]))
        assert r["wttc"] == ref["cost"][m, oracle.COST["wttc"]]
    assert worst < 1e-12, worst
