"""One-off robustness run on the GPU box: the metric sweep (through the C ABI) against the oracle on random batch
shapes -- M, A, T, ragged prediction lengths, agent types, metric subsets, thresholds, output modes.  Reuses the
comparison of tests/test_sweep_gpu.py.  usage: python tools/sweep_fuzz.py [n] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from frenetix_occlusion import synthetic as SY  # noqa: E402
from oracle import fo_oracle as oracle  # noqa: E402
import test_sweep_gpu as TS  # noqa: E402

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle.build()
    worst, worst32, n_noise = 0.0, 0.0, 0
    for it in range(n):
        M = int(rng.choice([1, 2, 63, 64, 65, 127, 200, 513, 1000]))
        A = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 64]))
        T = int(rng.choice([2, 3, 7, 16, 17, 30, 31]))
        seed = int(rng.integers(1 << 30))
        traj = SY.make_trajectories(M, T, 0.1, seed=seed)
        agents = SY.make_agents(A, T, 0.1, seed=seed + 1, lateral=float(rng.choice([3.0, 14.0])))
        agents["len"] = rng.integers(0 if rng.random() < 0.2 else 1, T + 1, A).astype(np.int32)
        if rng.random() < 0.5:      # every ObstacleType code of the harm model's mass / protection tables
            agents["type"] = rng.integers(0, 12, A).astype(np.int32)
        if rng.random() < 0.35:     # full covariances on some agents: any correlation the sweep accepts, loose to tight
            cov = agents["cov"]
            for k in np.nonzero(rng.random(A) < 0.5)[0]:
                scale = float(rng.choice([0.05, 0.3, 1.0, 2.5]))
                rho = float(rng.uniform(-0.99, 0.99)) * (np.cos(0.04 * np.arange(T)) if rng.random() < 0.5 else np.ones(T))
                sx, sy = np.sqrt(cov[k, :, 0, 0]) * scale, np.sqrt(cov[k, :, 1, 1]) * scale * float(rng.uniform(0.5, 2.0))
                cov[k, :, 0, 0], cov[k, :, 1, 1] = sx * sx, sy * sy
                cov[k, :, 0, 1] = cov[k, :, 1, 0] = rho * sx * sy
        allm = ["hr", "ttc", "ttce", "dce", "wttc", "cp"]
        metrics = allm if rng.random() < 0.5 else list(rng.choice(allm, int(rng.integers(1, 6)), replace=False))
        thr = {"harm": float(rng.uniform(0.05, 1)), "risk": float(rng.uniform(0.01, 1)), "ttc": float(rng.uniform(0, 3)),
               "dce": float(rng.uniform(0, 1)), "cp": float(rng.uniform(0.1, 1))}
        if rng.random() < 0.3:
            thr = {k: v for k, v in thr.items() if rng.random() < 0.5}
        # both sweep variants: horizon split over the waves of a workgroup (what batches this small take by default)
        # and one agent per wave
        os.environ["FO_SWEEP_SPLIT"] = "1" if it % 2 == 0 else "0"
        ref = oracle.sweep(traj, agents, SY.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr, nthreads=8)
        got = TS._hip_sweep(torch, traj, agents, SY.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr)
        w = TS._compare(oracle, ref, got)
        assert np.array_equal(ref["safe"], got["safe"])
        worst = max(worst, w)
        if "lists" in got and ("cp" in metrics or "hr" in metrics) and T > 1:
            # float32 list storage: every other output bit-identical, the lists within 1e-6 of the oracle
            g32 = TS._hip_sweep(torch, traj, agents, SY.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr, lists="f32")
            for key in ("cost", "safe", "pair_i"):
                assert np.array_equal(g32[key], got[key]), key
            assert np.array_equal(g32["pair_f"], got["pair_f"], equal_nan=True)
            fin = np.isfinite(ref["lists"])
            assert np.array_equal(np.isnan(ref["lists"]), np.isnan(g32["lists"]))
            if fin.any():
                w32 = float(np.abs(ref["lists"][fin] - g32["lists"][fin].astype(np.float64)).max())
                assert w32 < 1e-6, w32
                worst32 = max(worst32, w32)
            # float32 STORAGE of the float64 results (the headline format; every form of the queue kernel since round 6 -- split /
            # full grid by the environment above, metric subsets by `metrics`): the float64 lists of the same form, rounded
            gx = TS._hip_sweep(torch, traj, agents, SY.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr, lists="f32x")
            for key in ("cost", "safe", "pair_i"):
                assert np.array_equal(gx[key], got[key]), key
            assert np.array_equal(gx["pair_f"], got["pair_f"], equal_nan=True)
            # (bit-equal -- except where an entry is rounding noise itself: a collision probability of 6e-17 is what nine
            # cancelling erf differences leave of an exact zero, and two instantiations of one source may fuse a multiply-add
            # differently; seen once in 92 batches, |difference| 1e-18.  Entries that differ must both be below 1e-15.)
            want32 = got["lists"].astype(np.float32)
            bad = ~((gx["lists"] == want32) | (np.isnan(gx["lists"]) & np.isnan(want32)))
            if bad.any():
                worst_noise = float(np.maximum(np.abs(gx["lists"][bad]), np.abs(want32[bad])).max())
                assert worst_noise < 1e-15, ("f32x != float64 lists rounded", it, M, A, T, metrics, os.environ["FO_SWEEP_SPLIT"],
                                              int(bad.sum()), worst_noise, np.argwhere(bad)[:4].tolist())
                n_noise += int(bad.sum())
        red = TS._hip_sweep(torch, traj, agents, SY.VEHICLE_BMW320I, 0.1, metrics=metrics, thr=thr, mode="reduced")
        assert np.array_equal(red["safe"], got["safe"])
        c1, c2 = red["cost"], got["cost"]
        assert np.array_equal(np.isfinite(c1), np.isfinite(c2)) and np.allclose(c1[np.isfinite(c1)], c2[np.isfinite(c2)], rtol=0, atol=1e-12)
        print(it, "M", M, "A", A, "T", T, "metrics", ",".join(metrics), "worst", f"{w:.2e}", "safe", float(ref["safe"].mean()), flush=True)
    print("all", n, "batches within 1e-9 of the oracle, integers exact; worst float deviation", f"{worst:.2e}",
          "; float32 lists: worst deviation", f"{worst32:.2e}", "; f32x entries that differ from the rounded float64 entry (all below 1e-15):", n_noise)


if __name__ == "__main__":
    main()
