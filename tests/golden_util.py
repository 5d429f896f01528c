"""Helpers shared by the golden-vector tests: load a fixture and turn it into sweep inputs."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TYPE_CODES = {"car": 0, "truck": 1, "bus": 2, "bicycle": 3, "pedestrian": 4, "priorityvehicle": 5, "parkedvehicle": 6,
              "train": 7, "motorcycle": 8, "taxi": 9, "unknown": 10}      # = frenetix_occlusion._native.TYPE_CODES
CASES = ["probe_ped_crossing", "random_equal_len", "random_ragged", "short_traj", "ring_all_types", "correlated_cov", "angle_bins"]


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    traj = {k: g["traj_" + k] for k in ("x", "y", "theta", "v", "a")}
    agents = {"pos": g["agent_pos"], "yaw": g["agent_yaw"], "v": g["agent_v"], "cov": g["agent_cov"],
              "shape": g["agent_shape"], "raw_dims": g["agent_raw_dims"],
              "type": np.array([TYPE_CODES[str(t).lower()] for t in g["agent_type"]], dtype=np.int32),
              "len": g["agent_len"].astype(np.int32)}
    return g, traj, agents, tuple(g["vehicle"]), float(g["dt"])


def load_threshold_case():
    """tests/golden/thresholds.npz: (inputs, [(activated metrics, thresholds dict, evaluated metric names, safe[M])...])
    produced by the reference's own Metric.evaluate_metrics (gen_golden.py case 7)"""
    g = dict(np.load(os.path.join(GOLDEN, "thresholds.npz"), allow_pickle=False))
    traj = {k: g["traj_" + k] for k in ("x", "y", "theta", "v", "a")}
    agents = {"pos": g["agent_pos"], "yaw": g["agent_yaw"], "v": g["agent_v"], "cov": g["agent_cov"],
              "shape": g["agent_shape"], "raw_dims": g["agent_raw_dims"],
              "type": np.array([TYPE_CODES[str(t).lower()] for t in g["agent_type"]], dtype=np.int32),
              "len": g["agent_len"].astype(np.int32)}
    names = [str(n) for n in g["thr_names"]]
    configs = []
    for c in range(int(g["n_configs"])):
        thr = {n: (None if np.isnan(v) else float(v)) for n, v in zip(names, g[f"cfg{c}_thr"])}
        configs.append(([str(m) for m in g[f"cfg{c}_activated"]], thr, sorted(str(m) for m in g[f"cfg{c}_evaluated"]),
                        g[f"cfg{c}_safe"].astype(bool)))
    return traj, agents, tuple(g["vehicle"]), float(g["dt"]), configs


def load_be_case():
    """tests/golden/be_bisection.npz (gen_golden.py be): what the reference's own BE class returned -- required constant
    deceleration and brake threat number per (trajectory, prediction) -- for 40 candidates x 10 predictions; returns
    (golden dict, traj, agents, vehicle tuple, dt) in the oracle's input layout"""
    g = dict(np.load(os.path.join(GOLDEN, "be_bisection.npz"), allow_pickle=False))
    traj = {k: g["traj_" + k] for k in ("x", "y", "theta", "v", "a")}
    agents = {k: g["agent_" + k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "len")}
    agents["type"] = g["agent_type_code"]
    return g, traj, agents, tuple(float(q) for q in g["vehicle"]), float(g["dt"])


def load_dce_case():
    """tests/golden/dce_loop.npz (gen_golden.py dce): what the reference's own DCE class -- its walk over the time steps -- and
    its TTC / TTCE / WTTC on top returned for 31 candidates x 16 predictions (ragged, overlapping, touching, stationary,
    symmetric); the polygon distance under the loop is this repository's rectangle distance"""
    g = dict(np.load(os.path.join(GOLDEN, "dce_loop.npz"), allow_pickle=False))
    traj = {k: g["traj_" + k] for k in ("x", "y", "theta", "v", "a")}
    agents = {k: g["agent_" + k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "len")}
    agents["type"] = g["agent_type_code"]
    return g, traj, agents, tuple(float(q) for q in g["vehicle"]), float(g["dt"])
