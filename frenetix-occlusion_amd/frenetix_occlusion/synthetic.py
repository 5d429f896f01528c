"""Synthetic workloads of the shapes BASELINE.json names (SURVEY.md §8d "Value distributions / seeds").

This is the harness side of the path: what the external planner would hand over --
candidate Frenet-style trajectories -- and phantom-agent predictions of the form agent.py:398-426,520-536 emits.
numpy only; seeds are ``default_rng(20240131 + config_id)``.
"""
import numpy as np

# BMW 320i = CommonRoad vehicle 2 (configurations/simulation/vehicle.yaml:2): length, width, wb_rear_axle, mass, a_max
VEHICLE_BMW320I = (4.508, 1.610, 1.4227, 1093.3, 11.5)

# agent_manager section of the YAML (config/config.yaml:66-92): raw dims, default speed; prediction inflation factors
AGENT_DIMS = {"car": (4.8, 2.0), "truck": (9.0, 2.5), "bicycle": (2.0, 0.9), "pedestrian": (0.3, 0.5)}
AGENT_SPEED = {"car": 10.0, "truck": 10.0, "bicycle": 5.0, "pedestrian": 1.4}
TYPE_CODE = {"car": 0, "truck": 1, "bicycle": 3, "pedestrian": 4}
INFLATE_S, INFLATE_L = (1.2, 1.3), (1.4, 2.5)  # size_factor_{length,width}_{s,l}; bicycles use _l (agent.py:402-405)


def make_trajectories(M, T=31, dt=0.1, seed=20240131, ego_pos=(0.0, 0.0), ego_yaw=0.0, order="sampler"):
    """M candidate trajectories from one ego pose: v0 ~ U(3,12), end speed v0*U(0.3,1.3), lateral target U(-3,3),
    quintic blend over the horizon.  Returns dict of float64 [M,T] arrays x,y,theta,v,a.

    order="sampler": the draws are emitted the way a Frenet sampler emits its candidates -- nested loops, here
    (start-speed bucket, end-speed bucket, lateral target) like the sampling matrix of utils/frenetix_handler.py:82-105
    -- so neighbouring rows are neighbouring trajectories.  order="random": i.i.d. order (worst case for a wave)."""
    rng = np.random.default_rng(seed)
    v0 = rng.uniform(3.0, 12.0, (M, 1))
    v1 = v0 * rng.uniform(0.3, 1.3, (M, 1))
    d1 = rng.uniform(-3.0, 3.0, (M, 1))
    if order == "sampler":
        nb = max(1, int(round(M ** (1.0 / 3.0))))
        b0 = np.minimum((v0[:, 0] - 3.0) / 9.0 * nb, nb - 1).astype(np.int64)
        b1 = np.minimum(v1[:, 0] / (12.0 * 1.3) * nb, nb - 1).astype(np.int64)
        idx = np.lexsort((d1[:, 0], b1, b0))
        v0, v1, d1 = v0[idx], v1[idx], d1[idx]
    tau = np.linspace(0.0, 1.0, T)[None, :]
    blend = 10 * tau ** 3 - 15 * tau ** 4 + 6 * tau ** 5
    dblend = (30 * tau ** 2 - 60 * tau ** 3 + 30 * tau ** 4) / ((T - 1) * dt)
    v = v0 + (v1 - v0) * blend
    a = (v1 - v0) * dblend
    s = np.concatenate((np.zeros((M, 1)), np.cumsum(0.5 * (v[:, 1:] + v[:, :-1]) * dt, axis=1)), axis=1)
    d = d1 * blend
    dd_dt = d1 * dblend
    c, sn = np.cos(ego_yaw), np.sin(ego_yaw)
    x = ego_pos[0] + c * s - sn * d
    y = ego_pos[1] + sn * s + c * d
    theta = ego_yaw + np.arctan2(dd_dt, np.maximum(v, 1e-3))
    return {"x": x, "y": y, "theta": theta, "v": v, "a": a}


def predictions_from_spawns(pos0, yaw, kind_names, T=31, dt=0.1, speed=None, var0=0.1, var_factor=1.05):
    """Constant-velocity straight-line predictions (agent.py:451-536): velocity components rounded to 3 decimals
    (Q12), cov_k = 0.1 * 1.05^k * I (agent.py:260-280), inflated shape in 'shape', raw dims for DCE."""
    A = len(kind_names)
    pos0 = np.asarray(pos0, dtype=np.float64).reshape(A, 2)
    yaw = np.asarray(yaw, dtype=np.float64).reshape(A)
    spd = np.array([AGENT_SPEED[k] for k in kind_names]) if speed is None else np.asarray(speed, dtype=np.float64)
    vx, vy = np.round(spd * np.cos(yaw), 3), np.round(spd * np.sin(yaw), 3)
    t = (np.arange(T) * dt)[None, :, None]
    pos = pos0[:, None, :] + t * np.stack((vx, vy), -1)[:, None, :]
    var = var0 * var_factor ** np.arange(T)
    cov = np.zeros((A, T, 2, 2))
    cov[:, :, 0, 0] = var
    cov[:, :, 1, 1] = var
    raw = np.array([AGENT_DIMS[k] for k in kind_names])
    infl = np.array([INFLATE_L if k == "bicycle" else INFLATE_S for k in kind_names])
    return {"pos": pos, "yaw": np.repeat(yaw[:, None], T, 1), "v": np.repeat(spd[:, None], T, 1), "cov": cov,
            "shape": raw * infl, "raw_dims": raw, "type": np.array([TYPE_CODE[k] for k in kind_names], dtype=np.int32),
            "len": np.full(A, T, dtype=np.int32)}


def make_agents(A, T=31, dt=0.1, seed=20240131, ego_yaw=0.0, ego_pos=(0.0, 0.0), ahead=(6.0, 45.0), lateral=14.0):
    """A phantom predictions placed in the corridor ahead of the ego (stand-in for 'uniform over the occluded cells
    within 40 m ahead'): 50 % pedestrians, 25 % bicycles, 25 % cars; heading = corridor heading + {0, +-pi/2, pi}.
    With the default corridor 15-22 % of the (trajectory, agent) pairs pass the 5 m CP gate at some timestep and
    4-8 % collide (SURVEY 8d asks for >= 10 % and >= 1 %)."""
    rng = np.random.default_rng(seed + 7919)
    kinds = ["pedestrian", "pedestrian", "bicycle", "car"]
    names = [kinds[i % 4] for i in range(A)]
    s = rng.uniform(ahead[0], ahead[1], A)
    d = rng.uniform(-lateral, lateral, A)
    c, sn = np.cos(ego_yaw), np.sin(ego_yaw)
    p0 = np.stack((ego_pos[0] + c * s - sn * d, ego_pos[1] + sn * s + c * d), -1)
    yaw = ego_yaw + rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi], A) + rng.normal(0.0, 0.05, A)
    # pedestrians walk towards the lane centre
    for i, n in enumerate(names):
        if n == "pedestrian":
            yaw[i] = ego_yaw + (-np.pi / 2 if d[i] > 0 else np.pi / 2) + rng.normal(0.0, 0.1)
    return predictions_from_spawns(p0, yaw, names, T, dt)


def make_batch(M, A, T=31, dt=0.1, config_id=0):
    seed = 20240131 + config_id
    return make_trajectories(M, T, dt, seed), make_agents(A, T, dt, seed)
