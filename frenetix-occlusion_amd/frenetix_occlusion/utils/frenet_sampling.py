"""The Frenet samples of a phantom vehicle (ref: utils/frenetix_handler.py:82-105, utils/sampling.py:156-163) and the one
the reference keeps (agent.py:349-379).

The reference asks the frenetix C++ sampler for nine trajectories per candidate route -- lateral targets
d1 in {-0.5, 0, 0.5} (absolute offsets from the route's centre line: ``delta_d_min / delta_d_max`` of
frenetix_handler.py:46-47 through ``LateralPositionSampling``) x end speeds {0.8, 1, 1.2} v0, all reached at t1 = 3 s
with zero lateral velocity / acceleration and zero longitudinal acceleration at both ends -- and keeps the one whose
Cartesian speed varies least.  ``sampling_matrix`` restates the matrix handed to the sampler row by row (pinned to the
reference's own ``generate_sampling_matrix`` by tests/golden/sampling_matrix.npz); ``select_min_var_v`` evaluates the
nine speed profiles on a straight route (quartic s(t), quintic d(t), v = sqrt(s'^2 + d'^2)) and applies the rule --
which always lands on "hold the speed, move to the nearest lateral target" (``nearest_lateral_target``), what
fo_spawn_predict_kernel and the oracle build directly.  frenetix itself is absent here: the trajectory shape on curved
routes (its own curvature terms) stays unpinned.
"""
import numpy as np

D1_TARGETS = (-0.5, 0.0, 0.5)          # LateralPositionSampling(delta_d_min=-0.5, delta_d_max=0.5, 1).to_range(0)
SS1_FACTORS = (0.8, 1.0, 1.2)          # frenetix_handler.py:85
T1 = 3.0                               # frenetix_handler.py:84


def sampling_matrix(s0, d0, v0):
    """[9, 13]: t0, t1, s0, ss0, sss0, ss1, sss1, d0, dd0, ddd0, d1, dd1, ddd1 -- end speed outermost, d1 innermost"""
    rows = []
    for f in SS1_FACTORS:
        for d1 in D1_TARGETS:
            rows.append([0.0, T1, s0, v0, 0.0, v0 * f, 0.0, d0, 0.0, 0.0, d1, 0.0, 0.0])
    return np.array(rows)


def lateral_profile(d0, d1, t):
    """quintic with zero velocity and acceleration at both ends: d(t), d'(t) for t in [0, T1] (held beyond)"""
    tau = np.minimum(np.asarray(t, dtype=np.float64) / T1, 1.0)
    d = d0 + (d1 - d0) * (tau * tau * tau * (10.0 + tau * (-15.0 + 6.0 * tau)))
    dd = (d1 - d0) * (30.0 * tau * tau * (1.0 + tau * (-2.0 + tau))) / T1
    return d, dd


def longitudinal_speed(v0, v1, t):
    """quartic s(t) with s'(0) = v0, s''(0) = 0, s'(T1) = v1, s''(T1) = 0: s'(t)"""
    tau = np.minimum(np.asarray(t, dtype=np.float64) / T1, 1.0)
    return v0 + (v1 - v0) * (3.0 * tau * tau - 2.0 * tau * tau * tau)


def nearest_lateral_target(d0):
    """the d1 of D1_TARGETS nearest to d0, the first of equally near ones (sampling order)"""
    best = D1_TARGETS[0]
    for d1 in D1_TARGETS[1:]:
        if abs(d1 - d0) < abs(best - d0):
            best = d1
    return best


def select_min_var_v(rows, dt=0.1):
    """index of the row whose Cartesian speed profile on a straight route has the smallest variance (agent.py:361-372,
    first minimum)"""
    t = np.arange(int(round(T1 / dt)) + 1) * dt
    best, arg = np.inf, -1
    for i, r in enumerate(rows):
        _, dd = lateral_profile(r[7], r[10], t)
        v = np.sqrt(longitudinal_speed(r[3], r[5], t) ** 2 + dd ** 2)
        var = float(np.var(v))
        if var < best:
            best, arg = var, i
    return arg
