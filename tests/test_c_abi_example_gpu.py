"""The C ABI without Python or torch in the process: examples/sweep_from_c.c is compiled as plain C11 with gcc against
include/fo_hip.h + libfo_hip.so, run on the GPU, and its printed cost vectors are compared with the oracle on the same
inputs (restated here in numpy)."""
import math
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs():
    M, T, dt = 64, 31, 0.1
    t = np.arange(T)
    m = np.arange(M)
    psi = -0.2 + 0.4 * m / (M - 1)
    sp = 6.0 + 6.0 * ((m * 7) % M) / (M - 1)
    traj = {"x": sp[:, None] * dt * t[None, :] * np.cos(psi)[:, None], "y": sp[:, None] * dt * t[None, :] * np.sin(psi)[:, None],
            "theta": np.repeat(psi[:, None], T, 1), "v": np.repeat(sp[:, None], T, 1), "a": np.zeros((M, T))}
    pos = np.zeros((2, T, 2))
    pos[0, :, 0], pos[0, :, 1] = 15.0, -3.0 + 1.4 * dt * t
    pos[1, :, 0], pos[1, :, 1] = 22.0 + 3.0 * dt * t, 0.4
    var = 0.1 * np.power(1.05, t)
    cov = np.zeros((2, T, 2, 2))
    cov[:, :, 0, 0] = cov[:, :, 1, 1] = var
    agents = {"pos": pos, "yaw": np.stack([np.full(T, 1.5707963267948966), np.zeros(T)]),
              "v": np.stack([np.full(T, 1.4), np.full(T, 3.0)]), "cov": cov,
              "shape": np.array([[0.6, 0.65], [5.4, 2.34]]), "raw_dims": np.array([[0.5, 0.5], [4.5, 1.8]]),
              "type": np.array([4, 0], dtype=np.int32), "len": np.array([T, T], dtype=np.int32)}
    return traj, agents, dt


def test_plain_c_host_gets_the_oracles_numbers(oracle, tmp_path):
    exe = str(tmp_path / "sweep_from_c")
    lib = os.path.join(ROOT, "frenetix-occlusion_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-O2", os.path.join(ROOT, "examples", "sweep_from_c.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-L" + lib, "-lfo_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lm",
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = np.array([[float(v) for v in line.split()] for line in r.stdout.strip().splitlines()])
    assert rows.shape == (64, 7) and np.array_equal(rows[:, 0], np.arange(64))
    traj, agents, dt = _inputs()
    ref = oracle.sweep(traj, agents, (4.508, 1.610, 1.4227, 1093.3, 11.5), dt, thr={"harm": 0.1, "risk": 1.0})
    C = oracle.COST
    assert np.array_equal(rows[:, 1].astype(bool), ref["safe"].astype(bool))
    for col, name in ((2, "wttc"), (3, "min_dce"), (4, "max_obst_risk_all"), (5, "max_obst_harm_all"),
                      (6, "max_collision_probability_all")):
        a, b = ref["cost"][:, C[name]], rows[:, col]
        assert np.array_equal(np.isinf(a), np.isinf(b)), name
        f = np.isfinite(a)
        np.testing.assert_allclose(b[f], a[f], rtol=0, atol=1e-9, err_msg=name)
    assert 0 < ref["safe"].sum() < 64 and np.isfinite(ref["cost"][:, C["wttc"]]).any()   # the case is not trivial
