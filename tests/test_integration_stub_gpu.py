"""The reference-side binding printed in INTEGRATION.md §B is executed as it stands (only the library path is made
absolute) against mock agent-manager / trajectory objects, and its cost vectors are compared with the oracle: the
documented stub must keep working."""
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_binding_runs_and_matches_the_oracle(oracle):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    part_b = text[text.index("## B. Keep the reference package"):]
    code = re.search(r"```python\n(.*?)```", part_b, re.S).group(1)
    lib = os.path.join(ROOT, "frenetix-occlusion_amd", "lib", "libfo_hip.so")
    assert '"libfo_hip.so"' in code
    ns = {}
    exec(compile(code.replace('"libfo_hip.so"', repr(lib)), "INTEGRATION.md#B", "exec"), ns)
    ns["lib"].fo_last_error.restype = ns["C"].c_char_p

    import sys
    sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sweep import DEFAULT_HARM_COEFF
    traj, agents = S.make_batch(96, 5, config_id=77)
    names = {0: "Car", 1: "Truck", 2: "Bus", 3: "Bicycle", 4: "Pedestrian"}
    agents["type"] = np.array([0, 4, 3, 1, 4], dtype=np.int32)
    preds, objs = {}, {}
    for k in range(5):
        L = int(agents["len"][k])
        pid = int(str(10000 + k) + "0")
        preds[pid] = {"pos_list": agents["pos"][k, :L], "v_list": agents["v"][k, :L], "orientation_list": agents["yaw"][k, :L],
                      "cov_list": agents["cov"][k, :L], "shape": {"length": agents["shape"][k, 0], "width": agents["shape"][k, 1]}}
        objs[pid] = SimpleNamespace(agent_type=names[int(agents["type"][k])],
                                    shape=SimpleNamespace(length=agents["raw_dims"][k, 0], width=agents["raw_dims"][k, 1]))
    am = SimpleNamespace(predictions=preds, agent_by_prediction_id=lambda pid: objs[pid])
    veh = SimpleNamespace(**dict(zip(("length", "width", "wb_rear_axle", "mass", "a_max"), S.VEHICLE_BMW320I)))
    cfg = {"activated_metrics": ["hr", "ttc", "ttce", "dce", "wttc", "cp"],
           "metric_thresholds": {"harm": 0.1, "risk": 1, "be": None, "cp": None, "ttc": None, "dce": None}}
    hs = ns["HipSweep"](veh, 0.1, cfg, dict(DEFAULT_HARM_COEFF))
    hs.set_agents(am)
    tr = [SimpleNamespace(cartesian=SimpleNamespace(**{q: traj[q][m] for q in ("x", "y", "theta", "v")})) for m in range(96)]
    cost, safe = hs.evaluate(tr)
    torch.cuda.synchronize()
    ref = oracle.sweep(traj, agents, S.VEHICLE_BMW320I, 0.1, thr={"harm": 0.1, "risk": 1})
    got = cost.cpu().numpy()
    f = np.isfinite(ref["cost"])
    assert np.array_equal(f, np.isfinite(got))
    np.testing.assert_allclose(got[f], ref["cost"][f], rtol=0, atol=1e-9)
    assert np.array_equal(safe.cpu().numpy(), ref["safe"].astype(bool))
