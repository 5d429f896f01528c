"""The trajectory split of BASELINE configs[3] on the product path, on the GPU: ``Metric.evaluate_batch`` /
``FOInterface.trajectory_safety_assessment_batch`` / ``PlanningStep`` with ``shard=`` over ``distributed.CostGather``.

* world size 1 through RCCL (backend "nccl") in this process: the HIP sweep writes its cost rows into the collective's block,
  ``all_gather_into_tensor`` runs, results bit-identical to the unsharded call;
* two ranks on ONE GPU (what this box has), collective over gloo: block partition + HIP sweep per rank + one all-gather,
  bit-identical to the single-process result, the same trajectory picked on every rank;
* two ranks on two GPUs over RCCL: skipped unless ``torch.cuda.device_count() >= 2``.
Children are separate processes started with their RANK / WORLD_SIZE environment before they touch a GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def nccl_world1(torch_cuda):
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch_cuda.device("cuda", 0))
    yield dist
    if own:
        dist.destroy_process_group()


def test_world_size_one_nccl_sharded_sweep_is_bit_identical(torch_cuda, nccl_world1):
    torch = torch_cuda
    from frenetix_occlusion import distributed as D
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.metrics.metric import Metric
    assert nccl_world1.get_backend() == "nccl" and nccl_world1.get_world_size() == 1
    M, A = 1000, 24
    traj, agents = S.make_batch(M, A, config_id=31)

    class AM:       # the agent registry as far as Metric needs it
        dt = 0.1
        _manual = False
        _external = False

        def has_phantoms(self):
            return True

        def sweep_arrays(self):
            return [agents[k] for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")]

        predictions = {}

    cfg = {"activated_metrics": ["hr", "ttc", "ttce", "dce", "wttc", "cp"],
           "metric_thresholds": {"harm": 0.3, "risk": 0.2, "be": None, "cp": None, "ttc": None, "wttc": None, "ttce": None, "dce": None}}
    me = Metric(cfg, S.VEHICLE_BMW320I, AM())
    plain = me.evaluate_batch(traj, mode="pair")
    torch.cuda.synchronize()
    want = [t.cpu().numpy().copy() for t in (plain.cost, plain.safe, plain.result.pair_f, plain.result.pair_i)]
    cg = D.CostGather(M, force=True)         # world size 1, but THROUGH the collective (RCCL)
    assert cg.collective and cg.device.type == "cuda" and (cg.lo, cg.hi, cg.per) == (0, M, M)
    ptrs = (cg.mine.data_ptr(), cg.gathered.data_ptr())
    for i in range(3):
        ba = me.evaluate_batch(traj, mode="pair", shard=cg)
    torch.cuda.synchronize()
    assert cg.calls == 3 and ptrs == (cg.mine.data_ptr(), cg.gathered.data_ptr())
    assert ba.result.cost.data_ptr() == cg.mine.data_ptr()            # the sweep wrote into the collective's block
    assert ba.cost.data_ptr() == cg.gathered.data_ptr() and ba.rows == (0, M)
    got = [t.cpu().numpy() for t in (ba.cost, ba.safe, ba.result.pair_f, ba.result.pair_i)]
    for w, g in zip(want, got):
        assert np.array_equal(w, g, equal_nan=True)
    assert D.select_trajectory(ba.cost) == D.select_trajectory(plain.cost)
    # shard=True: the default group, no forced collective at world size 1 -- same numbers
    ba2 = me.evaluate_batch(traj, mode="reduced", shard=True)
    torch.cuda.synchronize()
    assert np.array_equal(ba2.cost.cpu().numpy(), want[0], equal_nan=True)
    with pytest.raises(ValueError):
        me.evaluate_batch(traj, mode="reduced", shard=D.CostGather(M + 1))


def _run_ranks(tmp_path, world, backend, M=333, extra=()):
    port = _free_port()
    out = str(tmp_path / f"dist_{backend}_{world}_{M}.npz")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--backend", backend,
                                       "--M", str(M), "--out", out, *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return np.load(out)


def _check(z, world, M):
    assert int(z["world"]) == world and int(z["n_agents"]) > 0
    assert z["cost_iface"].shape == (M, 16)
    assert np.array_equal(z["cost_iface"], z["ref_cost"], equal_nan=True)          # gathered == unsharded, bit for bit
    assert np.array_equal(z["safe_iface"], z["ref_safe"])
    assert np.array_equal(z["cost_step"], z["ref_cost_step"], equal_nan=True)
    assert np.array_equal(z["cost_step"], z["ref_cost"], equal_nan=True)           # the one-call step and the stage calls agree too
    assert int(z["pick"]) == int(z["ref_pick"])
    lo, hi = (int(q) for q in z["rows"])
    assert (lo, hi) == (0, -(-M // world))
    assert np.array_equal(z["local_pair"], z["ref_pair"][:, :, lo:hi], equal_nan=True)   # per-pair outputs: the rank's own rows


def test_two_ranks_on_one_gpu_shard_the_planning_step(torch_cuda, tmp_path):
    z = _run_ranks(tmp_path, 2, "gloo", M=333)
    _check(z, 2, 333)


@pytest.mark.parametrize("M", [10000, 10001])
def test_config3_partition_eight_ranks_on_one_gpu(torch_cuda, tmp_path, M):
    """BASELINE configs[3]'s exact partition on the GPU this box has: the bench line's 10 000 x 256 batch on the urban grid,
    EIGHT ranks (eight processes on the one GPU, each with its own context and HIP sweep; the collective over gloo -- RCCL
    refuses two ranks on one device), reduced outputs, the split inside ``PlanningStep(shard=...)``: the gathered cost
    [M, 16], the safe flags in it and the selected trajectory are bit-identical to the unsharded ``fo_step_run`` of rank 0;
    rank 7 holds rows 8 750 - 9 999.  M = 10 001: ceil(M / 8) = 1 251 rows per block, rank 7 holds 1 244 and its block is
    padded with NaN behind them; ``cost_all[:M]`` is clean.  What stays untested on one GPU: RCCL over xGMI with N > 1."""
    world = 8
    z = _run_ranks(tmp_path, world, "gloo", M=M, extra=("--config3",))
    assert int(z["world"]) == world and int(z["n_agents"]) == 256
    per = -(-M // world)
    assert [tuple(r) for r in z["rows"]] == [(min(r * per, M), min(r * per + per, M)) for r in range(world)]
    if M == 10000:
        assert tuple(z["rows"][7]) == (8750, 10000) and int(z["pad_rows"]) == 0
    else:
        assert tuple(z["rows"][7]) == (8757, 10001) and int(z["pad_rows"]) == 8 * 1251 - 10001 and bool(z["pad_is_nan"])
    assert z["cost_all"].shape == (M, 16)
    assert np.array_equal(z["cost_all"], z["ref_cost"], equal_nan=True)                     # gathered == unsharded, bit for bit
    assert np.array_equal(z["cost_all"][:, int(z["safe_col"])] > 0.5, z["ref_safe"].astype(bool))
    assert 0 < int(z["ref_safe"].sum()) < M
    assert int(z["pick"]) == int(z["ref_pick"]) >= 0


def test_two_ranks_two_gpus_rccl(torch_cuda, tmp_path):
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device); the one-GPU form of the same worker ran above")
    z = _run_ranks(tmp_path, 2, "nccl", M=333)
    _check(z, 2, 333)


def _multi_ego(world, extra=()):
    """tools/multi_ego_bench.py (BASELINE configs[4]: one fo_ctx / stream per ego, egos dealt round robin to the GPUs, no
    collective) as `world` processes, one per GPU; returns the JSON line of every rank"""
    import json
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "multi_ego_bench.py"), "--egos", "4", "--M", "256",
                                       "--A", "16", "--steps", "6", "--warmup", "3", *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    out = []
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, f"rank {r}:\n{e.decode(errors='replace')[-2000:]}"
        out.append(json.loads([ln for ln in o.decode().splitlines() if ln.startswith("{")][-1]))
    return out


def test_multi_ego_assignment_one_gpu(torch_cuda):
    """configs[4] on the GPU this box has: four egos, four contexts and streams, one process"""
    (j,) = _multi_ego(1)
    assert j["n_gpus"] == 1 and j["egos_on_this_gpu"] == 4 and len(j["phantoms_per_ego"]) == 4
    assert sum(j["phantoms_per_ego"]) > 0 and j["ms_per_step_all_egos"] > 0.0
    # sharing the GPU must not be slower than draining the egos one after the other
    assert j["ms_per_step_all_egos"] <= 1.5 * j["ms_per_step_egos_one_after_the_other"]


def test_multi_ego_assignment_two_gpus(torch_cuda):
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: egos 0, 2 on GPU 0 and 1, 3 on GPU 1")
    a, b = _multi_ego(2)
    assert (a["n_gpus"], b["n_gpus"]) == (2, 2) and {a["rank"], b["rank"]} == {0, 1}
    assert a["egos_on_this_gpu"] == 2 and b["egos_on_this_gpu"] == 2
