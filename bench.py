#!/usr/bin/env python3
"""Benchmark of the Frenetix-Occlusion per-planning-step hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode full|pair|reduced] [--lists f32x|f32|f64] [--M 10000] [--A 256]

One "step" = one planning step of BASELINE.json configs[2] with every input already resident in HBM:
    visibility ray fan (720 rays @ 0.5 deg, r = 50 m) over the synthetic urban lanelet net (~9.4e3 boundary edges,
    64 parked cars)  ->  cell classes + occluded-cell list (0.5 m cells)  ->  256 phantom agents sampled in the
    occluded cells + their predictions  ->  agent table  ->  trajectory x agent metric sweep (DCE / TTC / TTCE / WTTC /
    CP / harm / risk over T = 31) for 10 000 candidate trajectories  ->  threshold reduction
    (+ one RCCL all-gather of the per-trajectory cost vectors when N > 1).
`roofline.bound` names the binding unit from the committed rocprofv3 counters of the loaded library (`profiles/`): the larger
of `valu_issue_frac` (VALU issue slots taken) and `hbm_frac_of_achievable` (counter traffic over the 6.3 TB/s the chip reaches)
when that one exceeds 0.8, else "latency/issue mix (hbm x, valu y)" -- short of both limits at once.

N > 1 (BASELINE configs[3]): ONE batch of --M trajectories is block-partitioned over the ranks (`--scaling strong`, the
default; M/N per rank, scene stage and agents replicated, `cost [M][16]` all-gathered) -- `--scaling weak` gives every
rank its own --M trajectories instead.  `bench.py --gpus N` without a launcher starts its own N worker processes (one
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) before anything touches a GPU and
exits with their status; under `torch.distributed.run` it uses the environment it is given.

Output lists: `--lists f32x` (default) computes every list entry in float64 like the reference and stores it as float32 --
the storage SURVEY 8d prices (648 B per pair); `dtype` says "f64 (lists stored f32)".  `--lists f32` also evaluates the harm
entries away from the 5 m gate in float32 ARITHMETIC (`dtype` "f64+f32lists": narrower than the reference, reported as the
side leg `config.f32_lists`, never as the headline); `--lists f64` stores the lists as float64 like the reference's numpy
arrays.  All arithmetic that reaches a cost vector, a flag or a per-pair scalar is float64 in all three.

Set-up (untimed, before the W warm-up steps): the library measures the sweep kernel's four agents-per-wave settings on the
batch (`fo_sweep_autotune`, part of the C ABI: any caller gets the same choice) and keeps the best for the shape -- which
also brings the GPU to its sustained clocks, so that the timed K steps do not depend on W (`--no-autotune` skips it).
`--scene scenario1 --M 2000 --A 32` runs BASELINE configs[1] instead.  Beside the headline the default run measures
(`config.*`): the other list formats, reduced outputs, `shard_probe` (the step of a 1/8, 1/4, 1/2 shard of the batch and
the RCCL all-gather of one cost block, on this one GPU), `small_batch` (configs[1]) and `rules_step` (the reference's
own spawn rule families as a device-resident step).

Prints ONE JSON line on rank 0.  `value` = trajectory x agent metric evaluations per second over all ranks.
`roofline` follows SURVEY 8d: algorithmic bytes = 620 B per trajectory + 636 B per agent + 648 B per pair (full
outputs, fp32 storage) over the HIP-event duration of the sweep kernel; `frac_stored_bytes` is the same on the bytes
this build really moves (float64 inputs and pair scalars).  `parity` compares the GPU buffers of the last timed step
with the CPU oracle on EVERY pair of the batch; `cpu_baseline` holds the three CPU figures of SURVEY 8d (NumPy in the
reference's loop structure on one thread, the C port on one thread and on all cores).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
THR = {"harm": 0.1, "risk": 1}   # configurations/simulation/occlusion.yaml:20-28 of the reference's example


def bytes_8d(M, A, T, mode):
    """SURVEY 8d, fp32 storage: trajectory 5 arrays x T x 4 B, agent prediction (x, y, psi, v, var) x T x 4 B + 16 B,
    outputs 64 B per trajectory (reduced) / 48 B per pair (pair scalars) / 48 + 5 (T-1) 4 B per pair (full)."""
    b = M * 5 * T * 4 + A * (5 * T * 4 + 16)
    if mode == "reduced":
        b += M * 64
    if mode in ("pair", "full"):
        b += M * A * 48
    if mode == "full":
        b += M * A * 5 * (T - 1) * 4
    return b


def bytes_stored(M, A, T, mode, lists):
    """what this build moves: the float64 tile table the sweep kernel reads (x, y, cos, sin, theta, v, v cos, v sin per
    sample), the prepared agent rows (96 B) + constants, float64 pair scalars + int32 indices, lists in `lists`"""
    b = M * T * 8 * 8 + A * (T * 12 * 8 + 8 * 8)
    if mode == "reduced":
        b += M * 16 * 8
    if mode in ("pair", "full"):
        b += M * A * (12 * 8 + 4 * 4)
    if mode == "full":
        b += M * A * 5 * (T - 1) * (4 if lists in ("f32", "f32x") else 8)     # f32x: float64 results, float32 elements
    return b


def usable_cores():
    """cores this process may really use: cgroup quota if there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------------------------------------ N > 1 without a launcher
def launch_workers(n, argv):
    """`bench.py --gpus N` called directly: start N copies of this script, one per GPU, and exit with their status.
    The parent never imports torch and never touches a GPU (a process that has must not be replaced or forked)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc, deadline = 0, time.time() + float(os.environ.get("FO_BENCH_LAUNCH_TIMEOUT", "1500"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            c = p.poll()
            if c is not None:
                alive.remove(p)
                rc = rc or c
        if (rc or time.time() > deadline) and alive:      # one worker failed (or the job hangs): end exactly our children
            for p in alive:
                p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def launcher_selftest(args):
    """CPU check of the N > 1 plumbing (tests/test_bench_launcher_cpu.py): the workers rendezvous over gloo on
    127.0.0.1, take their block of ONE --M batch (strong scaling), all-gather a cost matrix whose rows are a function of
    the global trajectory index, and rank 0 prints a bench-shaped JSON line.  No GPU, no kernel, no oracle."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from frenetix_occlusion.distributed import CostGather, shard_bounds
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M, NC = args.M, 16
    cg = CostGather(M, device="cpu")          # the product's split: bounds, pre-allocated blocks, the one collective
    assert cg.world == world and cg.rank == rank
    lo, hi, per = cg.lo, cg.hi, cg.per
    rows = torch.arange(lo, hi, dtype=torch.float64)
    cg.block().copy_(rows[:, None] * 16.0 + torch.arange(NC, dtype=torch.float64)[None, :])
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cg.gather()
    dist.barrier()
    el = time.perf_counter() - t0
    # rank r's block sits at rows [r per, r per + its length): put the blocks back to back
    gathered = cg.gathered
    parts = [gathered[r * per: r * per + (shard_bounds(M, world, r)[1] - shard_bounds(M, world, r)[0])] for r in range(world)]
    full = torch.cat(parts)
    want = torch.arange(M, dtype=torch.float64)[:, None] * 16.0 + torch.arange(NC, dtype=torch.float64)[None, :]
    ok = full.shape == want.shape and bool((full == want).all())
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"metric": "launcher_selftest", "n_gpus": world, "ranks_seen": dist.get_world_size(),
                          "scaling": args.scaling or "strong", "M_total": M, "M_per_rank": per, "steps": args.steps,
                          "gather_ok": bool(flag.item() == 1.0), "ms_per_step": el / max(args.steps, 1) * 1e3}), flush=True)
    dist.destroy_process_group()
    return 0 if flag.item() == 1.0 else 1


# ------------------------------------------------------------------------------------------------ CPU figures + parity
def cpu_and_parity(S, N, traj, agents, out, lists_fmt, want_parity=True):
    """Runs the oracle (oracle/fo_oracle.c, a C port of the reference's per-trajectory loops) over the WHOLE batch in
    chunks on every usable core -- timing only the oracle calls (CPU baseline B2) and comparing every chunk with the
    GPU buffers of the last timed step (parity) -- then the same port on one thread (B1) and the NumPy restatement in
    the reference's loop structure on one thread (B0) on bounded samples."""
    import numpy as np
    import torch
    from oracle import fo_compare as CMP   # checker / baseline only
    from oracle import fo_numpy_ref as R
    from oracle import fo_oracle as O
    O.build()
    threads = usable_cores()
    M, A = traj["x"].shape[0], agents["pos"].shape[0]
    A_act = int((agents["len"] > 0).sum())
    full = out is not None and out.lists_raw is not None and want_parity
    views = out.list_views() if full else None
    chunk = 500
    t_or, n_or, par, bufs = 0.0, 0, None, None
    for lo in range(0, M, chunk):
        hi = min(lo + chunk, M)
        sub = {k: v[lo:hi] for k, v in traj.items()}
        same = bufs is not None and bufs["cost"].shape[0] == hi - lo
        t0 = time.perf_counter()
        ref = O.sweep(sub, agents, S.VEHICLE_BMW320I, 0.1, thr=THR, want_lists=True, nthreads=threads, out=bufs if same else None)
        dt = time.perf_counter() - t0
        if same or M <= chunk:          # the first pass over fresh buffers is page faults, not arithmetic
            t_or, n_or = t_or + dt, n_or + (hi - lo)
        bufs = ref
        if full:
            got = {"cost": out.cost[lo:hi].cpu().numpy(), "safe": out.safe[lo:hi].cpu().numpy(),
                   "pair_f": out.pair_f[:, :, lo:hi].permute(2, 1, 0).cpu().numpy(),
                   "pair_i": out.pair_i[:, :, lo:hi].permute(2, 1, 0).cpu().numpy(),
                   "lists": torch.stack([v[:, :, lo:hi] for v in views]).permute(3, 1, 0, 2).cpu().numpy()}
            par = CMP.merge(par, CMP.compare(ref, got))
    if n_or == 0:
        t_or, n_or = dt, hi - lo
    b2 = n_or * A_act / t_or
    # B1: the same C port, one thread, ~5 s
    m1 = int(min(M, max(8, b2 / threads * 5.0 / max(A_act, 1))))
    sub = {k: v[:m1] for k, v in traj.items()}
    O.sweep(sub, agents, S.VEHICLE_BMW320I, 0.1, thr=THR, want_lists=True, nthreads=1)
    t0 = time.perf_counter()
    O.sweep(sub, agents, S.VEHICLE_BMW320I, 0.1, thr=THR, want_lists=True, nthreads=1)
    b1 = m1 * A_act / (time.perf_counter() - t0)
    # B0: NumPy in the reference's loop structure, one thread, trajectories spread over the batch x the first 32 active
    # agents, sized for ~10 s by a short probe
    act = np.flatnonzero(agents["len"] > 0)[:32]
    ag0 = {k: v[act] for k, v in agents.items()}
    pick = lambda n: {k: v[np.linspace(0, M - 1, n).astype(np.int64)] for k, v in traj.items()}
    t0 = time.perf_counter()
    R.sweep(pick(2), ag0, S.VEHICLE_BMW320I, 0.1, thr=THR)
    probe = 2 * len(act) / max(time.perf_counter() - t0, 1e-6)
    m0 = int(min(M, 200, max(4, probe * 10.0 / max(len(act), 1))))
    t0 = time.perf_counter()
    R.sweep(pick(m0), ag0, S.VEHICLE_BMW320I, 0.1, thr=THR)
    b0 = m0 * len(act) / (time.perf_counter() - t0)
    cpu = {"value": b2, "unit": "pair-evals/s", "cores": threads, "kind": "port",
           "sample": f"{n_or} trajectories x {A_act} active agents of the same batch (the phantom set the GPU step produced), "
                     f"full outputs, oracle/fo_oracle.c, OpenMP over trajectories on {threads} threads, {t_or:.2f} s",
           "b0_numpy_1t": {"value": b0, "sample": f"{m0} trajectories spread over the batch x {len(act)} agents, "
                                                  "oracle/fo_numpy_ref.py (the reference's loop structure: per trajectory "
                                                  "-> per metric -> per agent -> per timestep), one thread; extrapolates "
                                                  "linearly in the pair count"},
           "b1_c_1t": {"value": b1, "sample": f"first {m1} trajectories x {A_act} agents, oracle/fo_oracle.c, one thread"},
           "b2_c_allcores": {"value": b2, "cores": threads}, "cpu_model": cpu_model(),
           "reference_python_indicative": "~1e3 pair-evals/s per core (SURVEY section 6; the reference itself cannot run "
                                          "here: shapely / commonroad / frenetix are not installable)"}
    parity = None
    if par is not None:
        # lists: float64 storage 1e-9; float32 storage of float64 results (f32x) 1e-9 on the value BEFORE the rounding (the
        # deviation beyond half a float32 ulp of the oracle's value) and 1e-6 on the stored value; float32 arithmetic 1e-6
        ltol = 1e-9 if lists_fmt == "f64" else 1e-6
        ok = bool(par["float_max_abs_err"] <= 1e-9 and par["list_max_abs_err"] <= ltol and
                  par["int_mismatches"] == 0 and par["pattern_mismatches"] == 0)
        parity = dict(par, checked_against="oracle/fo_oracle.c on every pair of the batch (full outputs)", tol_float=1e-9,
                      tol_lists=ltol, lists=lists_fmt)
        if lists_fmt == "f32x":
            parity["tol_lists_before_rounding"] = 1e-9
            ok = ok and par["list_err_beyond_f32_rounding"] <= 1e-9
        parity["ok"] = ok
    return cpu, parity


HBM_ACHIEVABLE_GBS = 6300.0   # what a plain fill reaches on this chip (guide; tools/microbench/write_bw.py)


def bound_word(hbm_frac, valu_frac):
    """the binding unit by the two counter fractions: a unit is named only when its fraction exceeds 0.8; below that the
    kernel is short of both limits at once and the line says so instead of picking the larger of two middling numbers"""
    if max(hbm_frac, valu_frac) > 0.8:
        return "hbm" if hbm_frac > valu_frac else "valu-issue"
    return f"latency/issue mix (hbm {hbm_frac:.2f}, valu {valu_frac:.2f})"


def roofline_checks(a8d, ast, kern_s, traffic):
    """what has to hold between the byte figures of one launch before they are printed: SURVEY 8d's bytes <= the bytes this
    build stores <= 1.1 x the counter traffic of the committed profile (when there is one), and no figure above what the
    chip reaches.  Returns (checks, ok)."""
    c = {"algorithmic_le_stored": a8d <= ast,
         "stored_le_1p1_traffic": (ast <= 1.1 * traffic) if traffic else None,
         "achieved_stored_le_hbm_achievable": ast / kern_s / 1e9 <= HBM_ACHIEVABLE_GBS,
         "achieved_le_hbm_achievable": a8d / kern_s / 1e9 <= HBM_ACHIEVABLE_GBS}
    return c, all(v is not False for v in c.values())


def committed_pmc(N, mode, lists_fmt):
    """What the committed rocprofv3 summary of THIS library says about the dominant kernel of an output mode:
    HBM traffic per launch (WRITE_SIZE + 2 x FETCH_SIZE, the guide's gfx950 correction; KB -> bytes), the VALU issue fraction
    SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs), the
    HBM fraction = traffic / time over the 6.3 TB/s the chip reaches, and the binding unit = the larger of the two fractions
    if it exceeds 0.8, else "latency/issue mix" with both figures (bound_word).  profiles/<tag>_build.json must name the fo_build_id() of the loaded library, else
    everything is None."""
    import csv
    base = os.environ.get("FO_PROFILE_TAG", "r06")
    tag = base + {"full/f32x": "_final", "full/f64": "_f64lists", "full/f32": "_f32", "reduced": "_reduced",
                  "pair": "_pair"}.get(mode + ("/" + lists_fmt if mode == "full" else ""), "_final")
    want = {"full/f32": "fo_sweep_queue_kernel<true, 2", "full/f64": "fo_sweep_queue_kernel<true, 1",
            "full/f32x": "fo_sweep_queue_kernel<true, 3", "reduced": "fo_sweep_queue_kernel<false, 0",
            "pair": "fo_sweep_queue_kernel<true, 0"}[mode + ("/" + lists_fmt if mode == "full" else "")]
    out = {"traffic": None, "valu_issue_frac": None, "bound": None, "profile": tag, "hbm_traffic_gbs": None, "hbm_frac": None}
    try:
        with open(os.path.join(ROOT, "profiles", f"{tag}_build.json")) as f:
            meta = json.load(f)
        if meta.get("build_id") != N.build_id():
            return out
        with open(os.path.join(ROOT, "profiles", f"{tag}_summary.csv")) as f:
            for row in csv.DictReader(f):
                if not row["kernel"].startswith(want) or ", false>" not in row["kernel"]:
                    continue
                t_s = float(row["avg_ns"]) * 1e-9
                if row.get("WRITE_SIZE") and row.get("FETCH_SIZE"):
                    out["traffic"] = (float(row["WRITE_SIZE"]) + 2.0 * float(row["FETCH_SIZE"])) * 1024.0
                    out["hbm_traffic_gbs"] = out["traffic"] / t_s / 1e9
                if row.get("SQ_INSTS_VALU") and row.get("GRBM_GUI_ACTIVE"):
                    out["valu_issue_frac"] = float(row["SQ_INSTS_VALU"]) * 4.0 / (1024.0 * float(row["GRBM_GUI_ACTIVE"]) / 8.0)
                if out["hbm_traffic_gbs"] is not None:
                    out["hbm_frac"] = out["hbm_traffic_gbs"] / HBM_ACHIEVABLE_GBS
                if out["hbm_frac"] is not None and out["valu_issue_frac"] is not None:
                    out["bound"] = bound_word(out["hbm_frac"], out["valu_issue_frac"])
                break
    except Exception:
        pass
    return out


def live_pmc(lists_fmt, mode="full"):
    """HBM traffic and VALU issue counters of the dominant kernel measured IN THIS RUN: two short `rocprofv3 --pmc` passes of
    this script as child processes (the guide's recipe: counters in passes of their own, --pmc with --kernel-trace only; FETCH_SIZE
    x 2 on gfx950, KB -> bytes), started before this process has touched the GPU.  Returns the same keys as committed_pmc(), or
    None when rocprofv3 is not there or a pass fails (the line then falls back to the committed summary of this library)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    want = {"full/f32": "fo_sweep_queue_kernel<true, 2", "full/f64": "fo_sweep_queue_kernel<true, 1",
            "full/f32x": "fo_sweep_queue_kernel<true, 3", "reduced": "fo_sweep_queue_kernel<false, 0",
            "pair": "fo_sweep_queue_kernel<true, 0"}[mode + ("/" + lists_fmt if mode == "full" else "")]
    vals, t0 = {}, time.time()
    top = tempfile.mkdtemp(prefix="fo_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for i, grp in enumerate((("FETCH_SIZE", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"), ("WRITE_SIZE",))):
            d = os.path.join(top, f"p{i}")
            cmd = [exe, "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-autotune",
                   "--no-extras", "--no-live-pmc", "--mode", mode, "--lists", lists_fmt]
            # (a session of its own: should a pass hang, the profiler AND the bench child under it are ended -- exactly the
            # process group started here, nothing matched by name)
            pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True,
                                  env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")), cwd=top)
            try:
                rc = pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None
            if rc != 0:
                return None
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                    if k.startswith(want) and ", false>" in k:
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in grp:
                if not acc.get(c):
                    return None
                vals[c] = sum(acc[c]) / len(acc[c])
                vals["launches"] = len(acc[c])
    except Exception:
        return None
    finally:
        shutil.rmtree(top, ignore_errors=True)
    return {"traffic": (vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"]) * 1024.0,
            "valu_issue_frac": vals["SQ_INSTS_VALU"] * 4.0 / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0),
            "launches": vals["launches"], "seconds": time.time() - t0}


def small_batch_step(local_rank, steps=300):
    """BASELINE configs[1] beside the headline: scenario1 geometry, 2 000 candidates x 32 phantom slots, the same planning
    step (scene stage + sampling + sweep + reduction, reduced outputs as a planner consumes them), own context; a few
    hundred steps of ~0.1 ms.  Reported under config.small_batch -- ms per planning step at the reference's own size."""
    import numpy as np
    import torch
    import yaml
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as SC
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.sweep import MetricSweep
    M, A, T = 2000, 32, 31
    ctx = N.Context(local_rank)
    sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
    ego = sc.ego_initial
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
        cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
    cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
    yaw = float(ego[2])
    ref = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    sm = SensorModel(sc.lanelets, ref, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=local_rank)
    sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
    sl = SpawnLocator(None, ref, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=THR, device=local_rank, ctx=ctx)
    sw.reserve(M, T, A, T)
    traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 2, ego_pos=ego[:2], ego_yaw=yaw)
    tr = [torch.as_tensor(traj[k]).to(f"cuda:{local_rank}") for k in ("x", "y", "theta", "v", "a")]
    from frenetix_occlusion.step import PlanningStep
    ps = PlanningStep(sm, sl, sw, *tr, mode="reduced")      # the whole step through ONE native call (fo_step_run)
    out = None

    def step_stages():      # the same step as five stage calls from Python (what round 2 measured)
        nonlocal out
        sm.launch(ego[:2], yaw)
        sw.set_agents(*sl.sample(ego[:2], yaw, float(ego[3])).sweep_args(), check=False)
        out = sw.run(*tr, mode="reduced", out=out)

    def step():
        ps.run(ego[:2], yaw, float(ego[3]))

    for _ in range(100):
        step_stages()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_stages()
    torch.cuda.synchronize()
    dt_stages = (time.perf_counter() - t0) / steps
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    # three timed runs, the median reported: at 0.09 ms per step the host thread that issues the step matters, and its
    # cores are shared with whatever else runs on the box (all three runs are in `ms_per_step_runs`)
    runs = []
    # (HIP events around every 16th sweep launch: an event record is a barrier packet of its own -- around every launch
    # the two records cost 6 us of the 80 us step, measured with FO_BENCH_TIME_EVERY_SMALL=1 / 1000)
    sw.ctx.timing(True, every=int(os.environ.get("FO_BENCH_TIME_EVERY_SMALL", "16")))
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t_issue_ = (time.perf_counter() - t0) / steps
        torch.cuda.synchronize()
        runs.append(((time.perf_counter() - t0) / steps, t_issue_))
    kms, kn = sw.ctx.timing_read()
    sw.ctx.timing(False)
    dt, t_issue = sorted(runs)[1]

    # the same step with FULL outputs in the headline list format (float64 results stored as float32): the queue kernel's
    # horizon-split instantiation of that format (round 6) beside the generic kernel the format used to drop to at this size
    def full_f32x(generic):
        if generic:
            os.environ["FO_SWEEP_GENERIC"] = "1"
        try:
            psf = PlanningStep(sm, sl, sw, *tr, mode="full", lists="f32x")
            for _ in range(50):
                psf.run(ego[:2], yaw, float(ego[3]))
            torch.cuda.synchronize()
            sw.ctx.timing(True, every=16)
            t0_ = time.perf_counter()
            for _ in range(steps):
                psf.run(ego[:2], yaw, float(ego[3]))
            torch.cuda.synchronize()
            d_ = (time.perf_counter() - t0_) / steps
            k_, n_ = sw.ctx.timing_read()
            sw.ctx.timing(False)
            return {"ms_per_step": d_ * 1e3, "sweep_kernel_ms": k_ / max(n_, 1), "sweep_grid": sw.ctx.last_launch()["grid"],
                    "sweep_block": sw.ctx.last_launch()["block"]}
        finally:
            os.environ.pop("FO_SWEEP_GENERIC", None)

    full_q, full_g = full_f32x(False), full_f32x(True)
    return {"workload": "BASELINE configs[1]: scenario1 geometry, 2000 trajectories x 32 phantom slots, T=31, reduced outputs",
            "full_f32x": dict(full_q, what="the same step with full outputs, lists = float64 results stored as float32 (the headline "
                                           "format): fo_sweep_queue_kernel<true, 3, true, true>, the horizon-split form",
                              generic_kernel=full_g),
            "ms_per_step": dt * 1e3, "host_issue_ms_per_step": t_issue * 1e3, "ms_per_step_stage_calls": dt_stages * 1e3,
            "entry": "fo_step_run (one native call per planning step); ms_per_step_stage_calls = the same step as five "
                     "stage calls from Python",
            "ms_per_step_runs": [r[0] * 1e3 for r in runs],
            "sweep_kernel_ms": kms / max(kn, 1), "A_active": int(sl.batch.n.item()), "steps": steps,
            "sweep_grid": sw.ctx.last_launch()["grid"]}


def future_visibility_leg(scene, tx, ty, n=5):
    """SURVEY 8f-2 (an extension: the reference never fills its `occ_*` cost terms): for the headline batch on the urban grid,
    every 5th sample of every candidate (10 000 x 7 poses) casts the visibility stage's own 720-ray fan against the static
    map + obstacles and counts the cells of the CURRENT occluded set it would reveal; HIP events around n calls."""
    import torch
    sm = scene["sm"]
    out = {}
    for rays in (720, 192):
        sm.future_visibility(tx, ty, t_stride=5, n_rays=rays)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            rev, area = sm.future_visibility(tx, ty, t_stride=5, n_rays=rays)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        M, K = int(tx.shape[0]), int(rev.shape[1])
        out[f"rays{rays}"] = {"ms": ms, "poses": M * K, "poses_per_sec": M * K / ms * 1e3, "rays_per_sec": M * K * rays / ms * 1e3,
                              "mean_revealed_last_pose": float(rev[:, -1].double().mean())}
    out["what"] = ("fo_scene_future_visibility (extension, SURVEY 8f-2): 10 000 trajectories x 7 poses (t_stride 5), a world-aligned full "
                   "fan per pose against 9 360 boundary pieces + 64 obstacles, revealed cells of the current occluded set "
                   f"({int(sm.n_occluded.item())} cells) by the chord rule; 720 rays = the visibility stage's fan, 192 = the round-1 figure")
    return out


def rules_step(local_rank, steps=60):
    """The reference's own spawn semantics as a device-resident planning step (fo_step_run, spawn_mode FO_SPAWN_RULES): scenario1
    geometry, 2 000 candidates, the three rule families of spawn_locator.py:145-578 on the cell classes -> their spawn points
    -> phantom agents with predictions -> sweep, nothing read back.  Two poses of the fixture: time step 0 and the step at
    which a visible dynamic obstacle qualifies for the Car / Bicycle rule (its candidate region is built: the dear case)."""
    import numpy as np
    import torch
    import yaml
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import interface
    from frenetix_occlusion import scenario as SC
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.sensor_model import SensorModel
    from frenetix_occlusion.spawn_locator import SpawnLocator
    from frenetix_occlusion.step import PlanningStep
    from frenetix_occlusion.sweep import MetricSweep
    from frenetix_occlusion.utils.fo_obstacle import FOObstacles
    M, T = 2000, 31
    ctx = N.Context(local_rank)
    sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
    ego0 = sc.ego_initial
    with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["accelerator"]["spawn"].update(mode="rules", routes=3, max_rule_points=8)
    yaw = float(ego0[2])
    path = ego0[None, :2] + np.linspace(-5.0, 80.0, 171)[:, None] * np.array([[math.cos(yaw), math.sin(yaw)]])
    obs = FOObstacles(sc.obstacles)
    sm = SensorModel(sc.lanelets, path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5, ctx=ctx, device=local_rank,
                     routes=3, intersections=sc.intersections)
    sl = SpawnLocator(None, path, cfg, sm, fo_obstacles=obs, dt=0.1, horizon=(T - 1) * 0.1)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=THR, device=local_rank, ctx=ctx)
    traj = S.make_trajectories(M, T, 0.1, seed=20240131 + 2, ego_pos=ego0[:2], ego_yaw=yaw)
    tr = [torch.as_tensor(traj[k]).to(f"cuda:{local_rank}") for k in ("x", "y", "theta", "v", "a")]
    ps = PlanningStep(sm, sl, sw, *tr, mode="reduced")
    out = {"workload": "scenario1 geometry, 2000 trajectories, spawn.mode rules (the reference's three rule families on the device, "
                       "<= 8 spawn points x 3 route slots), T=31, reduced outputs, fo_step_run; the ego drives the scenario's first 61 "
                       "time steps (0.76 m per step), every pose timed on its own", "steps_per_pose": steps, "poses": {}}
    per, n_dyn, n_pts, worst = [], 0, 0, 0.0
    for step in range(min(61, int(os.environ.get("FO_RULES_LAST_POSE", "60")) + 1)):   # (the variable: tools/rule_wtrace.py)
        ego = ego0[:2] + 0.7634 * step * np.array([math.cos(yaw), math.sin(yaw)])
        obs.update(step)
        sm.upload_obstacles(obs)
        # (the planner hands the ego's curvilinear position to evaluate_scenario, interface.py:148: an input of the step)
        if not getattr(sl, "_rules_ready", False):
            sl._rule_setup()
        ego_cl = sl._cs.convert_to_curvilinear_coords(float(ego[0]), float(ego[1]))
        for _ in range(10):
            ps.run(ego, yaw, float(ego0[3]), ego_cl)
        torch.cuda.synchronize()
        # three batches per pose, the median one counts (a pose is ~5 ms of GPU time: one 30 ms stall of the box -- seen once in
        # twenty runs, on a pose without spawn points -- would otherwise carry the whole drive's mean); the worst is kept beside it
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps // 3):
                ps.run(ego, yaw, float(ego0[3]), ego_cl)
            t_issue = (time.perf_counter() - t0) / (steps // 3)
            torch.cuda.synchronize()
            runs.append(((time.perf_counter() - t0) / (steps // 3), t_issue))
        worst = max(worst, max(r[0] for r in runs) * 1e3)
        dt, t_issue = sorted(runs)[1]
        h = sl.batch.host_head()
        kinds = [{0: "Car", 3: "Bicycle", 4: "Pedestrian"}[int(q[0])] for q in h["rule_points"][:h["rule_n"]]]
        per.append(dt * 1e3)
        n_pts += len(kinds)
        n_dyn += any(k != "Pedestrian" for k in kinds)
        if step in (0, 8, 25, 60) or dt * 1e3 == max(per):
            out["poses"][f"step{step}"] = {"ms_per_step": dt * 1e3, "host_issue_ms_per_step": t_issue * 1e3, "spawn_points": kinds}
    out["ms_per_step"] = float(np.mean(per))
    out["ms_per_step_max"] = float(np.max(per))
    out["ms_per_step_p50"] = float(np.median(per))
    out["ms_per_step_worst_batch"] = worst
    out["ms_per_step_definition"] = ("mean over the 61 poses of the drive (max / median pose beside it); a pose = the median of three "
                                     "batches of steps_per_pose / 3 steps, the slowest single batch of the drive under worst_batch")
    out["spawn_points_total"], out["poses_with_a_phantom_vehicle"] = n_pts, int(n_dyn)
    return out


def box_probe(local_rank):
    """What THIS box reaches on two plain library operations, beside the step it just timed: the boxes of the pool differ (the same
    library measured 0.487-0.553 ms on a dozen of them in round 5), and a bench line that carries a fill rate and a float64 GEMM
    rate of its own box says which kind it met.  Context only: `roofline.peak` stays the guide's 8 TB/s."""
    import torch
    dev = torch.device("cuda", local_rank)
    out = {}
    buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        buf.fill_(1)
    e0.record()
    for _ in range(8):
        buf.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    out["fill_gbs"] = 8 * buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del buf
    n = 4096
    a = torch.randn((n, n), dtype=torch.float64, device=dev)
    b = torch.randn((n, n), dtype=torch.float64, device=dev)
    c = a @ b
    e0.record()
    for _ in range(4):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    out["dgemm_f64_tflops"] = 4 * 2.0 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    out["what"] = "torch fill_ of 1 GiB and a 4096^3 float64 matmul on this box, after the timed steps (context for the box-to-box spread; not used in any figure above)"
    return out


def shard_probe(scene, sw, tensors, local_rank, N, steps=200):
    """What ONE GPU can measure about BASELINE configs[3] (the 10k x 256 batch block-partitioned over 8 / 4 / 2 GPUs): the planning
    step of a 1/8, 1/4 and 1/2 shard -- same scene stage, same agents, the library's measured agents-per-wave for the shard's
    shape -- and the all-gather of a [M/8][16] cost block through RCCL on one rank."""
    import numpy as np
    import torch
    from frenetix_occlusion.step import PlanningStep
    ego = scene["ego"]
    M_total = int(tensors[0].shape[0])
    res = {"definition": "ms per planning step of a shard of the headline batch on ONE GPU (scene stage + sampling + sweep + reduction; "
                         "reduced outputs = what a rank all-gathers, and full outputs with float32 lists)", "shards": {}}
    for div in (8, 4, 2):
        m = -(-M_total // div)
        row = {"M": m}
        for mode in ("reduced", "full"):
            tr = [t[:m] for t in tensors]
            ps = PlanningStep(scene["sm"], scene["sl"], sw, *tr, mode=mode, lists="f32")
            ps.run(ego[:2], float(ego[2]), float(ego[3]))
            o = sw.run(*tr, mode=mode, lists="f32", autotune=40)      # per-shape choice, kept in the context
            del o
            for _ in range(40):
                ps.run(ego[:2], float(ego[2]), float(ego[3]))
            torch.cuda.synchronize()
            sw.ctx.timing(True, every=4)
            t0 = time.perf_counter()
            for _ in range(steps):
                ps.run(ego[:2], float(ego[2]), float(ego[3]))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            kms, kn = sw.ctx.timing_read()
            sw.ctx.timing(False)
            ll = sw.ctx.last_launch()
            row[mode] = {"ms_per_step": dt * 1e3, "sweep_kernel_ms": kms / max(kn, 1), "sweep_grid": ll["grid"],
                         "agents_per_wave": ll["agents_per_wave"]}
            del ps
        res["shards"][f"1/{div}"] = row
    # the collective alone, one rank: [M/8][16] float64 through RCCL.  (RCCL prints a version banner into the C library's
    # stdout buffer, which would surface after the JSON line: stdout points at stderr while RCCL is alive.)
    import ctypes
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        import torch.distributed as dist
        own = not dist.is_initialized()
        if own:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        from frenetix_occlusion.distributed import CostGather
        per = -(-M_total // 8)
        cg1 = CostGather(per, device=torch.device("cuda", local_rank), force=True)    # one rank's block, through RCCL
        cg1.block().zero_()
        for _ in range(20):
            cg1.gather()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            cg1.gather()
        e1.record()
        torch.cuda.synchronize()
        res["allgather_ms"] = e0.elapsed_time(e1) / 100
        res["allgather"] = f"all_gather_into_tensor of one [{per}][16] float64 block, RCCL, world size 1 (launch + protocol floor; no xGMI hop)"
        if own:
            dist.destroy_process_group()
    except Exception as e:        # RCCL unavailable on the box: say so, keep the line
        res["allgather_ms"], res["allgather"] = None, f"not measured: {type(e).__name__}: {e}"
    finally:
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="full", choices=["full", "pair", "reduced"])
    ap.add_argument("--lists", default="f32x", choices=["f32x", "f32", "f64"],
                    help="per-timestep lists of --mode full: f32x (default) = float64 arithmetic like the reference, stored as float32 "
                         "(SURVEY 8d's storage); f32 = the harm entries away from the gate in float32 arithmetic as well; f64 = the "
                         "reference's own list dtype")
    ap.add_argument("--M", type=int, default=10000)
    ap.add_argument("--A", type=int, default=256)
    ap.add_argument("--T", type=int, default=31)
    ap.add_argument("--scene", default="urban", choices=["urban", "scenario1", "synthetic"],
                    help="urban: full planning step on the synthetic urban grid; synthetic: sweep only, fixed agents")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default): --M trajectories in total, split over the ranks (BASELINE configs[3]); "
                         "weak: --M trajectories per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baselines and the oracle parity check")
    ap.add_argument("--no-autotune", action="store_true", help="skip the agents-per-wave selection pass of the set-up")
    ap.add_argument("--entry", default="step", choices=["step", "stages"],
                    help="scene-based runs: the planning step through fo_step_run (one native call, nine launches) or as the five stage calls")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (other output modes, configs[1])")
    ap.add_argument("--order", default="sampler", choices=["sampler", "random"],
                    help="row order of the synthetic trajectories (synthetic.make_trajectories)")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic in this run (the default N = 1 run of the "
                         "headline workload does; the committed summary of the loaded library is the fallback)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher above us: become one -- before torch is imported or a GPU is touched
        sys.exit(launch_workers(args.gpus, sys.argv[1:]))
    if args.launcher_selftest:
        sys.exit(launcher_selftest(args))
    # roofline.traffic measured in THIS run: counter passes as child processes, before this process touches the GPU
    live = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_live_pmc and not args.no_extras and args.scene == "urban"
            and (args.M, args.A, args.T, args.mode) == (10000, 256, 31, "full")):
        live = live_pmc(args.lists, args.mode)

    import numpy as np
    import torch
    from frenetix_occlusion import _native as N
    from frenetix_occlusion import synthetic as S
    from frenetix_occlusion.distributed import CostGather
    from frenetix_occlusion.sweep import MetricSweep

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    scaling = args.scaling or "strong"
    dist = None
    use_dist = world > 1 or os.environ.get("FO_BENCH_FORCE_DIST") == "1"   # env: exercise the RCCL path on one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    M_total, A, T = args.M, args.A, args.T
    # The trajectory split is the product's (frenetix_occlusion.distributed.CostGather: block bounds, the two pre-allocated
    # blocks of the collective, the one all_gather_into_tensor): the step below hands it to PlanningStep / writes its cost
    # rows into its block.  strong (BASELINE configs[3]): ONE batch of --M trajectories, block-partitioned over the ranks;
    # weak: every rank its own --M trajectories = block `rank` of a batch of world x M.
    cg = None
    if use_dist:
        cg = CostGather(M_total if scaling == "strong" else M_total * world, device=torch.device("cuda", local_rank),
                        force=(world == 1))
        if cg.world != world:
            raise RuntimeError(f"the process group has {cg.world} ranks, WORLD_SIZE says {world}")
    if scaling == "strong":
        lo, hi = (cg.lo, cg.hi) if cg is not None else (0, M_total)
        M, per = hi - lo, (cg.per if cg is not None else M_total)
    else:
        lo, M, per = 0, M_total, M_total
    ctx = N.Context(local_rank)
    sw = MetricSweep(S.VEHICLE_BMW320I, 0.1, thresholds=THR, device=local_rank, ctx=ctx)
    sw.reserve(max(M, 1), T, A, T)
    d = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a)).to(dev, dt)

    scene = None
    seed = 20240131 + 3 + (1000 * rank if scaling == "weak" else 0)
    if args.scene in ("urban", "scenario1"):
        import yaml
        from frenetix_occlusion import interface
        from frenetix_occlusion import scenario as SC
        from frenetix_occlusion.sensor_model import SensorModel
        from frenetix_occlusion.spawn_locator import SpawnLocator
        if args.scene == "urban":
            sc = SC.synthetic_urban_grid()
        else:   # BASELINE configs[1]: scenario1.xml geometry (committed fixture), use with --M 2000 --A 32
            sc = SC.load_geometry_npz(os.path.join(ROOT, "tests", "golden", "scenario1_geometry.npz"))
        ego = sc.ego_initial
        with open(os.path.join(os.path.dirname(interface.__file__), "config", "config.yaml")) as f:
            cfg = yaml.safe_load(f)
            cfg["accelerator"]["spawn"]["mode"] = "cells"   # the BASELINE-config sampler (the YAML default is the reference's rule families)
        cfg["accelerator"]["spawn"].update(max_agents=A, all_occluded=True, max_dist=45.0)
        ref_path = ego[None, :2] + np.linspace(0.0, 80.0, 81)[:, None] * np.array([[math.cos(ego[2]), math.sin(ego[2])]])
        sm = SensorModel(sc.lanelets, ref_path, sensor_radius=50.0, sensor_angle=360.0, n_rays=720, cell_size=0.5,
                         ctx=ctx, device=local_rank)
        sm.upload_obstacles(sc.obstacle_arrays(0)[:3])
        sl = SpawnLocator(None, ref_path, cfg, sm, dt=0.1, horizon=(T - 1) * 0.1)
        scene = dict(sm=sm, sl=sl, ego=ego, edges=len(sm.map_geometry.edges), obstacles=len(sc.obstacles))
        traj_all = S.make_trajectories(M_total, T, 0.1, seed=seed, ego_pos=ego[:2], ego_yaw=float(ego[2]), order=args.order)
        agents = None
    else:
        traj_all = S.make_trajectories(M_total, T, 0.1, seed=seed, order=args.order)
        agents = S.make_agents(A, T, 0.1, seed=20240131 + 3)
        ag = [d(agents[k]) for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims")] + \
             [d(agents["type"], torch.int32), d(agents["len"], torch.int32)]
    traj = {k: v[lo:lo + M] for k, v in traj_all.items()} if scaling == "strong" else traj_all
    tx, ty, tth, tv, ta = (d(traj[k]) for k in ("x", "y", "theta", "v", "a"))
    out = None
    # strong scaling through PlanningStep(shard=cg): the step object takes the WHOLE batch and the split, and slices itself
    full = tuple(d(traj_all[k]) for k in ("x", "y", "theta", "v", "a")) if (cg is not None and scaling == "strong") else None

    def scene_stage():
        sm, sl, ego = scene["sm"], scene["sl"], scene["ego"]
        sm.launch(ego[:2], float(ego[2]))
        return sl.sample(ego[:2], float(ego[2]), float(ego[3])).sweep_args()

    planning_steps = {}   # --entry step: one PlanningStep (fo_step_run: the whole step in one native call) per output mode

    def step(mode=args.mode, lists=args.lists, res=None, gather=True):
        sharded_step = full is not None and gather     # (the set-up passes measure the kernel alone: no collective)
        if scene is not None and args.entry == "step" and M > 0:
            ps = planning_steps.get((mode, lists, sharded_step))
            if ps is None:
                from frenetix_occlusion.step import PlanningStep
                ps = planning_steps[(mode, lists, sharded_step)] = (
                    PlanningStep(scene["sm"], scene["sl"], sw, *full, mode=mode, lists=lists, shard=cg) if sharded_step else
                    PlanningStep(scene["sm"], scene["sl"], sw, tx, ty, tth, tv, ta, mode=mode, lists=lists))
            ego = scene["ego"]
            res = ps.run(ego[:2], float(ego[2]), float(ego[3]))       # (sharded: ends with cg.gather())
            if sharded_step:
                return res
        else:
            a_args = scene_stage() if scene is not None else ag
            sw.set_agents(*a_args, check=False)
            if M > 0:
                if res is None and cg is not None and scaling == "strong":
                    res = sw.alloc_out(M, T, sw.A, mode, lists, cost=cg.block())    # cost rows straight into the collective's block
                res = sw.run(tx, ty, tth, tv, ta, mode=mode, out=res, lists=lists)
        if cg is not None and gather:
            cg.gather(res.cost if M > 0 else None)      # (no copy when the rows were written into the block)
        return res

    # Everything slow on the host side happens first (the first timing() call creates the event pool; the check and the
    # agent count synchronise), so that from here to the timed region the GPU is never idle for longer than a
    # synchronize takes (microseconds): after an idle of milliseconds the MI355X runs two sweeps fast and the next ~15
    # 6-8 % slow, decaying over ~40 steps (power management; per-launch series in DESIGN §7) -- a 20-step window right
    # behind such an idle would consist of that dip.
    sw.ctx.timing(True)
    out = step(res=out)
    sw.ctx.call("fo_sweep_check", torch.cuda.current_stream().cuda_stream)
    n_active = A
    if scene is not None:
        n_active = int(scene["sl"].batch.n.item())
    sw.ctx.timing(False)
    # Set-up, before the W warm-up steps: the LIBRARY measures the sweep kernel's agents-per-wave settings on this batch
    # (fo_sweep_autotune: 4 settings x 100 launches, ~0.3 s) and keeps the best for the shape -- what any caller of the C ABI
    # gets, no environment variable.  The same pass brings the GPU to its sustained clocks: a cold MI355X runs the first
    # few dozen steps 10-15 % slower, so without it the result would depend on W.
    tune = {}
    if not os.environ.get("FO_SWEEP_APW") and not args.no_autotune and M > 0:
        o_t = sw.run(tx, ty, tth, tv, ta, mode=args.mode, lists=args.lists, autotune=100)
        tune = dict(sw.last_autotune["ms"])
        del o_t
        torch.cuda.empty_cache()
        for _ in range(100):
            out = step(res=out, gather=False)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = step(res=out)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events around the sweep kernel of every 4th timed step (FO_BENCH_TIME_EVERY=k): the two event records of a launch
    # are barrier packets of their own and cost the step ~6 us (1 % here, 7 % at the reference's own size, small_batch_step)
    time_every = int(os.environ.get("FO_BENCH_TIME_EVERY", "4"))
    if args.steps < 4 * time_every:
        time_every = 1
    sw.ctx.timing(True, every=time_every)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(res=out)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_each = sw.ctx.timing_read_each()
    kern_ms, kern_n = float(sum(kern_each)), len(kern_each)
    if os.environ.get("FO_BENCH_DUMP_SERIES"):
        with open(os.environ["FO_BENCH_DUMP_SERIES"], "w") as f:
            f.write(" ".join(f"{v:.4f}" for v in kern_each) + "\n")
    sw.ctx.timing(False)
    ag_ms = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the collective alone: 100 all-gathers of the cost blocks back to back
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            cg.gather()
        e1.record()
        torch.cuda.synchronize()
        ag_ms = e0.elapsed_time(e1) / 100

    # stage breakdown (outside the timed region): the scene stage alone, HIP events on the launch stream
    scene_ms = None
    if scene is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            scene_stage()
        e1.record()
        torch.cuda.synchronize()
        scene_ms = e0.elapsed_time(e1) / 20

    if rank == 0:
        M_job = M_total if scaling == "strong" else world * M
        pairs = M_job * n_active * args.steps
        kern_s = kern_ms / 1e3 / max(kern_n, 1)
        a8d, ast = bytes_8d(M, n_active, T, args.mode), bytes_stored(M, n_active, T, args.mode, args.lists)
        achieved = a8d / kern_s / 1e9
        launch = sw.ctx.last_launch()
        default_workload = (args.scene == "urban" and M_total == 10000 and A == 256 and T == 31 and args.mode == "full")
        pmc = {"traffic": None, "valu_issue_frac": None, "bound": None, "profile": None, "hbm_traffic_gbs": None, "hbm_frac": None}
        if default_workload and world == 1:
            pmc = committed_pmc(N, args.mode, args.lists)
        traffic_source = ("committed rocprofv3 summary of this library: profiles/" + pmc["profile"] + "_summary.csv") if pmc["traffic"] else None
        if live is not None:      # measured in this run: per launch, over this run's own kernel time
            pmc = dict(pmc, traffic=live["traffic"], valu_issue_frac=live["valu_issue_frac"],
                       hbm_traffic_gbs=live["traffic"] / kern_s / 1e9, hbm_frac=live["traffic"] / kern_s / 1e9 / HBM_ACHIEVABLE_GBS,
                       committed_traffic=pmc["traffic"])
            pmc["bound"] = bound_word(pmc["hbm_frac"], pmc["valu_issue_frac"])
            traffic_source = (f"this run: two rocprofv3 --pmc passes of bench.py as child processes ({live['launches']} launches of the "
                              f"kernel each, {live['seconds']:.0f} s), WRITE_SIZE + 2 x FETCH_SIZE")
        checks, checks_ok = roofline_checks(a8d, ast, kern_s, pmc["traffic"])
        if not checks_ok:     # a bug of this script, not of the box: say so loudly and print no impossible bandwidth
            print(f"bench.py: inconsistent byte figures {checks}", file=sys.stderr, flush=True)
            if os.environ.get("FO_BENCH_STRICT") == "1":
                raise AssertionError(f"roofline byte figures are inconsistent: {checks}")
        dtype = "f64" if (args.mode != "full" or args.lists == "f64") else ("f64+f32lists" if args.lists == "f32" else "f64 (lists stored f32)")
        res = {
            "metric": "trajectory_x_agent_metric_evals_per_sec", "value": pairs / elapsed, "unit": "pair-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: synthetic urban lanelet net, 10k trajectories x 256 phantoms, "
                                    "360 deg ray-cast @ 0.5 deg, T=31 (full planning step)" if world == 1 else
                                    "BASELINE configs[3]: the synthetic 10k x 256 batch of configs[2] block-partitioned "
                                    f"over {world} GPUs, one RCCL all-gather of the cost vectors per step"
                                    if scaling == "strong" else
                                    f"weak scaling: every one of {world} ranks evaluates its own {M} trajectories x 256 phantoms "
                                    "(NOT configs[3]), cost vectors all-gathered") if args.scene == "urban" else
                                   (f"BASELINE configs[1]: scenario1 geometry, {M_total} trajectories x {A} phantom slots, full "
                                    "metric set, T=31 (full planning step)") if args.scene == "scenario1" else
                                   "sweep only: 10k synthetic trajectories x 256 synthetic phantom predictions, T=31",
                       "M_total": M_job, "M_per_gpu": M, "A": A, "A_active": n_active, "T": T, "output_mode": args.mode,
                       "entry": ("fo_step_run (PlanningStep: one native call per planning step)" if scene is not None and args.entry == "step"
                                 else "stage calls"),
                       "list_storage": (args.lists + (" (harm entries away from the 5 m gate evaluated in float32; every cost, "
                                                     "flag and pair scalar float64)" if args.lists == "f32" else
                                                     " (float64 results rounded to float32 at the store)" if args.lists == "f32x" else ""))
                       if args.mode == "full" else None,
                       "traj_order": args.order, "metrics": ["hr", "ttc", "ttce", "dce", "wttc", "cp"],
                       "scene_stage_ms": scene_ms, "sweep_kernel_ms": kern_s * 1e3,
                       "agents_per_wave": launch["agents_per_wave"],
                       "agents_per_wave_source": ("FO_SWEEP_APW (environment)" if os.environ.get("FO_SWEEP_APW") else
                                                  "fo_sweep_autotune (library: measured per batch shape, kept in the context)" if tune
                                                  else "static rule of fo_sweep_run"),
                       "setup_autotune_ms_per_sweep": {str(k): round(v, 4) for k, v in tune.items()},
                       "boundary_edges": scene["edges"] if scene else None, "rays": 720 if scene else None,
                       "parallelism": f"traj-shard x{world}", "ranks_seen": cg.world if cg is not None else 1,
                       "split": (("frenetix_occlusion.distributed.CostGather" + (" inside PlanningStep(shard=...)" if (full is not None and scene is not None and args.entry == "step") else ""))
                                 if cg is not None else None),
                       "allgathers_issued": cg.calls if cg is not None else 0,
                       "allgather_ms": ag_ms, "allgather_bytes_per_rank": per * N.NC * 8 if use_dist else None,
                       "build_id": N.build_id()},
            # bound: from the two fractions the committed PMC summary of this library gives -- VALU issue slots taken, HBM traffic
            # over what the chip reaches -- both printed, a unit named only above 0.8 (bound_word); achieved / peak / frac stay SURVEY 8d's byte figure on the HBM peak
            "roofline": {"bound": pmc["bound"] or "unknown (no PMC summary of this library under profiles/)",
                         "kernel": "fo_sweep_queue_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic"], "traffic_profile": pmc["profile"],
                         "traffic_source": traffic_source, "traffic_committed_profile": pmc.get("committed_traffic"),
                         "hbm_traffic_gbs": pmc["hbm_traffic_gbs"], "hbm_achievable_gbs": HBM_ACHIEVABLE_GBS, "hbm_frac_of_achievable": pmc["hbm_frac"],
                         "valu_issue_frac": pmc["valu_issue_frac"],
                         "valu_issue_frac_definition": "SQ_INSTS_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), same source as traffic",
                         "bytes_definition": "SURVEY 8d, fp32 storage: 620 B/trajectory + 636 B/agent + per pair 48 B "
                                             "scalars (+ 600 B lists in full mode); 64 B/trajectory in reduced mode",
                         "algorithmic_bytes_per_launch": a8d, "stored_bytes_per_launch": ast,
                         "achieved_stored": ast / kern_s / 1e9 if checks_ok else None,
                         "frac_stored_bytes": ast / kern_s / 1e9 / HBM_PEAK_GBS if checks_ok else None,
                         "byte_checks": checks,
                         "kernel_ms": kern_s * 1e3, "launches_timed": kern_n, "timed_every": time_every,
                         "kernel_ms_p50": float(np.percentile(kern_each, 50)) if kern_each else None,
                         "kernel_ms_p95": float(np.percentile(kern_each, 95)) if kern_each else None,
                         "grid": launch["grid"], "block": launch["block"],
                         "kernel_pair_evals_per_sec": M * n_active / kern_s},
        }
        if out is not None and out.pair_f is not None:
            res["config"]["gate_pair_frac"] = float((out.pair_f[N.PF["max_collision_probability"]] > 0).double().mean())
            res["config"]["collision_pair_frac"] = float((out.pair_f[N.PF["dce"]] == 0).double().mean())
            res["config"]["safe_traj_frac"] = float(out.safe.double().mean())
        if world == 1 and not args.no_cpu_baseline:
            # oracle on EVERY pair: parity of the last timed step's buffers + the CPU baselines (before the extras below
            # reuse the context)
            if scene is not None:
                b = scene["sl"].batch
                agents = {k: getattr(b, k).cpu().numpy() for k in ("pos", "yaw", "v", "cov", "shape", "raw_dims", "type", "len")}
            res["cpu_baseline"], res["parity"] = cpu_and_parity(S, N, traj, agents, out, args.lists)
        else:
            res["cpu_baseline"], res["parity"] = None, None
        if world == 1 and default_workload and not args.no_extras:
            def side(mode, lists, n=100):
                r = None
                for _ in range(60):
                    r = step(mode, lists, r)
                torch.cuda.synchronize()
                sw.ctx.timing(True, every=4)
                t_r = time.perf_counter()
                for _ in range(n):
                    r = step(mode, lists, r)
                torch.cuda.synchronize()
                dt_r = (time.perf_counter() - t_r) / n
                kms_r, kn_r = sw.ctx.timing_read()
                sw.ctx.timing(False)
                ks = kms_r / max(kn_r, 1) / 1e3
                b8, bs = bytes_8d(M, n_active, T, mode), bytes_stored(M, n_active, T, mode, lists)
                del r
                pm = committed_pmc(N, mode, lists)
                d_ = {"ms_per_step": dt_r * 1e3, "pair_evals_per_sec": M * n_active / dt_r, "sweep_kernel_ms": ks * 1e3,
                      "bound": pm["bound"], "valu_issue_frac": pm["valu_issue_frac"], "hbm_traffic_gbs": pm["hbm_traffic_gbs"],
                      "hbm_frac_of_achievable": pm["hbm_frac"],
                      "profile": pm["profile"], "steps": n}
                if mode != "reduced":     # (64 B per trajectory against a compute-bound kernel: a byte fraction says nothing there)
                    d_.update(frac=b8 / ks / 1e9 / HBM_PEAK_GBS, frac_stored_bytes=bs / ks / 1e9 / HBM_PEAK_GBS)
                return d_
            del out
            torch.cuda.empty_cache()
            res["config"]["box"] = box_probe(local_rank)
            # the same step with the lists in the other element type, and with reduced outputs (cost vectors + flags:
            # what a planner loop consumes), beside the headline
            for other in ("f64", "f32x", "f32"):
                if other != args.lists:
                    res["config"][{"f64": "f64_lists", "f32": "f32_lists", "f32x": "f32_exact_lists"}[other]] = side("full", other)
            res["config"]["f32_lists"]["what"] = ("the harm list entries away from the 5 m gate in float32 ARITHMETIC (FO_LISTS_F32: narrower than the "
                                                  "reference's float64, |error| < 4e-7): what that shortcut would buy against the headline")
            res["config"]["reduced_outputs"] = side("reduced", args.lists)
            res["config"]["future_visibility"] = future_visibility_leg(scene, tx, ty)
            res["config"]["shard_probe"] = shard_probe(scene, sw, (tx, ty, tth, tv, ta), local_rank, N)
            planning_steps.clear()
            torch.cuda.empty_cache()
            res["config"]["small_batch"] = small_batch_step(local_rank)
            res["config"]["small_batch_full_f32x"] = res["config"]["small_batch"].pop("full_f32x")
            res["config"]["rules_step"] = rules_step(local_rank)
        if use_dist:   # RCCL's start-up banner sits in the C library's stdout buffer: let it out first, the JSON line last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
