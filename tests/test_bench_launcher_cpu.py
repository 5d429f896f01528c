"""`python bench.py --gpus N` must be runnable as the driver runs it -- with and without a launcher above it.  On CPU
(gloo) this drives exactly that plumbing: the self-launcher (N worker processes, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT), the rendezvous on 127.0.0.1, the block partition of ONE batch (strong scaling = BASELINE
configs[3]) and the all-gather of the per-trajectory cost rows in rank order.  The GPU work itself is not part of it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


@pytest.mark.parametrize("n,M", [(2, 101), (3, 10)])
def test_bench_launches_its_own_workers(n, M):
    p, lines = _run({}, ["--gpus", str(n), "--launcher-selftest", "--M", str(M), "--steps", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1                                   # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["gather_ok"] and d["scaling"] == "strong"
    assert d["M_total"] == M and d["M_per_rank"] == -(-M // n)


def test_bench_under_an_external_launcher():
    """what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` amounts to"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "0", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest", "--M", "64", "--steps", "2"]
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    cmd[cmd.index("--master-port") + 1] = str(s.getsockname()[1])
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    q = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr[-2000:]
    d = json.loads([ln for ln in q.stdout.splitlines() if ln.startswith("{")][0])
    assert d["ranks_seen"] == 2 and d["gather_ok"]


def test_a_failing_worker_fails_the_launcher():
    p, _ = _run({"FO_BENCH_LAUNCH_TIMEOUT": "120"}, ["--gpus", "2", "--launcher-selftest", "--M", "8", "--steps", "1",
                                                     "--mode", "nonsense"])
    assert p.returncode != 0


def test_byte_figures_of_the_bench_line():
    """the bench line's own arithmetic (no GPU): SURVEY 8d's bytes of the headline batch, what the build stores per list
    format -- `f32x` is float32 ELEMENTS, priced at 4 bytes (round 5 priced it at 8 and printed 6.9 TB/s "stored") --, the
    consistency checks between them, and the word for the binding unit"""
    sys.path.insert(0, ROOT)
    import bench
    M, A, T = 10000, 256, 31
    assert bench.bytes_8d(M, A, T, "full") == 10000 * 620 + 256 * 636 + 2560000 * 648 == 1665242816
    s64, s32, s32x = (bench.bytes_stored(M, A, T, "full", f) for f in ("f64", "f32", "f32x"))
    assert s32 == s32x == 1843338240 and s64 - s32 == M * A * 5 * 30 * 4
    # the figures of round 5's line: algorithmic <= stored <= 1.1 x counter traffic, nothing above what a fill reaches
    checks, ok = bench.roofline_checks(1665242816, s32x, 0.488e-3, 1993311232.0)
    assert ok and all(v is True for v in checks.values())
    # ... and what round 5 printed: float64-list bytes for the f32x run -- 6.9 TB/s, and 1.7 x the counter traffic
    checks, ok = bench.roofline_checks(1665242816, s64, 0.488e-3, 1993311232.0)
    assert not ok and checks["achieved_stored_le_hbm_achievable"] is False and checks["stored_le_1p1_traffic"] is False
    checks, ok = bench.roofline_checks(1665242816, s32x, 0.488e-3, None)          # no counters: that check is skipped, not failed
    assert ok and checks["stored_le_1p1_traffic"] is None
    assert bench.bound_word(0.95, 0.52) == "hbm" and bench.bound_word(0.03, 0.85) == "valu-issue"
    assert bench.bound_word(0.62, 0.58) == "latency/issue mix (hbm 0.62, valu 0.58)"


def test_live_counter_passes_are_parsed_and_failures_fall_back(tmp_path, monkeypatch):
    """bench.live_pmc without a GPU: a stand-in `rocprofv3` on PATH that writes what the real one does for `--pmc ... --output-format
    csv -d DIR` (one row per dispatch and counter).  Parsed: the mean per launch of the headline instantiation only (not the
    correlated-covariance body's name, not other kernels), KB -> bytes, FETCH_SIZE x 2, the issue fraction; a pass that fails,
    or one that lacks a counter, yields None (the bench line then falls back to the committed summary)."""
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text('''#!%s
import os, sys
a = sys.argv[1:]
if os.environ.get("FAKE_FAIL") == "1":
    sys.exit(3)
ctrs = a[a.index("--pmc") + 1:a.index("--kernel-trace")]
d = a[a.index("-d") + 1]
os.makedirs(os.path.join(d, "host", "123"), exist_ok=True)
vals = {"FETCH_SIZE": 70000.0, "WRITE_SIZE": 1800000.0, "GRBM_GUI_ACTIVE": 8.0e6, "SQ_INSTS_VALU": 1.5e8}
names = ["void (anonymous namespace)::fo_sweep_queue_kernel<true, 3, true, false>((anonymous namespace)::SweepArgs)",
         "void (anonymous namespace)::fo_sweep_queue_kernel<true, 3, true, true>((anonymous namespace)::SweepArgs)",
         "(anonymous namespace)::fo_reduce_kernel(int)"]
with open(os.path.join(d, "host", "123", "run_counter_collection.csv"), "w") as f:
    f.write("Kernel_Name,Counter_Name,Counter_Value\\n")
    for rep in range(4):
        for k, n in enumerate(names):
            for c in ctrs:
                if os.environ.get("FAKE_DROP") == c:
                    continue
                f.write('"%%s",%%s,%%r\\n' %% (n, c, vals[c] * (1 if k == 0 else 7)))
''' % sys.executable)
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    got = bench.live_pmc("f32x")
    assert got is not None and got["launches"] == 4
    assert got["traffic"] == (1800000.0 + 2 * 70000.0) * 1024.0
    assert abs(got["valu_issue_frac"] - 1.5e8 * 4 / (1024 * 8.0e6 / 8)) < 1e-12
    assert not [p for p in os.listdir(tmp_path) if p.startswith("fo_pmc_")]        # its scratch directory is gone
    monkeypatch.setenv("FAKE_DROP", "WRITE_SIZE")
    assert bench.live_pmc("f32x") is None
    monkeypatch.delenv("FAKE_DROP")
    monkeypatch.setenv("FAKE_FAIL", "1")
    assert bench.live_pmc("f32x") is None
