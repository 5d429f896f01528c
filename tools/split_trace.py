#!/usr/bin/env python3
"""Phases of the workgroups of one sweep launch in its horizon-split form (tuning build -DFO_TRACE=1, tools/build_variant.sh):
start -> tables in LDS -> agent constants and first rows resident -> pass 1 -> pass 2 -> segments folded -> end, by wave 0.
usage (GPU box): FO_HIP_LIB=.../libfo_hip_trace.so python tools/split_trace.py [bench args; default: the reference-size step]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "split_trace.bin")
env = dict(os.environ, FO_SWEEP_TRACE=out, FO_SWEEP_TRACE_DUMP="1", FO_SWEEP_TRACE_PHASES="1")
args = sys.argv[1:] or ["--scene", "scenario1", "--M", "2000", "--A", "32", "--mode", "reduced"]
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--no-autotune", "--steps", "3",
                "--warmup", "30"] + args, env=env, stdout=subprocess.DEVNULL, check=True)
t = np.fromfile(out, dtype=np.int64).reshape(-1, 4)
n = len(t) - 32768
wg, ph = t[:n], t[32768:32768 + n]
ok = (wg[:, 1] > 0) & (ph[:, 3] > 0)
t0 = wg[ok, 0].min()
us = lambda a: (a - t0) * 0.01
cols = [("start", wg[:, 0]), ("tables in LDS", wg[:, 3]), ("agent resident", ph[:, 0]), ("pass 1 done", ph[:, 1]), ("pass 2 done", ph[:, 2]),
        ("segments folded", ph[:, 3]), ("end", wg[:, 1])]
first = ok & (us(wg[:, 0]) < 5.0)
late = ok & ~first
life = (wg[:, 1] - wg[:, 0]) * 0.01
heavy = first & (life > np.percentile(life[first], 92))
for name, sel in (("first-round workgroups", first & ~heavy), ("the heaviest 8 % of them", heavy), ("second-round workgroups", late)):
    print(f"{name}: {int(sel.sum())}")
    prev = None
    for cname, c in cols:
        rel = (c[sel] - wg[sel, 0]) * 0.01
        print(f"   {cname:>16s}: +{rel.mean():6.2f} us after start (p95 {np.percentile(rel, 95):6.2f})" + ("" if prev is None else f"   phase {np.mean(rel - prev):6.2f}"))
        prev = rel
