#!/usr/bin/env python3
"""Workgroup timeline of one sweep launch (tuning build -DFO_TRACE=1: tools/build_variant.sh trace -DFO_TRACE=1).
usage (GPU box): FO_HIP_LIB=.../libfo_hip_trace.so python tools/wg_trace.py [bench args]  -> gpurun_out/wg_trace.txt
Prints how many workgroups are resident over time (10 ns ticks of the 100 MHz wall clock), their life times, and the
share of the kernel's span during which the chip is full."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "wg_trace.bin")
env = dict(os.environ, FO_SWEEP_TRACE=out, FO_SWEEP_TRACE_DUMP="1")
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--no-autotune", "--steps", "3",
                "--warmup", "30"] + sys.argv[1:], env=env, stdout=subprocess.DEVNULL, check=True)
t = np.fromfile(out, dtype=np.int64).reshape(-1, 4)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01     # microseconds
span = en.max()
life = en - st
print(f"workgroups {len(t)}  kernel span {span:.1f} us  life mean {life.mean():.1f} us  p5 {np.percentile(life, 5):.1f}  "
      f"p50 {np.percentile(life, 50):.1f}  p95 {np.percentile(life, 95):.1f}  max {life.max():.1f}")
xcc = (t[:, 2] >> 32) & 0xF
print("workgroups per XCC:", np.bincount(xcc.astype(int)))
grid = np.linspace(0, span, 41)
res = [(int(((st <= g) & (en > g)).sum())) for g in grid]
print("resident workgroups over time (40 steps):", res)
full = max(res)
print(f"first start spread {st[np.argsort(st)[:full]].max():.1f} us; time with >= 95 % of {full} resident: "
      f"{sum(r >= 0.95 * full for r in res) / len(res):.2f} of the span; last 10 % of workgroups end after {np.percentile(en, 90):.1f} us")
order = np.argsort(st)
print("start of the i-th workgroup (us), every 256th:", np.round(st[order][::256], 1).tolist())
# life time by workgroup position in the launch (chunk-major: the tapered phases come last)
n = len(t)
idx = np.arange(n)
for lo, hi in ((0, int(0.5 * n)), (int(0.5 * n), int(0.8 * n)), (int(0.8 * n), int(0.9 * n)), (int(0.9 * n), n)):
    sel = slice(lo, hi)
    print(f"workgroups {lo}..{hi}: life mean {life[sel].mean():.1f} us  min {life[sel].min():.1f}  max {life[sel].max():.1f}  start mean {st[sel].mean():.1f}  end mean {en[sel].mean():.1f}")
fill = (t[:, 3] - t[:, 0]) * 0.01
ok = t[:, 3] > 0
if ok.any():
    print(f"start -> tables in LDS (us): mean {fill[ok].mean():.2f}  p50 {np.percentile(fill[ok], 50):.2f}  p95 {np.percentile(fill[ok], 95):.2f}  max {fill[ok].max():.2f}; "
          f"first 756 workgroups mean {fill[ok][:756].mean():.2f}, the rest {fill[ok][756:].mean():.2f}")
