#!/usr/bin/env python3
"""Per-wave time line of the persistent sweep (tuning build -DFO_TRACE=1): life time, time inside the work source
(next_agent), agents taken.  usage (GPU box): FO_HIP_LIB=.../libfo_hip_trace.so python tools/wave_trace.py [bench args]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "wave_trace.bin")
env = dict(os.environ, FO_SWEEP_TRACE=out, FO_SWEEP_TRACE_DUMP="1")
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--no-autotune", "--steps", "3",
                "--warmup", "30"] + sys.argv[1:], env=env, stdout=subprocess.DEVNULL, check=True)
t = np.fromfile(out, dtype=np.int64).reshape(-1, 4)
wg, wv = t[:32768], t[32768:]
wg = wg[wg[:, 1] > 0]
wv = wv[wv[:, 1] > 0]
t0 = wg[:, 0].min()
print(f"workgroups {len(wg)}: span {(wg[:, 1].max() - t0) * 0.01:.1f} us")
life = (wv[:, 1] - wv[:, 0]) * 0.01
sched = wv[:, 2] * 0.01
print(f"waves {len(wv)}: life mean {life.mean():.1f} us (p5 {np.percentile(life, 5):.1f}, p95 {np.percentile(life, 95):.1f}); "
      f"end of the waves' last agent: p5 {np.percentile((wv[:, 1] - t0) * 0.01, 5):.1f} p50 {np.percentile((wv[:, 1] - t0) * 0.01, 50):.1f} "
      f"max {((wv[:, 1] - t0) * 0.01).max():.1f} us")
print(f"time inside the work source per wave: mean {sched.mean():.1f} us (p95 {np.percentile(sched, 95):.1f}) = "
      f"{sched.sum() / life.sum():.3f} of the waves' life; calls per wave mean {wv[:, 3].mean():.1f} (min {wv[:, 3].min()}, max {wv[:, 3].max()}); "
      f"per call {sched.sum() / wv[:, 3].sum():.2f} us")
