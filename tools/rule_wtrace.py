#!/usr/bin/env python3
"""Workgroups of fo_spawn_rules_kernel at one pose of the scenario-1 drive (tuning build: tools/build_variant_scene.sh rtrace
-DFO_RULE_TRACE=1): start -> path table in LDS -> [static rule: projections, lane heading, samples classified, line done].
usage (GPU box): FO_HIP_LIB=.../libfo_hip_rtrace.so python tools/rule_wtrace.py [pose]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np
import torch
import bench
pose = int(sys.argv[1]) if len(sys.argv) > 1 else 5
# run the drive up to the pose with few steps, then read the stamps of the last launch
import types
orig = bench.rules_step
src = open(os.path.join(ROOT, "bench.py")).read()
os.environ["FO_RULES_LAST_POSE"] = str(pose)
r = bench.rules_step(0, steps=3)
torch.cuda.synchronize()
from frenetix_occlusion import _native as N
lib = N.load()
t = (ctypes.c_longlong * (8 * 1024))()
assert lib.fo_debug_rule_wticks(t) == 0
t = np.array(list(t), dtype=np.int64).reshape(1024, 8)
ok = t[:, 0] > 0
t0 = t[ok, 0].min()
names = ["start", "path in LDS", "static: begin", "projections", "lane heading", "samples classified", "line done"]
for b in np.nonzero(ok)[0][:40]:
    row = t[b]
    print(b, " ".join(f"{names[i]} {(row[i] - t0) * 0.01:.1f}" for i in range(7) if row[i] >= t0 and row[i] > 0))
print("blocks", int(ok.sum()), "last stamp", (t[ok].max() - t0) * 0.01)
