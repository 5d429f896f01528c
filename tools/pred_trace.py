#!/usr/bin/env python3
"""Phases of one workgroup of the phantom prediction kernel (tuning build: tools/build_variant_scene.sh ptrace -DFO_PRED_TRACE=<block>).
usage (GPU box): FO_HIP_LIB=.../libfo_hip_ptrace.so python tools/pred_trace.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import numpy as np
import bench

import torch
r = bench.small_batch_step(0, steps=50)
torch.cuda.synchronize()
from frenetix_occlusion import _native as N
lib = N.load()
t = (ctypes.c_longlong * 16)()
assert lib.fo_debug_pred_ticks(t) == 0
t = np.array(list(t), dtype=np.int64)
names = {0: "start", 1: "counts in LDS", 2: "block found", 3: "flags read", 4: "pick + heading", 5: "lanelet", 6: "route staged", 7: "prediction written", 9: "table rows issued"}
t0 = t[0]
print("step", round(r["ms_per_step"], 4))
for i in sorted(names):
    if t[i]:
        print(f"{names[i]:>22s}: {(t[i] - t0) * 0.01:6.2f} us")

if hasattr(lib, "fo_debug_ray_ticks"):
    t = (ctypes.c_longlong * 16)()
    assert lib.fo_debug_ray_ticks(t) == 0
    t = np.array(list(t), dtype=np.int64)
    rn = {0: "start", 1: "direction known", 5: "first boxes culled", 6: "pieces scanned", 2: "obstacle sides", 3: "wave minimum", 4: "workgroup barrier", 7: "written"}
    print("ray workgroup (same block index), wave 0; surviving chunks of the first group:", int(t[12]))
    for i in (0, 1, 5, 6, 2, 3, 4, 7):
        if t[i]:
            print(f"{rn[i]:>22s}: {(t[i] - t[0]) * 0.01:6.2f} us")
