#!/usr/bin/env python3
"""The `config.rules_step` leg of bench.py on its own (the reference's spawn rule families as a device-resident planning step,
scenario 1, 61 poses) -- for `rocprofv3 --kernel-trace --stats -- python3 tools/rules_step_bench.py`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "frenetix-occlusion_amd"))
import bench  # noqa: E402

r = bench.rules_step(0, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20)
print(json.dumps({k: v for k, v in r.items() if k != "poses"}))
