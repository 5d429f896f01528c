#!/usr/bin/env python3
"""Condense rocprofv3 output dirs (gpurun_out/<tag>_stats, <tag>_pmcN) into profiles/<tag>_summary.{csv,md}:
per kernel: calls, average duration (kernel trace) and the per-launch mean of every collected counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles"


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


dur = defaultdict(list)
for f in glob.glob(os.path.join(src, f"{tag}_stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
ctr = defaultdict(lambda: defaultdict(list))
meta = {}
for f in glob.glob(os.path.join(src, f"{tag}_pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
os.makedirs(dst, exist_ok=True)
names = sorted(dur, key=lambda k: -sum(dur[k]))
counters = sorted({c for k in ctr for c in ctr[k]})
with open(os.path.join(dst, f"{tag}_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_ns", "min_ns", "max_ns", "grid", "wg", "lds", "vgpr", "sgpr"] + counters)
    for k in names:
        if not k.startswith("fo_"):
            continue
        d = dur[k]
        w.writerow([k, len(d), round(sum(d) / len(d)), min(d), max(d)] + list(meta.get(k, [""] * 5)) +
                   [("%.6g" % (sum(ctr[k][c]) / len(ctr[k][c])) if ctr[k].get(c) else "") for c in counters])
if os.path.exists(os.path.join(src, f"{tag}_build.json")):     # the library the profile belongs to (bench.py checks it)
    import shutil
    shutil.copy(os.path.join(src, f"{tag}_build.json"), os.path.join(dst, f"{tag}_build.json"))
print(open(os.path.join(dst, f"{tag}_summary.csv")).read())
