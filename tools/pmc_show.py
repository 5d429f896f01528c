#!/usr/bin/env python3
"""tools/pmc_show.py <tag> [kernel-substring]: per-launch mean of every counter collected by tools/pmc_ab.sh, one column
per library variant."""
import csv, glob, os, re, sys
from collections import defaultdict
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "fo_sweep_queue_kernel"
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join("gpurun_out", f"{tag}_*_pmc*", "**", "*counter_collection.csv"), recursive=True):
    v = re.match(rf".*{tag}_(.*)_pmc\d+", f).group(1)
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            vals[r["Counter_Name"]][v].append(float(r["Counter_Value"]))
vs = sorted({v for c in vals for v in vals[c]})
print(f"{'counter':<44}" + "".join(f"{v:>16}" for v in vs))
for c in sorted(vals):
    print(f"{c:<44}" + "".join(f"{(sum(vals[c][v]) / len(vals[c][v]) if vals[c].get(v) else float('nan')):>16.5g}" for v in vs))
