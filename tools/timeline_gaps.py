#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 --kernel-trace CSV: busy time, idle gaps between kernels, and the kernels of
one steady-state step in launch order.  usage: tools/timeline_gaps.py <run_kernel_trace.csv> [kernels-per-step marker]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "fo_reduce_kernel"
# steps end with the marker kernel
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
steps = [(ends[i - 1] + 1, ends[i]) for i in range(1, len(ends))]
only = sys.argv[3] if len(sys.argv) > 3 else "fo_sweep_queue_kernel<true, true"   # full-output steps
steps = [(a, b) for a, b in steps if any(only in r["Kernel_Name"] for r in rows[a:b + 1])]
steps = steps[len(steps) // 2:]          # steady state
tot = busy = 0.0
for a, b in steps:
    t0 = int(rows[a - 1]["End_Timestamp"]) if a > 0 else int(rows[a]["Start_Timestamp"])
    tot += int(rows[b]["End_Timestamp"]) - t0
    busy += sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b + 1])
n = len(steps)
print(f"{n} steps: wall {tot / n / 1e3:.1f} us/step, kernels busy {busy / n / 1e3:.1f} us/step, gaps {(tot - busy) / n / 1e3:.1f} us/step")
a, b = steps[-1]
prev = int(rows[a - 1]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  gap {(s - prev) / 1e3:6.1f} us  run {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:70]}")
    prev = e
