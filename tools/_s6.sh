#!/bin/bash
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/xt -o run -- python3 bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > /tmp/xt.log 2>&1 || tail -5 /tmp/xt.log
python3 - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/xt/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 3 steps: find the last occurrences of the rays kernel
idx = [i for i, r in enumerate(rows) if 'fo_rays' in r['Kernel_Name']]
for s in idx[150:153]:
    t0 = int(rows[s]['Start_Timestamp']); prev_end = None
    out = []
    for r in rows[s:s + 8]:
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        nm = re.search(r'fo_(\w+?)_kernel', r['Kernel_Name'])
        out.append('%s %.1f-%.1f (gap %.1f)' % (nm.group(1) if nm else '?', (a - t0) / 1e3, (b - t0) / 1e3, 0 if prev_end is None else (a - prev_end) / 1e3))
        prev_end = b
    print(' | '.join(out))
PY
