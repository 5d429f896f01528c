#!/bin/bash
# tools/step_timeline.sh [bench args]: the launches of three planning steps from the middle of a bench run, with start / end
# relative to the step's first launch and the GAP in front of each (rocprofv3 --kernel-trace).  Default: the reference-size
# step (bench.py --scene scenario1 --M 2000 --A 32 --mode reduced).  This is how the 5.5 us holes that HIP-event records
# leave in front of the timed launch and the launch after it were found.
export TMPDIR=/tmp
ARGS="${@:---scene scenario1 --M 2000 --A 32 --mode reduced}"
rm -rf /tmp/fo_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fo_tl -o run -- python3 bench.py $ARGS --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > /tmp/fo_tl.log 2>&1 || tail -5 /tmp/fo_tl.log
python3 - <<'PY'
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob('/tmp/fo_tl/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fo_rays' in r['Kernel_Name']]
for s, e in zip(idx[150:153], idx[151:154]):
    t0, prev_end, out = int(rows[s]['Start_Timestamp']), None, []
    for r in rows[s:e]:
        a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        nm = re.search(r'fo_(\w+?)_kernel', r['Kernel_Name'])
        out.append('%s %.1f-%.1f (gap %.1f)' % (nm.group(1) if nm else '?', (a - t0) / 1e3, (b - t0) / 1e3, 0 if prev_end is None else (a - prev_end) / 1e3))
        prev_end = b
    print(' | '.join(out))
PY
