#!/bin/bash
export TMPDIR=/tmp
export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_dup.so
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/xd -o run -- python3 bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > /tmp/xd.log 2>&1 || tail -5 /tmp/xd.log
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/xd/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'fo_spawn_predict' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows][-400:]
import statistics
print('first launch of a pair', round(statistics.mean(d[0::2]), 2), 'second', round(statistics.mean(d[1::2]), 2))
PY
