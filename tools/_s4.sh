#!/bin/bash
# predict-kernel ablations: kernel time by variant (small step under rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
for v in base xp1 xp2 xp4 xp8 xp15; do
  if [ $v != base ]; then export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; fi
  rm -rf /tmp/xs_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xs_$v -o run -- python3 bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > /tmp/xs_$v.log 2>&1 || tail -5 /tmp/xs_$v.log
  python3 - $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/xs_%s/run_kernel_stats.csv' % sys.argv[1])))
import re
print(sys.argv[1], {(re.search(r'fo_\w+', r['Name']) or re.search(r'\w+', r['Name'])).group(0)[3:20]: round(float(r['AverageNs']) / 1e3, 2) for r in rows[:9]})
PY
done
