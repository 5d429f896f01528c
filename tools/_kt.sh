#!/bin/bash
# kernel times of the small step (rocprofv3 --kernel-trace --stats) for the libraries named on the command line ("base" = the tree's)
export TMPDIR=/tmp
for v in "$@"; do
  if [ $v != base ]; then export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; else unset FO_HIP_LIB; fi
  rm -rf /tmp/xs_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xs_$v -o run -- python3 bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > /tmp/xs_$v.log 2>&1 || tail -5 /tmp/xs_$v.log
  python3 - $v <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open('/tmp/xs_%s/run_kernel_stats.csv' % sys.argv[1])))
d = {(re.search(r'fo_\w+', r['Name']) or re.search(r'\w+', r['Name'])).group(0)[3:20]: (round(float(r['AverageNs']) / 1e3, 2), int(r['Calls'])) for r in rows[:9]}
print(sys.argv[1], {k: v[0] for k, v in d.items()}, 'sum of per-step kernels', round(sum(v[0] * v[1] for v in d.values() if v[1] >= 300) / 321, 2))
PY
done
