#!/bin/bash
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x_small_stats -o run -- python3 bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > gpurun_out/x_small.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/x_small_stats/run_kernel_stats.csv')))
tot=0
for r in rows[:9]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2)); 
PY
rm -f gpurun_out/x_small_stats/*trace.csv gpurun_out/x_small_stats/*agent_info.csv
