#!/bin/bash
for e in 1 1000 8 1 1000 8; do
FO_BENCH_TIME_EVERY_SMALL=$e python - <<'PY'
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'frenetix-occlusion_amd')
import bench
r = bench.small_batch_step(0, steps=300)
print('every', os.environ['FO_BENCH_TIME_EVERY_SMALL'], 'ms_per_step', round(r['ms_per_step'], 4), [round(x, 4) for x in r['ms_per_step_runs']], 'stage calls', round(r['ms_per_step_stage_calls'], 4), 'kernel', round(r['sweep_kernel_ms'], 4))
PY
done
