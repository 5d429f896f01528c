#!/bin/bash
# small-step A/B over agents per workgroup of the split form
python -m pytest tests/test_sweep_gpu.py -x -q -k "horizon_split" 2>&1 | tail -3
for v in 1 2 3 4 1 2 3 4; do
  FO_SWEEP_SPLIT_APW=$v python bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 200 --steps 600 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('apw $v', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['roofline'].get('grid'))"
done
