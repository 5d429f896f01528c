#!/bin/bash
# sweep-kernel time per output mode at sustained clocks (run on the GPU box): full / pair / reduced
for mode in full pair reduced; do
  python bench.py --no-cpu-baseline --mode $mode --steps 100 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$mode', 'step %.4f ms' % d['ms_per_step'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], 'apw', d['config']['agents_per_wave'], d['config']['setup_autotune_ms_per_step'])"
done
