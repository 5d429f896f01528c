#!/bin/bash
python -m pytest tests -x -q -m gpu -k "${1:-split or configs or rules_step or scene}" 2>&1 | tail -3
for i in 1 2; do
python bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 200 --steps 600 --no-extras 2>/dev/null | python tools/_pj.py small
done
python bench.py --no-cpu-baseline 2>/dev/null | python tools/_pj.py headline
