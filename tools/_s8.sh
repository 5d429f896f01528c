#!/bin/bash
for e in 1 8 1 8; do
FO_BENCH_TIME_EVERY=$e python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python tools/_pj.py "every$e"
done
