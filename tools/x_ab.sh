#!/bin/bash
# tools/x_ab.sh <mode> <variant>... : default vs experiment libraries (lib/variants), sustained clocks, one output mode
MODE=$1; shift
for v in default "$@" default; do
if [ $v = default ]; then unset FO_HIP_LIB; else export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; fi
timeout 300 python bench.py --steps 50 --warmup 150 --no-autotune --no-cpu-baseline --mode $MODE 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE $v step=%.4f kernel=%.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
