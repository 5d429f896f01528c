#!/bin/bash
# tools/lists_ab.sh "<bench args>" <variant>... : default vs experiment libraries (lib/variants) at sustained clocks, one bench
# configuration; every library twice, in two passes over the list (boxes differ by +-3 %, and one box drifts by ~1 % within
# minutes: only runs of one call compare, and only differences that show in both passes count), e.g.
#   bash tools/lists_ab.sh "--lists f32x" r4 ord3n
ARGS=$1; shift
STEPS=${AB_STEPS:-120}
for pass in 1 2; do
for v in default "$@"; do
if [ $v = default ]; then unset FO_HIP_LIB; else export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; fi
timeout 300 python bench.py --steps $STEPS --warmup 150 --no-autotune --no-cpu-baseline --no-extras $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pass $pass $v step=%.4f kernel=%.4f p50=%.4f frac=%.3f' % (d['ms_per_step'], r['kernel_ms'], r.get('kernel_ms_p50') or 0, r['frac']))"
done
done
