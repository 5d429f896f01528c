# tools/warm_ab.sh <variant>... : default vs variant libraries at sustained clocks (150 untimed steps first, no set-up pass)
for v in default "$@" default; do
if [ $v = default ]; then unset FO_HIP_LIB; else export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; fi
timeout 300 python bench.py --steps 50 --warmup 150 --no-autotune --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v step=%.4f kernel=%.4f frac=%.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
