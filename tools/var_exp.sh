timeout 300 python -m pytest tests/test_sweep_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in default "$@"; do for mode in reduced full; do
if [ $v = default ]; then unset FO_HIP_LIB; else export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; fi
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v mode=$mode kernel_ms=%.3f grid=%d block=%d' % (d['roofline']['kernel_ms'], d['roofline']['grid'], d['roofline']['block']))"
done; done
