for v in default x1 x32 x64 x103 default; do
if [ $v = default ]; then unset FO_HIP_LIB; else export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so; fi
timeout 300 python bench.py --steps 5 --warmup 2 --no-autotune --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['small_batch']; print('$v small-batch step=%.4f sweep=%.4f' % (s['ms_per_step'], s['sweep_kernel_ms']))"
done
