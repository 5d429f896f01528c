for v in tq15w2 tq15w3; do for mode in reduced full; do
FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$v.so timeout 200 python bench.py --T 16 --steps 20 --warmup 3 --no-cpu-baseline --mode $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v mode=$mode kernel_ms=%.3f' % (d['roofline']['kernel_ms']))"
done; done
