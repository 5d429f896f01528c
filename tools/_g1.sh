set -x
python -m pytest tests/test_sweep_gpu.py -m gpu -x -q -k "float32" 2>&1 | tail -5
python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "headline" 2>&1 | tail -5
python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; tail -3 gpurun_out/r03a_bench.err; cat gpurun_out/r03a_bench.json
FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_widef32.so python bench.py --no-cpu-baseline --no-extras --warmup 100 > gpurun_out/r03a_bench_wide.json 2>/dev/null; cat gpurun_out/r03a_bench_wide.json
python bench.py --no-cpu-baseline --no-extras --warmup 100 > gpurun_out/r03a_bench_f32.json 2>/dev/null; cat gpurun_out/r03a_bench_f32.json
