for apw in 1 2 4 8; do
  FO_SWEEP_APW=$apw timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('apw=$apw kernel_ms=%.3f grid=%d step=%.3f' % (d['roofline']['kernel_ms'], d['roofline']['grid'], d['ms_per_step']))"
done
