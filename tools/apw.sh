for apw in 1 2 4 8 16; do
  for mode in reduced full; do
    FO_SWEEP_APW=$apw timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('apw=$apw mode=$mode kernel_ms=%.3f grid=%d' % (d['roofline']['kernel_ms'], d['roofline']['grid']))"
  done
done
