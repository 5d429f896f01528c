for ab in 0 1 2 4 3 5 6 7; do
  for mode in reduced full; do
    FO_SWEEP_ABLATE=$ab timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate=$ab mode=$mode kernel_ms=%.3f' % d['roofline']['kernel_ms'])"
  done
done
