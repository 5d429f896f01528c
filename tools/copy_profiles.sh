#!/bin/bash
# after tools/collect_profiles.sh + tools/summarize_pmc.py ran on the GPU box (tools/_g12.sh): gpurun_out -> profiles/
set -e
cd $(dirname $0)/..
for t in r03_final r03_f64lists r03_reduced; do
  cp gpurun_out/${t}_stats/run_kernel_stats.csv profiles/${t}_kernel_stats.csv
  cp gpurun_out/summ/${t}_summary.csv profiles/${t}_summary.csv
  cp gpurun_out/summ/${t}_build.json profiles/
done
cp gpurun_out/r03_small_batch_stats/run_kernel_stats.csv profiles/r03_small_batch_kernel_stats.csv
cp gpurun_out/r03_spawn_rules_stats/run_kernel_stats.csv profiles/r03_spawn_rules_kernel_stats.csv
cp gpurun_out/r03_final_bench.json profiles/
python3 - <<'PY'
import json, csv
d = json.load(open('profiles/r03_final_bench.json'))
print('step', d['ms_per_step'], 'value', d['value'], 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'parity', d['parity']['ok'])
print({k: (d['config'][k]['ms_per_step'], d['config'][k].get('sweep_kernel_ms')) for k in ('f64_lists', 'reduced_outputs', 'small_batch')})
print('bench build', d['config']['build_id'][:12], 'profile build', json.load(open('profiles/r03_final_build.json'))['build_id'][:12])
for t in ['r03_final', 'r03_f64lists', 'r03_reduced']:
    for r in csv.DictReader(open(f'profiles/{t}_summary.csv')):
        if 'queue' in r['kernel'] and float(r['avg_ns']) > 3e5:
            print(t, r['kernel'], r['avg_ns'], 'W %.3f GB' % (float(r['WRITE_SIZE']) * 1024 / 1e9), '2F %.3f GB' % (float(r['FETCH_SIZE']) * 2 * 1024 / 1e9), r['SQ_INSTS_VALU'], r['SQ_INSTS_SALU'])
PY
