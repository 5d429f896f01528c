#!/bin/bash
# tools/copy_profiles.sh <rNN>: after tools/collect_round.sh <rNN> ran on the GPU box: gpurun_out -> profiles/
set -e
R=${1:-r06}
cd $(dirname $0)/..
for t in ${R}_final ${R}_f64lists ${R}_f32 ${R}_reduced; do
  cp gpurun_out/${t}_stats/run_kernel_stats.csv profiles/${t}_kernel_stats.csv
  cp gpurun_out/summ/${t}_summary.csv profiles/${t}_summary.csv
  cp gpurun_out/summ/${t}_build.json profiles/
done
for t in small_batch spawn_rules rules_step fv; do cp gpurun_out/${R}_${t}_stats/run_kernel_stats.csv profiles/${R}_${t}_kernel_stats.csv; done
cp gpurun_out/summ/${R}_fv_summary.csv profiles/${R}_fv_summary.csv
cp gpurun_out/${R}_fv.log profiles/${R}_fv_bench.txt
cp gpurun_out/${R}_final_bench.json profiles/
python3 - <<PY
import json, csv
R = "$R"
d = json.load(open(f"profiles/{R}_final_bench.json"))
print("step", d["ms_per_step"], "value", d["value"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "parity", d["parity"]["ok"])
print("bench build", d["config"]["build_id"][:12], "profile build", json.load(open(f"profiles/{R}_final_build.json"))["build_id"][:12])
for t in [f"{R}_final", f"{R}_f64lists", f"{R}_f32", f"{R}_reduced"]:
    for r in csv.DictReader(open(f"profiles/{t}_summary.csv")):
        if "queue" in r["kernel"] and float(r["avg_ns"]) > 3e5:
            print(t, r["kernel"], r["avg_ns"], "W %.3f GB" % (float(r["WRITE_SIZE"]) * 1024 / 1e9), "2F %.3f GB" % (float(r["FETCH_SIZE"]) * 2 * 1024 / 1e9), r["SQ_INSTS_VALU"], r["SQ_INSTS_SALU"], r["GRBM_GUI_ACTIVE"])
PY
