#!/bin/bash
# tools/collect_round.sh <rNN>: everything profiles/<rNN>_* is made of, on the GPU box (gpurun -- 'bash tools/collect_round.sh r04'):
# kernel trace + PMC passes of the headline step in its output modes, kernel traces of the small step, the rules step and the
# spawn-rule bench, the un-profiled default bench line; condensed by tools/summarize_pmc.py into gpurun_out/summ/.
R=${1:-r06}
export TMPDIR=/tmp
bash tools/collect_profiles.sh ${R}_final                  # the headline: --lists f32x (float64 arithmetic, float32 storage)
bash tools/collect_profiles.sh ${R}_f64lists --lists f64
bash tools/collect_profiles.sh ${R}_f32 --lists f32        # side leg: float32 arithmetic for the harm entries away from the gate
bash tools/collect_profiles.sh ${R}_reduced --mode reduced
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_small_batch_stats -o run -- python3 bench.py --scene scenario1 --M 2000 --A 32 --mode reduced --no-cpu-baseline --no-autotune --warmup 100 --steps 200 --no-extras > gpurun_out/${R}_small_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_spawn_rules_stats -o run -- python3 tools/spawn_rules_bench.py > gpurun_out/${R}_spawn_rules.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_rules_step_stats -o run -- python3 tools/rules_step_bench.py > gpurun_out/${R}_rules_step.log 2>&1
# the future-visibility sweep at the stage's 720-ray fan: kernel trace + two counter passes
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_fv_stats -o run -- python3 tools/future_visibility_bench.py > gpurun_out/${R}_fv.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${R}_fv_pmc1 -o run -- python3 tools/future_visibility_bench.py 2000 > gpurun_out/${R}_fv_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/${R}_fv_pmc2 -o run -- python3 tools/future_visibility_bench.py 2000 > gpurun_out/${R}_fv_pmc2.log 2>&1
for t in ${R}_final ${R}_f64lists ${R}_f32 ${R}_reduced ${R}_fv; do python3 tools/summarize_pmc.py $t gpurun_out gpurun_out/summ > /dev/null; done
# the default bench line quotes the committed counters of the library it runs (roofline.bound / traffic / valu_issue_frac): put
# this run's summaries where it looks for them (on the box's copy of the tree; tools/copy_profiles.sh does the same at home)
for t in ${R}_final ${R}_f64lists ${R}_f32 ${R}_reduced; do cp gpurun_out/summ/${t}_summary.csv gpurun_out/summ/${t}_build.json profiles/; done
python bench.py > gpurun_out/${R}_final_bench.json 2> gpurun_out/${R}_final_bench.err
# keep the merge small: drop the raw counter dumps
find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*agent_info.csv" -delete
ls gpurun_out/summ; du -sh gpurun_out
python3 - <<PY
import json
d = json.load(open("gpurun_out/${R}_final_bench.json"))
c = d["config"]
print("step %.4f value %.4g kernel %.4f frac %.3f dtype %s parity %s" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["dtype"], d["parity"]["ok"]))
print({k: round(c[k]["sweep_kernel_ms"], 4) for k in ("f64_lists", "f32_lists", "reduced_outputs")}, "small", round(c["small_batch"]["ms_per_step"], 4), "rules", round(c["rules_step"]["ms_per_step"], 4), round(c["rules_step"]["ms_per_step_max"], 4))
print("shards", {k: (round(v["reduced"]["ms_per_step"], 4), round(v["full"]["ms_per_step"], 4)) for k, v in c["shard_probe"]["shards"].items()}, "allgather", c["shard_probe"]["allgather_ms"])
PY
