#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + PMC passes of the bench command.
# usage: tools/collect_profiles.sh <tag> [bench args...]     -> gpurun_out/<tag>_{stats,pmcN}/...
# PMC passes are separate runs (rocprofv3 refuses nothing here: --pmc only with --kernel-trace).
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-autotune --no-extras $*"
# the kernel-trace pass runs the bench step 200 times at one kernel configuration (no agents-per-wave selection pass,
# which would mix four configurations into the average; 150 untimed steps bring the clocks up like that pass does) so
# that its average kernel duration is comparable with the HIP-event figure bench.py prints
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 bench.py --no-cpu-baseline --no-autotune --no-extras --warmup 150 $* > $OUT/${TAG}_stats.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o run -- python3 bench.py $ARGS > $OUT/${TAG}_pmc$i.log 2>&1
done
# which library these numbers belong to (bench.py prints roofline.traffic only when this id equals the loaded library's)
LISTS=f32x; case " $* " in *" --lists f64 "*) LISTS=f64;; *" --lists f32 "*) LISTS=f32;; esac
python3 - > $OUT/${TAG}_build.json <<PY
import json, sys
sys.path.insert(0, "frenetix-occlusion_amd")
from frenetix_occlusion import _native as N
print(json.dumps({"build_id": N.build_id(), "lists": "$LISTS", "bench_args": "$*"}))
PY
ls -R $OUT | grep -c csv
