#!/bin/bash
# tools/pmc_ab.sh <tag> <variant|default> "<counters>" ["<counters>" ...]: PMC passes of the bench step with one library
# variant (lib/variants), sweep kernel only -> gpurun_out/<tag>_<variant>_pmcN/
TAG=$1; V=$2; shift 2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
if [ $V = default ]; then unset FO_HIP_LIB; else export FO_HIP_LIB=$PWD/frenetix-occlusion_amd/lib/variants/libfo_hip_$V.so; fi
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/${TAG}_${V}_pmc$i -o run -- python3 bench.py --steps 5 --warmup 30 --no-cpu-baseline --no-autotune > $OUT/${TAG}_${V}_pmc$i.log 2>&1
done
