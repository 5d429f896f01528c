#!/bin/bash
# tools/build_variant_scene.sh <name> <extra hipcc flags for fo_scene.hip...> -> lib/variants/libfo_hip_<name>.so (tuning builds,
# e.g. -DFO_RULE_TRACE=1: wall-clock stamps of the dynamic spawn rule's phases, read by tools/spawn_rules_bench.py)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/frenetix-occlusion_amd/csrc; L=$R/frenetix-occlusion_amd/lib/variants; mkdir -p $L/$NAME
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C"
/opt/rocm/bin/hipcc $F -c $C/fo_api.hip -o $L/$NAME/fo_api.o
cp $R/frenetix-occlusion_amd/lib/fo_sweep.o $L/$NAME/fo_sweep.o
/opt/rocm/bin/hipcc $F -ffp-contract=off "$@" -c $C/fo_scene.hip -o $L/$NAME/fo_scene.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libfo_hip_$NAME.so $L/$NAME/*.o
echo $L/libfo_hip_$NAME.so
