#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> frenetix-occlusion_amd/lib/variants/libfo_hip_<name>.so (tuning builds)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/frenetix-occlusion_amd/csrc; L=$R/frenetix-occlusion_amd/lib/variants; mkdir -p $L/$NAME
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C"
/opt/rocm/bin/hipcc $F -c $C/fo_api.hip -o $L/$NAME/fo_api.o
/opt/rocm/bin/hipcc $F "$@" -c $C/fo_sweep.hip -o $L/$NAME/fo_sweep.o
/opt/rocm/bin/hipcc $F -ffp-contract=off -c $C/fo_scene.hip -o $L/$NAME/fo_scene.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libfo_hip_$NAME.so $L/$NAME/*.o
echo $L/libfo_hip_$NAME.so
