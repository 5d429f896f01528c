#!/bin/bash
# tools/build_variant_src.sh <name> <fo_sweep source file> <extra hipcc flags...> -> lib/variants/libfo_hip_<name>.so
# like build_variant.sh, but the sweep translation unit comes from another file (e.g. an older revision: git show REV:... > /tmp/x.hip)
set -e
NAME=$1; SRC=$2; shift; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/frenetix-occlusion_amd/csrc; L=$R/frenetix-occlusion_amd/lib/variants; mkdir -p $L/$NAME
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C"
/opt/rocm/bin/hipcc $F -c $C/fo_api.hip -o $L/$NAME/fo_api.o
/opt/rocm/bin/hipcc $F "$@" -x hip -c $SRC -o $L/$NAME/fo_sweep.o
/opt/rocm/bin/hipcc $F -ffp-contract=off -c $C/fo_scene.hip -o $L/$NAME/fo_scene.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libfo_hip_$NAME.so $L/$NAME/*.o
echo $L/libfo_hip_$NAME.so
