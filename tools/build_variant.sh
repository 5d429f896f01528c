#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> frenetix-occlusion_amd/lib/variants/libfo_hip_<name>.so (tuning builds)
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/frenetix-occlusion_amd/csrc; L=$R/frenetix-occlusion_amd/lib/variants; mkdir -p $L/$NAME
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C"
/opt/rocm/bin/hipcc $F -c $C/fo_api.hip -o $L/$NAME/fo_api.o
# the sweep's own backend flags of the product build (__graft_entry__.HIP_SOURCES) unless SWEEP_FLAGS is set (possibly empty):
# a variant differs from the default library in what its name says and in nothing else
if [ -z "${SWEEP_FLAGS+x}" ]; then SWEEP_FLAGS=$(cd $R && python3 -c "import __graft_entry__ as g; print(' '.join(g.HIP_SOURCES['fo_sweep.hip']))"); fi
/opt/rocm/bin/hipcc $F $SWEEP_FLAGS "$@" -c $C/fo_sweep.hip -o $L/$NAME/fo_sweep.o
/opt/rocm/bin/hipcc $F -ffp-contract=off -c $C/fo_scene.hip -o $L/$NAME/fo_scene.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libfo_hip_$NAME.so $L/$NAME/*.o
echo $L/libfo_hip_$NAME.so
