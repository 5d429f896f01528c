#!/bin/bash
# tools/build_variant_asm.sh <name> <sed script> [hipcc flags...]: a tuning library whose SWEEP device code went through a
# textual rewrite of the compiler's assembly (hipcc -S -> sed -> assembler -> lld -> bundle -> host object), e.g.
#   tools/build_variant_asm.sh e64 's/v_cndmask_b32_e32 \(.*\), vcc$/v_cndmask_b32_e64 \1, vcc/'
set -e
NAME=$1; SED=$2; shift; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/frenetix-occlusion_amd/csrc; L=$R/frenetix-occlusion_amd/lib/variants; W=$L/$NAME; mkdir -p $W
B=/opt/rocm/lib/llvm/bin
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$C"
/opt/rocm/bin/hipcc $F -c $C/fo_api.hip -o $W/fo_api.o
/opt/rocm/bin/hipcc $F -ffp-contract=off -c $C/fo_scene.hip -o $W/fo_scene.o
/opt/rocm/bin/hipcc $F "$@" --cuda-device-only -S -o $W/sweep0.s $C/fo_sweep.hip
sed -e "$SED" $W/sweep0.s > $W/sweep.s
echo "lines changed: $(diff $W/sweep0.s $W/sweep.s | grep -c '^>')"
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/sweep.s -o $W/sweep.dev.o
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/sweep.out $W/sweep.dev.o
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/sweep.out -output=$W/sweep.hipfb
/opt/rocm/bin/hipcc $F "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/sweep.hipfb -c $C/fo_sweep.hip -o $W/fo_sweep.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libfo_hip_$NAME.so $W/fo_api.o $W/fo_scene.o $W/fo_sweep.o
echo $L/libfo_hip_$NAME.so
