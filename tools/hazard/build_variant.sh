#!/bin/bash
# Builds a variant of librmav.so whose policy translation unit (csrc/rmav_policy_abi.hip: the MFMA rollout kernels) is compiled
# with other flags and / or with its gfx950 assembly PATCHED before assembling - for the stale-read investigation
# (profiles/r05/packed_f32_hazard.md).  The product build is the Makefile's; these go to tools/hazard/_build/ (not tracked).
# Usage: tools/hazard/build_variant.sh <name> "<extra hipcc flags for the policy TU>" [<patch spec for tools/hazard/patch_asm.py> ...]
#   e.g. build_variant.sh slp_vf1 "-mllvm -amdgpu-mfma-vgpr-form=1"                      (SLP vectoriser on: the reproducer)
#        build_variant.sh slp_vf1_nop3 "-mllvm -amdgpu-mfma-vgpr-form=1" after:pk_opsel:3 (s_nop 3 after every cross-register packed op)
set -e
NAME=$1; EXTRA=$2; shift 2
HERE=$(cd $(dirname $0) && pwd); PKG=$HERE/../../reinmav-gym_amd; OUT=$HERE/_build/$NAME; mkdir -p $OUT
LLVM=/opt/rocm/lib/llvm/bin
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-command-line-argument"
cd $PKG
[ -f build/rmav_abi.o ] || make build/rmav_abi.o
/opt/rocm/bin/hipcc $BASE $EXTRA -S --cuda-device-only -o $OUT/dev.s csrc/rmav_policy_abi.hip

if [ $# -gt 0 ]; then python3 $HERE/patch_asm.py $OUT/dev.s "$@" > $OUT/patch.log; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $OUT/dev.s -o $OUT/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $OUT/dev.out $OUT/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$OUT/dev.out -output=$OUT/dev.hipfb
/opt/rocm/bin/hipcc $BASE $EXTRA --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $OUT/dev.hipfb -c -o $OUT/host.o csrc/rmav_policy_abi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/librmav.so build/rmav_abi.o $OUT/host.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -f $OUT/dev.o $OUT/dev.out $OUT/dev.hipfb $OUT/host.o
echo "$NAME: $(ls -la $OUT/librmav.so | awk '{print $5}') bytes; $(cat $OUT/patch.log 2>/dev/null | tail -1)"
