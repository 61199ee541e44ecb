#!/bin/bash
# Builds a variant of the library with extra device-compile flags: tools/build_variant.sh <name> <flags...>
# -> raytracing-1w_amd/librt1w_<name>.so (select it with RT1W_LIB=...).  Experiments only.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/raytracing-1w_amd/csrc
make -s scene.o scenes.o output.o jit.o context_ref.o context_f32.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -Wno-unused-result -mllvm -spec-exec-max-speculation-cost=0 -mllvm -structurizecfg-skip-uniform-regions=1 -mllvm -simplifycfg-hoist-common=false -mllvm -amdgpu-sdwa-peephole=0 -I$ROOT/include -I. "$@" -c context.hip -o context_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 scene.o scenes.o output.o jit.o context_ref.o context_f32.o context_$NAME.o -Wl,--version-script=librt1w.map -o $ROOT/raytracing-1w_amd/librt1w_$NAME.so
echo built librt1w_$NAME.so
