#!/bin/bash
# usage: scripts/build_engine_variant.sh NAME "<-D flags>" — a variant of engine.hip linked into alphagomoku_amd/libagx_NAME.so (the other objects as built);
# run anything with it through AGX_NO_BUILD=1 AGX_LIB_PATH=alphagomoku_amd/libagx_NAME.so (AGX_NO_BUILD skips the source-hash check for such developer variants)
cd "$(dirname "$0")/.."
name="$1"; flags="$2"
# RELINK=1: only link /tmp/engine_NAME.o again (the other objects have been rebuilt meanwhile)
if [ -z "$RELINK" ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-ilp $flags -Iinclude -c ${ENGINE_SRC:-alphagomoku_amd/csrc/engine.hip} -o /tmp/engine_$name.o || exit 1
fi
objs=$(ls alphagomoku_amd/csrc/*.o | grep -v "csrc/engine.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alphagomoku_amd/libagx_$name.so $objs /tmp/engine_$name.o -lz || exit 1
echo built $name
