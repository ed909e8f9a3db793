#!/bin/bash
# usage: scripts/ab_nn_build.sh NAME "<-D flags>" — a variant of nn_forward.hip linked into alphagomoku_amd/libagx_NAME.so (the other objects as built)
cd "$(dirname "$0")/.."
name="$1"; flags="$2"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-ilp $flags -Iinclude -c ${NN_SRC:-alphagomoku_amd/csrc/nn_forward.hip} -o /tmp/nn_forward_$name.o || exit 1
objs=$(ls alphagomoku_amd/csrc/*.o | grep -v nn_forward.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alphagomoku_amd/libagx_$name.so $objs /tmp/nn_forward_$name.o -lz || exit 1
echo built $name
