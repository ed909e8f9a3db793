#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side C++ (GPU sanitizers are not available on the pool): the oracle
# (oracle/ag_*.cpp) and the host halves of libagx.so (table builder, game buffer / sample framing, host utilities; the HIP objects are linked
# in as built), driven by the whole CPU test suite.  Output: the suite's tail + every sanitizer report.
cd "$(dirname "$0")/.."
out=${1:-/tmp/agx_san}
mkdir -p $out
FLAGS="-std=c++17 -O1 -g -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ $FLAGS -shared -o $out/libagoracle_san.so oracle/ag_*.cpp -lpthread || exit 1
objs=""
for f in tables_host host_util game_buffer build_id; do
  g++ $FLAGS -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c alphagomoku_amd/csrc/$f.cpp -o $out/$f.o || exit 1
  objs="$objs $out/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $out/libagx_san.so alphagomoku_amd/csrc/agx_api.o alphagomoku_amd/csrc/nn_forward.o alphagomoku_amd/csrc/engine.o $objs -lz 2> $out/link.err \
  || g++ -shared -fPIC -fsanitize=address,undefined -o $out/libagx_san.so alphagomoku_amd/csrc/agx_api.o alphagomoku_amd/csrc/nn_forward.o alphagomoku_amd/csrc/engine.o $objs -L/opt/rocm/lib -lamdhip64 -lz || exit 1
export AGO_LIB_PATH=$out/libagoracle_san.so AGX_LIB_PATH=$out/libagx_san.so
export LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:log_path=$out/asan UBSAN_OPTIONS=print_stacktrace=1:log_path=$out/ubsan
timeout 3000 python -m pytest tests -m "not gpu" -q -x -p no:cacheprovider 2>&1 | tail -6
echo "sanitizer reports:"; ls $out | grep -c "^asan\.\|^ubsan\." ; head -40 $out/asan.* $out/ubsan.* 2>/dev/null
