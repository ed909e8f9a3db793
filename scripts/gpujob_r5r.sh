#!/bin/bash
# round 5, job R: on top of 16 waves per compute unit — how the 2 752 bytes of LDS are split between action stack and frames (A<entries>), and the instruction-count
# variants that were flat at 12 waves (TM merged centre edit, TE evaluate weights as immediates, TN non-temporal pattern gathers, TMEN all + T2); AGX_QUICK builds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
run() {
  v=$1; shift
  if [ "$v" = "-" ]; then cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so; else cp alphagomoku_amd/libagx_$v.so alphagomoku_amd/libagx.so; fi
  AGX_NO_BUILD=1 python bench.py --steps 300 --warmup 30 --age-steps 1500 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v $*', '->', round(d['value']), round(d['ms_per_step'],2), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, d['speculative_solver'])"
}
{
for v in A448 A480 A416 A384 TM TE TN TMEN A448 A480 A416 A384 TM TE TN TMEN; do run $v; done
} > gpurun_out/r5r_tune2.txt 2>&1
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
cat gpurun_out/r5r_tune2.txt
