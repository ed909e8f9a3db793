#!/bin/bash
# usage: scripts/nn_prof_variant.sh P blocks filters board — scripts/nn_prof.py with the -DAGX_NN_PROFILE variant library
cd "$(dirname "$0")/.."
cp alphagomoku_amd/libagx.so /tmp/libagx_keep.so
cp alphagomoku_amd/libagx_$1.so alphagomoku_amd/libagx.so
AGX_NO_BUILD=1 python scripts/nn_prof.py $2 $3 $4
cp /tmp/libagx_keep.so alphagomoku_amd/libagx.so
