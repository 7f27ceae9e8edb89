#!/bin/bash
# Diagnostic: an alternative build of the library with extra defines, for A/B runs through DENET_HIP_LIB
# (scripts/diag/ab_lib.sh).  usage: build_alt.sh <name> [-DMACRO=..]...   ->  denet_amd/libdenet_alt_<name>.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/build/alt_$name
mkdir -p $out
for f in api masked_sum proto_extend classify upsample_loss metrics head_step; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function "$@" \
      -c $root/denet_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/denet_amd/libdenet_alt_$name.so $out/*.o
echo built $root/denet_amd/libdenet_alt_$name.so
