#!/bin/bash
# Build an A/B variant of the library: scripts/build_variant.sh <name> [-DFLAG ...]  ->  scripts/diag/lib_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../denet_amd/csrc"
tmp=$(mktemp -d)
for f in api masked_sum proto_extend classify upsample_loss metrics head_step; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function "$@" -c $f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/diag/lib_$name.so $tmp/*.o
rm -rf $tmp
