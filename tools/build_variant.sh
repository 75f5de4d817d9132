#!/bin/bash
# A/B build of the library with extra compiler flags, out of tree:   bash tools/build_variant.sh <name> "<flags>"
#   -> _ab/libludvm_hip_<name>.so (git-ignored; travels to the GPU box with the snapshot; LUDVM_HIP_LIB=... picks it)
set -e
NAME=$1; FLAGS=$2
SRC=$(cd "$(dirname "$0")/../ludvm_amd/csrc" && pwd)
OUT=$(cd "$(dirname "$0")/.." && pwd)/_ab
B=/tmp/ludvm_variant_$NAME; mkdir -p $B $OUT
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=fast -fvisibility=hidden $FLAGS"
pids=()
for u in context launch comm order induce wake march flowfield spatial_order; do
  $CXX -c -o $B/$u.o $SRC/$u.hip & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libludvm_hip_$NAME.so $B/*.o
rm -rf $B
echo $OUT/libludvm_hip_$NAME.so
