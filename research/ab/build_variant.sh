#!/bin/bash
# An alternative build of libworogen.so for research/ab/run_ab.sh: bash research/ab/build_variant.sh <name> "<extra compiler flags>" [file.hip ...]
# Only the listed translation units (default planet.hip) are recompiled with the flags; the rest comes from the in-tree build directory.
set -e
cd "$(dirname "$0")/../../planet_heightmap_generation_amd/csrc"
name=$1; flags=$2; shift 2; units=${@:-planet.hip}
make -s >/dev/null
mkdir -p build/ab_$name
objs=""
for o in build/*.o; do
  b=$(basename $o); skip=0
  for u in $units; do [ "$b" = "$u.o" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags -c $u -o build/ab_$name/$u.o 2>&1 | grep -E "error|occupancy|spill" || true
  objs="$objs build/ab_$name/$u.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../research/ab/libworogen_$name.so -lpthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built research/ab/libworogen_$name.so
