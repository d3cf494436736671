#!/bin/bash
# Builds variants of libworogen.so that differ in ONE compile-time constant each (research/ab/variants/libworogen_<name>.so), for A/B runs with WO_LIBWOROGEN=<path>.
# Usage: research/ab/build_variants.sh  (from anywhere; takes a few minutes)
set -e
cd "$(dirname "$0")/../../planet_heightmap_generation_amd/csrc"
make -s -j8
V=../../research/ab/variants
mkdir -p $V
build_variant() {   # name, object to rebuild, flags
  local name=$1 obj=$2 flags=$3
  rm -rf $V/build_$name; cp -rp build $V/build_$name; rm -f $V/build_$name/$obj
  make -s -j8 B=$V/build_$name OUT=$V/libworogen_$name.so EXTRA="$flags" $V/build_$name/$obj
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $V/build_$name/*.o -o $V/libworogen_$name.so -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
  rm -rf $V/build_$name
  echo "built $name"
}
build_variant rs8 radix.hip.o -DWO_RS_ITEMS=8
build_variant rs32 radix.hip.o -DWO_RS_ITEMS=32
build_variant tw8 planet.hip.o -DWO_THERMAL_WAVES=8
build_variant tw4 planet.hip.o -DWO_THERMAL_WAVES=4
build_variant br512 basin.hip.o -DWO_BASIN_RANGE_SLOTS=512
build_variant br128 basin.hip.o -DWO_BASIN_RANGE_SLOTS=128
