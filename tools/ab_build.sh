#!/bin/bash
# A/B builds of ONE kernel file: tools/ab_build.sh <file.hip> <tag> [extra hipcc flags]  ->  medicalseg_amd/lib/ab/libmsegk_<tag>.so
# (all other objects from build/; select at run time with MSEGK_LIB=medicalseg_amd/lib/ab/libmsegk_<tag>.so)
set -e
cd "$(dirname "$0")/.."
f=$1; tag=$2; shift 2
mkdir -p build/ab medicalseg_amd/lib/ab
base=$(basename ${f%.hip})
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imedicalseg_amd/csrc -Wno-unused-value -Wno-comment "$@" -c $f -o build/ab/${base}_$tag.o
OBJS=""
for o in build/msk_*.o; do
  case $o in build/msk_dp_test.o) continue;; build/$base.o) OBJS="$OBJS build/ab/${base}_$tag.o";; *) OBJS="$OBJS $o";; esac
done
hipcc --offload-arch=gfx950 -shared -fPIC -o medicalseg_amd/lib/ab/libmsegk_$tag.so $OBJS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo medicalseg_amd/lib/ab/libmsegk_$tag.so
