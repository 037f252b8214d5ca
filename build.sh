#!/bin/bash
# Build libmsegk.so (gfx950 HIP kernels + C ABI) and libmsegk_test.so (same objects + the host transport of tests/test_gpu_dp2.py).
# hipcc cross-compiles without a GPU.  Usage: ./build.sh [-v]
set -e
cd "$(dirname "$0")"
OUT=medicalseg_amd/lib
mkdir -p $OUT build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imedicalseg_amd/csrc -Wno-unused-value -Wno-comment"
OBJS=""
for f in medicalseg_amd/csrc/*.hip; do
  o=build/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ include/msegk.h -nt $o ] || [ medicalseg_amd/csrc/msk_common.h -nt $o ] || [ medicalseg_amd/csrc/msk_conv.h -nt $o ] || [ medicalseg_amd/csrc/msk_wbf.h -nt $o ]; then
    rm -f $o   # a failed compile must not leave the previous object behind (the jobs run in the background)
    hipcc $FLAGS $EXTRA -c $f -o $o &
  fi
  OBJS="$OBJS $o"
done
wait
for o in $OBJS; do
  if [ ! -f $o ]; then echo "build failed: $o was not produced" >&2; exit 1; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsegk.so $OBJS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/libmsegk.so"
# test build: the same objects, msk_dp.hip recompiled with the host transport (two ranks as two processes on ONE GPU,
# tests/test_gpu_dp2.py) -- the release library does not contain it
T=build/msk_dp_test.o
if [ ! -f $T ] || [ medicalseg_amd/csrc/msk_dp.hip -nt $T ] || [ medicalseg_amd/csrc/msk_common.h -nt $T ] || [ include/msegk.h -nt $T ]; then
  hipcc $FLAGS $EXTRA -DMSK_TEST_TRANSPORT -c medicalseg_amd/csrc/msk_dp.hip -o $T
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsegk_test.so ${OBJS/build\/msk_dp.o/$T} -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/libmsegk_test.so"

