#!/bin/bash
# Knock-out probes of wbf_gemm_fused_k (msk_conv_wbf.hip, -DWBF_PROBE=bits): which resource bounds the 32-channel matrix stage?
# Builds are made with tools/ab_build.sh msk_conv_wbf.hip probe<bits> -DWBF_PROBE=<bits>; run on the GPU box:
#   bash tools/probe_fused.sh [c] [size] > gpurun_out/r06/probe_fused.txt
c=${1:-32}; s=${2:-128}
for v in 0 32 1 2 4 8 16 6 15; do
  lib=medicalseg_amd/lib/ab/libmsegk_probe$v.so
  [ -f $lib ] || continue
  echo "== WBF_PROBE=$v (1 no tile refill, 2 no B loads, 4 no A reads, 8 no stores, 16 no MFMAs)"
  MSEGK_LIB=$lib python tools/bench_conv.py --c $c --size $s --iters 5 --profile 2>&1 | grep -E "wbf_gemm|fwd |dgrad " | head -6
done
