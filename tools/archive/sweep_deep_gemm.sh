#!/bin/bash
# matrix stage of the LUConv levels in isolation, tile variant by shape (GPU box, repo root): tools/sweep_deep_gemm.sh
# variants (msk_conv_wbf.hip kVars): CN 32: 4 (MR 2) / 0 / 6 (MR 4);  CN 64: 5 (MR 2) / 1 (MR 4);  CN 128+: 3 (MR 2) / 2 (MR 4)
run() { python tools/bench_conv.py --c $1 --size $2 --iters 20 --profile "${@:3}" 2>/dev/null | grep -E "wbf_gemm|c=" | head -4 | tr '\n' ' ' | sed 's/  */ /g'; echo; }
for shape in "128 32" "128 16" "256 16" "256 8"; do set -- $shape; for v in 3 2; do for f in 1 0; do echo -n "c=$1 s=$2 variant=$v fuse=$f : "; run $1 $2 --opt wbf_variant=$v --opt wbf_fuse=$f; done; done; done
for shape in "64 64" "64 32"; do set -- $shape; for v in 5 1; do for f in 1 0; do echo -n "c=$1 s=$2 variant=$v fuse=$f : "; run $1 $2 --opt wbf_variant=$v --opt wbf_fuse=$f; done; done; done
for shape in "32 64"; do set -- $shape; for v in 4 0 6; do for f in 1 0; do echo -n "c=$1 s=$2 variant=$v fuse=$f : "; run $1 $2 --opt wbf_variant=$v --opt wbf_fuse=$f; done; done; done
