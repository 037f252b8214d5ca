for z in "" "--zero"; do python tools/bench_conv.py --c 32 --size 128 --profile $z 2>&1 | grep -E "wbf_gemm|wbf_wgrad_k|wbf_t|fwd|dgrad|wgrad  " ; done
