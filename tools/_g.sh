for t in 1 2 4 8; do echo tpb $t; for c in 32 64 128; do s=$((4096/c)); python tools/bench_conv.py --c $c --size $s --iters 10 --profile --opt wbf_tpb=$t 2>&1 | grep "wbf_gemm"; done; done
python -m pytest tests/test_gpu_wbf.py -q -x 2>&1 | tail -1
