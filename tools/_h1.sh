for v in -1 0 6; do echo variant $v; python tools/bench_conv.py --c 32 --size 128 --iters 10 --profile --opt conv_split=2 --opt wbf_variant=$v 2>&1 | grep "wbf_gemm\|wbf_wgrad_h2\|absmax\|wbf_tin\|wbf_ty\|wbf_tout"; done
for v in -1 1; do echo variant $v; python tools/bench_conv.py --c 64 --size 64 --iters 10 --profile --opt conv_split=2 --opt wbf_variant=$v 2>&1 | grep "wbf_gemm"; done
for v in -1 2; do echo variant $v; python tools/bench_conv.py --c 128 --size 32 --iters 10 --profile --opt conv_split=2 --opt wbf_variant=$v 2>&1 | grep "wbf_gemm"; done
python tools/bench_conv.py --c 32 --size 128 --iters 10 --profile --opt conv_split=2 --zero 2>&1 | grep "wbf_gemm\|wbf_wgrad_h2"
