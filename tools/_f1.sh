set -x
python -m pytest tests/test_gpu_ops.py -q -x -k "foldn or fold_bn or (fwd_dgrad_wgrad and 32-)" 2>&1 | tail -15 > gpurun_out/f1.log
python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 5 --profile > gpurun_out/f1b.log 2>&1
python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 5 --impl 22 --profile > gpurun_out/f1c.log 2>&1
