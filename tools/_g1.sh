python -m pytest tests/test_gpu_ops.py -q -x -k foldn 2>&1 | tail -1
bash tools/pmc_kernel.sh conv_foldn -- python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 3 2>&1 | grep "LDS_BANK\|LDS_IDX\|MFMA_BUSY"
python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 20 --profile 2>&1 | grep "conv_foldn"
bash tools/pmc_kernel.sh wbf_wgrad_k -- python tools/bench_conv.py --c 32 --size 128 --iters 2 2>&1 | grep "LDS_BANK\|LDS_IDX\|MFMA_BUSY\|GRBM"
bash tools/pmc_kernel.sh wbf_tout_k -- python tools/bench_conv.py --c 32 --size 128 --iters 2 2>&1 | grep "LDS_BANK\|LDS_IDX\|GRBM"
