bash tools/pmc_kernel.sh conv_tk_h2 -- python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 3 2>&1 | grep "LDS\|MFMA_BUSY\|GRBM\|WAVE_CYC\|WAIT"
echo ---
bash tools/pmc_kernel.sh conv_foldn_h2 -- python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 3 2>&1 | grep "LDS\|MFMA_BUSY\|GRBM\|WAVE_CYC\|WAIT"
