export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/rp4 gpurun_out/rp5
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $R/gpurun_out/rp4 -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/rp4/err.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_WAVES -d $R/gpurun_out/rp5 -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/rp5/err.log
ls gpurun_out/rp4 gpurun_out/rp5
