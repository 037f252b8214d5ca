#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: run ON THE GPU BOX from the repo root
#   gpurun -- 'bash tools/profile_gpu.sh'  then (here)  python tools/summarize_rocprof.py r04
# Pass 1: --kernel-trace --stats of the SAME command the bench line comes from (bench.py, default workload).
# Passes 2-3 profile FOUR steps (3 warm-up + 1: PROFILE_STEPS=4 for tools/summarize_rocprof.py), so that first-use work (lazy weight
# packs, arena memsets) is a small part of the per-step average.  Passes 2-4: PMC counters, each in its own run (MI355X_MICROARCH.md, rocprofv3 section): FETCH_SIZE, WRITE_SIZE
# (HBM traffic) and the MFMA-busy / GPU-active cycle counters of the dominant kernel.
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
for d in prof_stats prof_fetch prof_write prof_sq; do rm -rf $R/gpurun_out/$d; mkdir -p $R/gpurun_out/$d; done
run() { (cd $R && "$@"); }
run rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-serialized --skip-strict-fp32 > $R/gpurun_out/prof_stats/bench.json 2> $R/gpurun_out/prof_stats/err.log
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o r01 -- \
    python bench.py --steps 1 --warmup 3 --no-cpu-baseline --skip-serialized --skip-strict-fp32 > /dev/null 2> $R/gpurun_out/prof_fetch/err.log
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o r01 -- \
    python bench.py --steps 1 --warmup 3 --no-cpu-baseline --skip-serialized --skip-strict-fp32 > /dev/null 2> $R/gpurun_out/prof_write/err.log
run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d $R/gpurun_out/prof_sq -o r01 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-serialized --skip-strict-fp32 --opt wgrad_async=0 \
    > /dev/null 2> $R/gpurun_out/prof_sq/err.log
# rocprofv3 may nest its files under <host>/<pid>: flatten, and drop the per-launch traces (large)
for d in prof_stats prof_fetch prof_write prof_sq; do
  find $R/gpurun_out/$d -mindepth 2 -name "*.csv" -exec mv {} $R/gpurun_out/$d/ \;
  find $R/gpurun_out/$d -name "*kernel_trace.csv" -size +8M -delete
done
ls -la $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_sq | head -40
tail -2 $R/gpurun_out/prof_stats/bench.json | cut -c1-300
