export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_stats/bench.json 2> gpurun_out/prof_stats/err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o r01 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_fetch/err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o r01 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_write/err.log
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -type f | head; cat gpurun_out/prof_stats/bench.json | cut -c1-300
