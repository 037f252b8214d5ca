export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf $R/gpurun_out/ktrace; mkdir -p $R/gpurun_out/ktrace
(cd $R && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o kt -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --skip-serialized --opt wgrad_async=0 > /dev/null 2> $R/gpurun_out/ktrace/err.log)
find $R/gpurun_out/ktrace -mindepth 2 -name "*.csv" -exec mv {} $R/gpurun_out/ktrace/ \;
cd $R; python tools/kernel_gaps.py gpurun_out/ktrace/kt_kernel_trace.csv > gpurun_out/gaps.log 2>&1
rm -f gpurun_out/ktrace/kt_kernel_trace.csv
