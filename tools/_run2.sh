cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for fz in 1 0; do
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --opt wbf_fuse=$fz > gpurun_out/r2_bench_f$fz.json 2> gpurun_out/r2_err.txt
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --steps 5 --opt wbf_fuse=$fz --opt wgrad_async=0 --shapes --profile-out gpurun_out/r2_serial_f$fz.tsv > gpurun_out/r2_bench_serial_f$fz.json 2>> gpurun_out/r2_err.txt
done
