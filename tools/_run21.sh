cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_gpu.sh > gpurun_out/profile_gpu.log 2>&1
timeout 900 python bench.py > gpurun_out/r21_bench_default.json 2> gpurun_out/r21_err.txt
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --steps 5 --opt wgrad_async=0 --shapes --profile-out gpurun_out/r21_serial_shapes.tsv > /dev/null 2>> gpurun_out/r21_err.txt
timeout 2800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r21_tests.txt
