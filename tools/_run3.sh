cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wbf.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3_tests.txt
timeout 600 python bench.py --no-cpu-baseline --skip-serialized > gpurun_out/r3_bench.json 2> gpurun_out/r3_err.txt
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --opt wbf_prepack=0 > gpurun_out/r3_bench_noprepack.json 2>> gpurun_out/r3_err.txt
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --steps 5 --opt wgrad_async=0 --shapes --profile-out gpurun_out/r3_serial.tsv > gpurun_out/r3_bench_serial.json 2>> gpurun_out/r3_err.txt
