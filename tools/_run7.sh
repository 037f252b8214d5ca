cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for z in 1 0 1 0; do
MSEGK_FUSE_SMALL=$z timeout 600 python bench.py --no-cpu-baseline --skip-serialized 2>> gpurun_out/r7_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('fuse_small=$z', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['final_loss'])" >> gpurun_out/r7.txt
done
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --steps 5 --opt wgrad_async=0 --shapes --profile-out gpurun_out/r7_serial.tsv > gpurun_out/r7_bench_serial.json 2>> gpurun_out/r7_err.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_deepsup.py tests/test_gpu_dp.py tests/test_gpu_dp2.py tests/test_gpu_wbf.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r7_tests.txt
