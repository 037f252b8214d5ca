cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_deepsup.py tests/test_gpu_dp.py tests/test_gpu_dp2.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5_tests.txt
for z in 1 0; do
MSEGK_ZERO_COPY_SKIP=$z timeout 600 python bench.py --no-cpu-baseline --skip-serialized 2>> gpurun_out/r5_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('zerocopy=$z', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['final_loss'])" >> gpurun_out/r5.txt
done
