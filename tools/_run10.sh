cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1000 python -m pytest tests/test_gpu_wbf.py -q -m gpu -s -k "sparse_outliers or real_loss_gradient" 2>&1 | grep -E "outliers 2|dy\|:|dgrad |passed|failed" | cut -c1-400 > gpurun_out/r10_adv.txt
timeout 1500 python -m pytest tests/test_gpu_wbf.py tests/test_gpu_unet3d.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r10_tests.txt
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --skip-serialized 2>> gpurun_out/r10_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['final_loss'])" >> gpurun_out/r10.txt; done
