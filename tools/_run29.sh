cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wbf.py tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "fwd_ex_stats or vnet or trajectory or c1 or in_tr" 2>&1 | tail -4 > gpurun_out/r29.txt
for i in 1 2; do python bench.py --no-cpu-baseline --skip-serialized 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['ms_per_step_median'], j['final_loss'])" >> gpurun_out/r29.txt; done
