cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r20.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --skip-serialized "$@" 2>> gpurun_out/r20_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', j['ms_per_step'], j['ms_per_step_median'], j['roofline']['avg_launch_ms'], j['final_loss'])" >> gpurun_out/r20.txt; }
run "prepack async " 
run "prepack main  " --opt wbf_prepack=2
run "prepack async "
run "prepack main  " --opt wbf_prepack=2
timeout 900 python -m pytest tests/test_gpu_wbf.py tests/test_gpu_model.py -x -q -m gpu -k "packed_weight or trajectory or checkpoint or training" 2>&1 | tail -3 >> gpurun_out/r20.txt
