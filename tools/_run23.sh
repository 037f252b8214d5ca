cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r23.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --skip-serialized "$@" 2>> gpurun_out/r23_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', j['ms_per_step'], j['ms_per_step_median'])" >> gpurun_out/r23.txt; }
run "plain                 "
run "forced dp_mode 1      " --force-syncbn-collectives
MSEGK_DP_MODE=0 run "forced dp_mode 0      " --force-syncbn-collectives
MSEGK_DP_MODE=2 run "forced dp_mode 2      " --force-syncbn-collectives
run "plain                 "
