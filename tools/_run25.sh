cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r25.txt
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --skip-serialized "$@" 2>> gpurun_out/r25_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', j['ms_per_step'], j['ms_per_step_median'])" >> gpurun_out/r25.txt; }
run "plain                 "
MSEGK_DP_MODE=3 run "forced dp_mode 3      " --force-syncbn-collectives
MSEGK_DP_MODE=0 run "forced dp_mode 0      " --force-syncbn-collectives
MSEGK_DP_MODE=2 run "forced dp_mode 2      " --force-syncbn-collectives
MSEGK_DP_MODE=3 run "forced dp_mode 3      " --force-syncbn-collectives
MSEGK_DP_MODE=0 MSEGK_DP_OVERLAP=0 run "forced mode 0 no-overlap" --force-syncbn-collectives
