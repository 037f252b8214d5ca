cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --skip-serialized "$@" 2>> gpurun_out/r16_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', j['ms_per_step'], j['ms_per_step_median'], j['roofline']['avg_launch_ms'])" >> gpurun_out/r16.txt; }
run "default       "
run "ew8           " --opt ew_cap=8
run "ew4           " --opt ew_cap=4
run "ew8 red4      " --opt ew_cap=8 --opt reduce_cap=4
run "ew4 red4      " --opt ew_cap=4 --opt reduce_cap=4
run "ew6 red6      " --opt ew_cap=6 --opt reduce_cap=6
run "default       "
run "serial        " --opt wgrad_async=0
run "serial ew4red4" --opt wgrad_async=0 --opt ew_cap=4 --opt reduce_cap=4
