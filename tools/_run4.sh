cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { # name env... -- opts
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --skip-serialized $OPTS 2>> gpurun_out/r4_err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', j['ms_per_step'], j['roofline']['avg_launch_ms'])" >> gpurun_out/r4.txt
}
OPTS="--opt wgrad_fork=0" run "prio=low fork=0" X=1
OPTS="--opt wgrad_fork=1" run "prio=low fork=1" X=1
OPTS="--opt wgrad_fork=0" run "prio=norm fork=0" MSEGK_SIDE_PRIORITY=0
OPTS="--opt wgrad_fork=1" run "prio=norm fork=1" MSEGK_SIDE_PRIORITY=0
OPTS="--opt wgrad_fork=0" run "cufrac=2 fork=0" MSEGK_SIDE_CU_FRAC=2
OPTS="--opt wgrad_fork=1" run "cufrac=2 fork=1" MSEGK_SIDE_CU_FRAC=2
OPTS="--opt wgrad_fork=0" run "cufrac=4 fork=0" MSEGK_SIDE_CU_FRAC=4
OPTS="--opt wgrad_fork=1" run "cufrac=4 fork=1" MSEGK_SIDE_CU_FRAC=4
OPTS="--opt wgrad_async=0" run "serial" X=1
