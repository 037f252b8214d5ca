cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for m in 1 0; do
MSEGK_DP_MODE=$m timeout 600 python bench.py --no-cpu-baseline --skip-serialized --steps 5 --force-syncbn-collectives --profile-out gpurun_out/r24_mode$m.tsv > gpurun_out/r24_mode$m.json 2>> gpurun_out/r24_err.txt
done
