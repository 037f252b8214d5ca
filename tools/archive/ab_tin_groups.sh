# A/B of option "wbf_tin_groups" (W-tile chunking of the transform kernels) on one box: bash tools/ab_tin_groups.sh "0 -1 2048 4096"
for rep in 1 2; do for g in ${1:-0 -1 2048 4096 8192}; do
python bench.py --no-cpu-baseline --skip-strict-fp32 --opt wbf_tin_groups=$g 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('groups $g', 'step', b['ms_per_step'], 'median', b['ms_per_step_median'], 'serial', r['hbm']['serialized_kernel_ms_per_step'], 'transforms', r['hbm']['buckets']['lu_transforms']['ms'])"
done; done
for rep in 1 2; do for g in ${1:-0 -1 2048 4096 8192}; do echo -n "groups $g  "; python tools/bench_workloads.py --model VNet --steps 10 --warmup 3 --opt wbf_tin_groups=$g | tail -1; done; done
