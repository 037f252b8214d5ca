# A/B of one integer option on one box, headline step + serialized buckets: bash tools/ab_opt_sweep.sh reduce_vpl "64 32 16 8"
key=$1; shift
for rep in 1 2; do for g in $1; do
python bench.py --no-cpu-baseline --skip-strict-fp32 --opt $key=$g 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('$key $g', 'step', b['ms_per_step'], 'median', b['ms_per_step_median'], 'serial', r['hbm']['serialized_kernel_ms_per_step'], {k:v['ms'] for k,v in r['hbm']['buckets'].items()})"
done; done
