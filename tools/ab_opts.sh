# usage: tools/ab_opts.sh "opts1" "opts2" ...   each a space-separated list of KEY=INT for bench.py --opt ("-" = none)
for o in "$@"; do
  args=""; [ "$o" != "-" ] && for kv in $o; do args="$args --opt $kv"; done
  python bench.py --no-cpu-baseline --skip-strict-fp32 $args 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('[$o]', 'step', b['ms_per_step'], 'median', b['ms_per_step_median'], 'wgrad', r['wgrad_kernel']['avg_launch_ms'], 'gemm', r['avg_launch_ms'], 'serial', r['hbm']['serialized_kernel_ms_per_step'], 'loss', b['final_loss'])"
done
