# usage: tools/ab_bench.sh tag1 tag2 ...   (tags of medicalseg_amd/lib/ab/libmsegk_<tag>.so; "default" = the product library)
for v in "$@"; do
  lib=medicalseg_amd/lib/ab/libmsegk_$v.so; [ $v = default ] && lib=medicalseg_amd/lib/libmsegk.so
  MSEGK_LIB=$lib python bench.py --no-cpu-baseline --skip-strict-fp32 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('$v', 'step', b['ms_per_step'], 'median', b['ms_per_step_median'], 'wgrad', r['wgrad_kernel']['avg_launch_ms'], 'gemm', r['serialized']['avg_launch_ms'], 'serial', r['hbm']['serialized_kernel_ms_per_step'], {k:v['ms'] for k,v in r['hbm']['buckets'].items()})"
done
