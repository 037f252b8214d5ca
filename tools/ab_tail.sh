#!/bin/bash
# tail of the step: eager updates beside the late weight gradient (default) vs behind it; wgrad_c1 workgroups per CU (GPU box, repo root)
B="python bench.py --no-cpu-baseline --skip-serialized --skip-strict-fp32"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_without_events'), d['final_loss'])"; }
for i in 1 2 3; do
  echo -n "default          : "; $B | f
  echo -n "eager_tail_main=0: "; $B --opt eager_tail_main=0 | f
  echo -n "wgrad_c1_wpc=3   : "; $B --opt wgrad_c1_wpc=3 | f
  echo -n "wgrad_c1_wpc=4   : "; $B --opt wgrad_c1_wpc=4 | f
done
