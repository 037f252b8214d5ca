#!/bin/bash
# same-box A/B of whole-step variants: tools/probes/ab_step.sh "<label>|<env assignments>|<bench args>" ...   (3 interleaved repetitions)
for rep in 1 2 3; do
  for spec in "$@"; do
    IFS='|' read -r label envs args <<< "$spec"
    line=$(env $envs python bench.py --no-cpu-baseline --skip-strict-fp32 --skip-serialized --steps 40 $args 2>/dev/null | grep '^{')
    echo "$label rep$rep $(python -c "import json,sys; r=json.loads(sys.argv[1]); print(r['ms_per_step'], r['ms_per_step_median'], r['roofline']['avg_launch_ms'])" "$line")"
  done
done
