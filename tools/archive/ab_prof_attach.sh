#!/bin/bash
# cost of the roofline kernel's HIP events in the measured step (GPU box, repo root): every step / every 4th / attached vs marker brackets
B="python bench.py --no-cpu-baseline --skip-serialized --skip-strict-fp32"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], 'with/without events', d.get('ms_per_step_with_events'), d.get('ms_per_step_without_events'), 'avg launch', r['avg_launch_ms'], r['launches'], r['frac'])"; }
for i in 1 2 3; do
  echo -n "every 4, attached : "; $B | f
  echo -n "every 1, attached : "; $B --roofline-every 1 | f
  echo -n "every 1, brackets : "; $B --roofline-every 1 --opt prof_attach=0 | f
  echo -n "every 4, brackets : "; $B --opt prof_attach=0 | f
done
