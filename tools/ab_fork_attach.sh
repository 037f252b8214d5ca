#!/bin/bash
# fork point of the weight-gradient stream: attached to the last data-gradient dispatch (default) vs a marker packet (GPU box, repo root)
B="python bench.py --no-cpu-baseline --skip-serialized --skip-strict-fp32"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_without_events'), d['final_loss'])"; }
for i in 1 2 3; do
  echo -n "fork_attach=1: "; $B | f
  echo -n "fork_attach=0: "; $B --opt fork_attach=0 | f
done
