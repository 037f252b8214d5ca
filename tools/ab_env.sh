#!/bin/bash
# A/B of environment switches / bench options on ONE box: tools/ab_env.sh "ENV=.. --opt .." ...  (each run twice, interleaved)
for rep in 1 2; do
  for o in "$@"; do
    envs=""; args=""
    for t in $o; do case "$t" in *=*) if [[ "$t" == --* ]]; then args="$args $t"; else envs="$envs $t"; fi;; *) args="$args $t";; esac; done
    ms=$(env $envs python bench.py --no-cpu-baseline --skip-strict-fp32 --skip-serialized $args 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "rep $rep [$o] $ms ms"
  done
done
