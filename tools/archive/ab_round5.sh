#!/bin/bash
# A/B of bench.py options on ONE box: tools/ab_round5.sh "<opts A>" "<opts B>" ...  (each run twice, interleaved)
B="python bench.py --no-cpu-baseline --skip-strict-fp32 --skip-serialized"
for rep in 1 2; do
  for o in "$@"; do
    ms=$($B $o 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "rep $rep [$o] $ms ms"
  done
done
