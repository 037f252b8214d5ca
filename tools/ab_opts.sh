#!/bin/bash
# interleaved A/B of option sets on the default bench: tools/ab_opts.sh RUNS "k=v k=v" "k=v" ...   ("-" = defaults)   (GPU box, repo root)
B="python bench.py --no-cpu-baseline --skip-serialized --skip-strict-fp32"
f() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_without_events'), d['final_loss'])"; }
runs=$1; shift
for i in $(seq 1 $runs); do
  for set in "$@"; do
    o=""; if [ "$set" != "-" ]; then for kv in $set; do o="$o --opt $kv"; done; fi
    echo -n "[$set] "; $B $o | f
  done
done
