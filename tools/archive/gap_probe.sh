#!/bin/bash
# launch gaps by kernel name for library variants: tools/gap_probe.sh <tag>=<lib.so> ...   (GPU box, repo root; serial run)
R=$(pwd); export TMPDIR=/tmp
for kv in "$@"; do
  tag=${kv%%=*}; lib=${kv#*=}
  d=$R/gpurun_out/gap_$tag; rm -rf $d; mkdir -p $d
  (cd $R && MSEGK_LIB=$R/$lib rocprofv3 --kernel-trace --output-format csv -d $d -o tr -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --skip-serialized --skip-strict-fp32 --opt wgrad_async=0 > $d/bench.json 2> $d/err.log)
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python tools/gap_by_kernel.py $d/tr_kernel_trace.csv > $d/gaps.txt 2>&1
  echo "== $tag"; head -12 $d/gaps.txt
  find $d -name "*.csv" -size +1M -delete
done
