#!/bin/bash
# kernel trace of 3 warm steps of the default bench -> gpurun_out/prof_tl/timeline.txt  (run on the GPU box from the repo root)
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/prof_tl; mkdir -p $R/gpurun_out/prof_tl
(cd $R && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o tl -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --skip-serialized --skip-strict-fp32 "$@" > $R/gpurun_out/prof_tl/bench.json 2> $R/gpurun_out/prof_tl/err.log)
find $R/gpurun_out/prof_tl -mindepth 2 -name "*.csv" -exec mv {} $R/gpurun_out/prof_tl/ \;
cd $R; python tools/stream_timeline.py gpurun_out/prof_tl/tl_kernel_trace.csv > gpurun_out/prof_tl/timeline.txt 2>&1
head -c 400 gpurun_out/prof_tl/bench.json; echo; cat gpurun_out/prof_tl/timeline.txt
find gpurun_out/prof_tl -name "*kernel_trace.csv" -size +20M -delete
