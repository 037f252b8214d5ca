#!/bin/bash
# kernel trace of 3 warm steps of the default bench -> gpurun_out/prof_tr/{step_trace.txt,timeline.txt}  (GPU box, repo root)
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/prof_tr; mkdir -p $R/gpurun_out/prof_tr
(cd $R && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tr -o tr -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --skip-serialized --skip-strict-fp32 "$@" > $R/gpurun_out/prof_tr/bench.json 2> $R/gpurun_out/prof_tr/err.log)
find $R/gpurun_out/prof_tr -mindepth 2 -name "*.csv" -exec mv {} $R/gpurun_out/prof_tr/ \;
cd $R; python tools/step_trace.py gpurun_out/prof_tr/tr_kernel_trace.csv > gpurun_out/prof_tr/step_trace.txt 2>&1
python tools/stream_timeline.py gpurun_out/prof_tr/tr_kernel_trace.csv > gpurun_out/prof_tr/timeline.txt 2>&1
tail -4 gpurun_out/prof_tr/step_trace.txt
find gpurun_out/prof_tr -name "*kernel_trace.csv" -size +20M -delete
