#!/bin/bash
# ONE evidence pass of round 5 on one box (GPU box, repo root): the bench line, its serialized per-shape profile, the per-launch trace
# and two-stream timeline, the rocprofv3 kernel stats + HBM counters of the same command, the configs[3]/[4] workloads (with the
# in-loop preprocessing) and the preprocessing roofline.  Copy what is to be judged into profiles/ with tools/collect_r05.py.
python bench.py > gpurun_out/e_bench_line.json 2> gpurun_out/e_bench_line.err; cut -c1-240 gpurun_out/e_bench_line.json
python bench.py --no-cpu-baseline --skip-strict-fp32 --opt wgrad_async=0 --shapes --profile-out gpurun_out/e_serial_shapes.tsv > gpurun_out/e_serial.json 2>/dev/null
bash tools/trace_gpu.sh > gpurun_out/e_trace.log 2>&1; tail -2 gpurun_out/e_trace.log
bash tools/trace_gpu.sh --opt wgrad_async=0 > /dev/null 2>&1; mv gpurun_out/prof_tr/step_trace.txt gpurun_out/e_step_trace_serial.txt; bash tools/trace_gpu.sh > /dev/null 2>&1
bash tools/profile_gpu.sh > gpurun_out/e_profile_gpu.log 2>&1
python tools/bench_workloads.py --model VNet --shape 512,512,12 --num-classes 20 --batch 1 --inloop-preprocess --steps 8 --json-out gpurun_out/e_mri_vnet.json 2>&1 | grep -E "ms/step" 
python tools/bench_workloads.py --model VNetDeepSup --shape 512,512,12 --num-classes 20 --batch 1 --json-out gpurun_out/e_mri_ds.json 2>&1 | grep -E "ms/step"
python tools/bench_workloads.py --model UNet3D --precision fp16 --shape 192,192,64 --num-classes 3 --batch 2 --json-out gpurun_out/e_unet_fp16.json 2>&1 | grep -E "ms/step"
python tools/bench_preprocess_roofline.py --out gpurun_out/e_preprocess.json > gpurun_out/e_preprocess.txt 2>&1
python tools/bench_ks.py --levels 2 --fine-ld 32 --sets 4 > gpurun_out/e_bench_ks.txt 2>&1
