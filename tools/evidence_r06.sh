#!/bin/bash
# ONE evidence pass of round 6 on one box (GPU box, repo root): the driver's bench line, its serialized per-shape profile, the
# rocprofv3 kernel stats + HBM counters + MFMA-busy counters of the same command (tools/profile_gpu.sh), the configs[3] / [4]
# workloads and the clocks while the step runs.  Afterwards (build container): python tools/summarize_rocprof.py r06 and copy what is
# to be judged from gpurun_out/ into profiles/.
mkdir -p gpurun_out
python bench.py > gpurun_out/e_bench_line.json 2> gpurun_out/e_bench_line.err; cut -c1-240 gpurun_out/e_bench_line.json
python bench.py --no-cpu-baseline --skip-strict-fp32 --opt wgrad_async=0 --shapes --profile-out gpurun_out/e_serial_shapes.tsv > gpurun_out/e_serial.json 2>/dev/null
bash tools/profile_gpu.sh > gpurun_out/e_profile_gpu.log 2>&1
python tools/bench_workloads.py --model VNet --shape 512,512,12 --num-classes 20 --batch 1 --inloop-preprocess --steps 8 --json-out gpurun_out/e_mri_vnet.json 2>&1 | grep -E "ms/step"
python tools/bench_workloads.py --model VNetDeepSup --shape 512,512,12 --num-classes 20 --batch 1 --json-out gpurun_out/e_mri_ds.json 2>&1 | grep -E "ms/step"
python tools/bench_workloads.py --model UNet3D --precision fp16 --shape 192,192,64 --num-classes 3 --batch 2 --json-out gpurun_out/e_unet_fp16.json 2>&1 | grep -E "ms/step"
python tools/bench_preprocess_roofline.py --out gpurun_out/e_preprocess.json > gpurun_out/e_preprocess.txt 2>&1
bash tools/probes/clock_sample.sh > gpurun_out/e_clocks.txt 2>&1; tail -3 gpurun_out/e_clocks.txt
