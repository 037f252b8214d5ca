cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r15_bench_default.json 2> gpurun_out/r15_err.txt
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --steps 5 --opt wgrad_async=0 --shapes --profile-out gpurun_out/r15_serial_shapes.tsv > /dev/null 2>> gpurun_out/r15_err.txt
timeout 600 python tools/bench_workloads.py --model VNet --json-out gpurun_out/r15_mri_vnet.json > gpurun_out/r15_wl.txt 2>&1
timeout 600 python tools/bench_workloads.py --model VNetDeepSup --json-out gpurun_out/r15_mri_deepsup.json >> gpurun_out/r15_wl.txt 2>&1
timeout 600 python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp16 --json-out gpurun_out/r15_unet_fp16.json >> gpurun_out/r15_wl.txt 2>&1
timeout 600 python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp32 --json-out gpurun_out/r15_unet_fp32.json >> gpurun_out/r15_wl.txt 2>&1
timeout 600 python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp16 --profile-out gpurun_out/r15_unet_fp16_events.tsv >> gpurun_out/r15_wl.txt 2>&1
timeout 600 python tools/bench_workloads.py --model VNet --profile-out gpurun_out/r15_mri_vnet_events.tsv >> gpurun_out/r15_wl.txt 2>&1
