set -x
python bench.py > gpurun_out/e_bench_line.json 2> gpurun_out/e_bench_line.err
python bench.py --no-cpu-baseline --skip-strict-fp32 --opt wgrad_async=0 --shapes --profile-out gpurun_out/e_serial_shapes.tsv > gpurun_out/e_serial.json 2>/dev/null
bash tools/profile_gpu.sh > gpurun_out/e_profile_gpu.log 2>&1
bash tools/timeline_gpu.sh > gpurun_out/e_timeline.log 2>&1
python tools/bench_workloads.py --model VNet --shape 512,512,12 --num-classes 20 --batch 1 --json-out gpurun_out/e_mri_vnet.json --profile-out gpurun_out/e_mri_vnet.tsv > gpurun_out/e_mri_vnet.log 2>&1
python tools/bench_workloads.py --model VNetDeepSup --shape 512,512,12 --num-classes 20 --batch 1 --json-out gpurun_out/e_mri_ds.json > gpurun_out/e_mri_ds.log 2>&1
python tools/bench_workloads.py --model UNet3D --precision fp16 --shape 192,192,64 --num-classes 3 --batch 2 --json-out gpurun_out/e_unet_fp16.json --profile-out gpurun_out/e_unet_fp16.tsv > gpurun_out/e_unet_fp16.log 2>&1
python tools/bench_workloads.py --model UNet3D --precision fp32 --shape 192,192,64 --num-classes 3 --batch 2 --json-out gpurun_out/e_unet_fp32.json > gpurun_out/e_unet_fp32.log 2>&1
python -m pytest tests/test_gpu_fullsize_parity.py -q -s 2>&1 | grep -v "^$" > gpurun_out/e_fullsize_parity.log
tail -3 gpurun_out/e_fullsize_parity.log
cut -c1-200 gpurun_out/e_bench_line.json
