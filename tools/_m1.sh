python tools/bench_workloads.py --model VNet --steps 5 2>&1 | tail -1
python tools/bench_workloads.py --model VNetDeepSup --steps 5 2>&1 | tail -1
python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp32 --steps 5 2>&1 | tail -1
python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp16 --steps 5 --profile-out gpurun_out/unet3d_fp16.tsv 2>&1 | tail -1
MSEGK_CONV_SPLIT=3 python tools/bench_workloads.py --model UNet3D --shape 192,192,64 --num-classes 3 --batch 2 --precision fp32 --steps 5 2>&1 | tail -1
python tools/bench_infer.py 2>&1 | tail -2
for b in 1 4 8 16; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --skip-serialized 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', d['ms_per_step'], round(d['value']/1e6,1), 'M voxels/s')"; done
python tools/winograd_numerics.py 2>&1 | tail -8
