# second evidence pass of round 4 (after the ragged transform axis, the LDS-tile staging and the paired gconv_kst_k)
python -m pytest tests -m gpu -q > gpurun_out/f_gputest.log 2>&1; tail -3 gpurun_out/f_gputest.log
python bench.py > gpurun_out/f_bench_line.json 2> gpurun_out/f_bench_line.err; cut -c1-220 gpurun_out/f_bench_line.json
python tools/bench_workloads.py --model VNet --shape 512,512,12 --num-classes 20 --batch 1 --json-out gpurun_out/f_mri_vnet.json 2>&1 | tail -2 | head -1
python tools/bench_workloads.py --model VNetDeepSup --shape 512,512,12 --num-classes 20 --batch 1 --json-out gpurun_out/f_mri_ds.json 2>&1 | tail -2 | head -1
python tools/bench_workloads.py --model VNet --shape 512,512,12 --num-classes 20 --batch 1 --profile-out gpurun_out/f_mri_vnet.tsv > /dev/null 2>&1
python tools/bench_workloads.py --model UNet3D --precision fp16 --shape 192,192,64 --num-classes 3 --batch 2 --json-out gpurun_out/f_unet_fp16.json 2>&1 | tail -2 | head -1
python -m pytest tests/test_gpu_fullsize_parity.py -q -s 2>&1 | grep -v "^$" > gpurun_out/f_fullsize_parity.log; tail -2 gpurun_out/f_fullsize_parity.log
