python tools/bench_workloads.py --model VNet --steps 5 2>&1 | tail -1
python tools/bench_workloads.py --model VNet --steps 5 --profile-out gpurun_out/mri.tsv 2>&1 | tail -1
sort -t$'\t' -k3 -g -r gpurun_out/mri.tsv | head -16
python -m pytest tests/test_gpu_wbf.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
