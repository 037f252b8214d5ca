python -m pytest tests/test_gpu_ops.py -q -x -k "interp" 2>&1 | tail -2
python -m pytest tests/test_gpu_deepsup.py -q -x 2>&1 | tail -2
python tools/bench_workloads.py --model VNetDeepSup --steps 5 2>&1 | tail -1
