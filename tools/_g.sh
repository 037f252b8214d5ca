python -m pytest tests/test_gpu_ops.py -q -x -k "join_fwd_bwd" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py tests/test_gpu_deepsup.py tests/test_gpu_dp.py tests/test_gpu_dp2.py -q -x 2>&1 | tail -3
r() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-serialized "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
r; r
