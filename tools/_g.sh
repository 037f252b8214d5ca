python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
python tools/bench_infer.py 2>&1 | tail -2
