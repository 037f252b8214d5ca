python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])"
python -m pytest tests/test_gpu_wbf.py tests/test_gpu_model.py -q -x 2>&1 | tail -2
