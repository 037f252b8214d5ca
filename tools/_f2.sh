python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_unet3d.py -q -x 2>&1 | tail -3
