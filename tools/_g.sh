r() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-serialized "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo auto; r; r
echo form2; r --opt bwd_fuse=2; r --opt bwd_fuse=2
python -m pytest tests/test_gpu_wbf.py tests/test_gpu_dp.py -q -x 2>&1 | tail -1
