for rep in 1 2; do
MSEGK_BWD_FUSE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('python unfused', d['ms_per_step'], d['value'])"
for m in 0 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt bwd_fuse=$m 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bwd_fuse=$m', d['ms_per_step'], d['value'])"
done
done
