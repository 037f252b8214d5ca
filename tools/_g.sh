python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-serialized 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
