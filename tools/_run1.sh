python -m pytest tests/test_gpu_wbf.py -q -s 2>&1 | grep -E "^wbf|passed|failed|Error|error" | head -40
for v in 0 4; do python tools/bench_conv.py --c 32 --size 128 --opt wbf_variant=$v --profile 2>&1 | grep -E "wbf_|fwd|dgrad"; done
for v in 0 4; do python tools/bench_conv.py --c 32 --size 64 --opt wbf_variant=$v 2>&1 | grep -E "fwd|dgrad"; done
python tools/bench_conv.py --c 32 --size 64 --opt wino_bf3=0 2>&1 | grep -E "fwd|dgrad"
