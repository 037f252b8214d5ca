python -m pytest tests/test_gpu_wbf.py -q -k "wgrad or fwd_ex" 2>&1 | tail -2
for c_s in "32 128" "64 64" "128 32" "256 16" "256 8"; do set -- $c_s; python tools/bench_conv.py --c $1 --size $2 --profile 2>&1 | grep -E "wbf_wgrad_reduce|wgrad  " ; done
