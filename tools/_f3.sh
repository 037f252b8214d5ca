python -m pytest tests/test_gpu_ops.py -q -x -k "convT or conv3d_fwd_dgrad" 2>&1 | tail -3
python tools/bench_ks.py --levels 4 2>&1 | grep "down fwd\|up   dgrad"
