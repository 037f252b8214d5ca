bash tools/profile_gpu.sh > gpurun_out/profile_gpu.log 2>&1
python bench.py > gpurun_out/bench_line_default.json 2> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --steps 5 --opt wgrad_async=0 --shapes --profile-out gpurun_out/bench_hip_events_serial_shapes.tsv > gpurun_out/bench_shapes.json 2>/dev/null
tail -1 gpurun_out/bench_line_default.json | cut -c1-200
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
