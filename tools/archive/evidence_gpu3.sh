# rocprofv3 passes + bench line + serialized per-shape profile of the FINAL round-4 tree, one box
python bench.py > gpurun_out/h_bench_line.json 2> gpurun_out/h_bench_line.err
python bench.py --no-cpu-baseline --skip-strict-fp32 --opt wgrad_async=0 --shapes --profile-out gpurun_out/h_serial_shapes.tsv > gpurun_out/h_serial.json 2>/dev/null
bash tools/profile_gpu.sh > gpurun_out/h_profile_gpu.log 2>&1
bash tools/timeline_gpu.sh > gpurun_out/h_timeline.log 2>&1
cut -c1-200 gpurun_out/h_bench_line.json
