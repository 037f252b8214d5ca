bash tools/profile_gpu.sh > gpurun_out/profile_gpu.log 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt wgrad_async=0 --shapes --profile-out gpurun_out/serial.tsv > gpurun_out/serial_line.json 2> gpurun_out/serial_err.log
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log
tail -c 600 gpurun_out/bench_line.json
