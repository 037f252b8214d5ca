#!/bin/bash
# sample sclk / power while the training step runs (is the step power-limited?): tools/probes/clock_sample.sh > gpurun_out/r06/clocks.txt
python bench.py --no-cpu-baseline --skip-strict-fp32 --skip-serialized --steps 500 --warmup 20 > /tmp/clk_line.json 2>/dev/null &
pid=$!
sleep 6
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower -d 0 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr '\n' ';'; echo
  sleep 0.4
done
wait $pid
python -c "import json; r=json.loads([l for l in open('/tmp/clk_line.json') if l.startswith('{')][0]); print('ms_per_step', r['ms_per_step'])"
echo "idle:"; sleep 2; rocm-smi --showclocks --showpower -d 0 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ';'; echo
