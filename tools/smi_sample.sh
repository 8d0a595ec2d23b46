#!/bin/bash
# sample power / clocks while the bench runs (ordinary user, read-only)
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-leg --no-kernel-timing > gpurun_out/smi_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 1 14); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr '\n' ';' ; echo
  sleep 1
done
wait $BP
tail -c 300 gpurun_out/smi_bench.json
