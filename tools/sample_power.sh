#!/bin/bash
# Diagnostic: sample GPU clock and power while the inference bench runs (is the body sweep power-limited?)
python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-fp32-extra > gpurun_out/bench_power.json 2> gpurun_out/bench_power.err &
BP=$!
for i in $(seq 1 45); do
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*: //' | tr '\n' ' ')"
  sleep 1
done
wait $BP
tail -c 300 gpurun_out/bench_power.json
