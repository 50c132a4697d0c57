#!/bin/bash
# Samples rocm-smi power / clocks while a long rollout bench runs (is the sustained regime power-limited?).
python bench.py --steps ${STEPS:-60000} --warmup 4 --no-cpu-baseline > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
for i in $(seq 40); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "t=$i $(rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E 'Power|sclk|GPU use' | tr -s ' ' | tr '\n' ';')"
  sleep 1
done
wait $BP
cut -c1-250 gpurun_out/power_bench.json
