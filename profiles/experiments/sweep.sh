#!/bin/bash
# N / noise sweep of the rollout bench (one line per config) -> DESIGN.md section 7 table
for A in "--envs 4096" "--envs 32768" "--envs 65536" "--envs 131072" "--envs 262144" "--envs 65536 --noise" "--envs 1048576 --noise" "--envs 65536 --variant no-auto-reset"; do
  python bench.py --steps ${STEPS:-128} --warmup 4 --no-cpu-baseline $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$A', round(d['value']/1e9,2), 'G steps/s', round(d['roofline']['launch_us'],1), 'us', round(d['roofline']['achieved'],0), 'GB/s frac', round(d['roofline']['frac'],3), d['roofline']['kernel'])"
done
