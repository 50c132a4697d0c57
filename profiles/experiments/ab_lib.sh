#!/bin/bash
# Interleaved A/B of two builds of libs2d_hip.so in separate processes on ONE device (bench rollout, N=65536).
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do for L in $A $B; do
  S2D_LIB=$L python bench.py --steps 64 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], round(d['value']/1e9,2), 'G steps/s', round(d['roofline']['launch_us'],1), 'us', d['roofline']['kernel'])"
done; done
