#!/bin/bash
# Interleaved A/B of several builds of libs2d_hip.so on ONE device: 11v11 match rollout bench (8 192 matches).
# usage: ab_match_many.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for r in $(seq $R); do for L in "$@"; do
  S2D_LIB=$L python bench.py --task match --steps 16 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], round(d['value']/1e9,3), 'G match-steps/s', round(d['roofline']['launch_us'],1), 'us')"
done; done
