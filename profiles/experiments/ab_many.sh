#!/bin/bash
# Interleaved comparison of several builds of libs2d_hip.so (headline rollout): ab_many.sh ROUNDS lib1 lib2 ...
R=$1; shift
for r in $(seq $R); do for L in "$@"; do
  S2D_LIB=$L python bench.py --steps ${STEPS:-64} --warmup 4 --no-cpu-baseline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], round(d['value']/1e9,2), 'G steps/s', round(d['roofline']['launch_us'],1), 'us')"
done; done
