#!/bin/bash
# pipeline (S2D_ROLLOUT_WS=1) vs unified (=0) rollout kernel across batch sizes -> the kWsMaxEnvs switch point
for N in 65536 131072 196608 262144 524288 1048576; do for W in 1 0; do for X in "" "--noise"; do
  S2D_ROLLOUT_WS=$W python bench.py --steps ${STEPS:-64} --warmup 4 --no-cpu-baseline --envs $N $X 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=$N ws=$W $X', round(d['value']/1e9,2), 'G steps/s', round(d['roofline']['launch_us'],1), 'us', d['roofline']['kernel'])"
done; done; done
